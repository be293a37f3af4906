"""Probe: the kernels of ONE eager step through the drop-in modules at crop size (torch profiler table).
usage: module_step_kernels.py [group|proto|adegroup|adeproto|cocoproto|cocogroup] [ce]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
import scaleprotoseg_amd as spx
from torch.profiler import profile, ProfilerActivity
from scaleprotoseg_amd.loss import PixelWiseCrossEntropyLoss
from scaleprotoseg_amd.model_multiscale_group import PPNetMultiScale as GroupNet

class BB(nn.Module):
    def __init__(s, ch):
        super().__init__(); s.base = nn.Sequential(nn.Conv2d(3, ch, 1), nn.Conv2d(ch, ch, 1))
    def __repr__(s): return "MSC(standin)"
    def forward(s, x): return x

dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "group"
ce = len(sys.argv) > 2 and sys.argv[2] == "ce"
torch.manual_seed(0)
mk = dict(add_on_layers_type="deeplab_simple", patch_classification=True, num_scales=4)
B, H, W, K = 10, 65, 65, 19
if kind == "group":
    net = GroupNet(BB(256), 64, (228, 64, 1, 1), [], 19, num_groups=3, **mk).to(dev)
elif kind == "adegroup":
    B, K = 2, 150
    net = GroupNet(BB(256), 64, (1800, 64, 1, 1), [], 150, num_groups=3, **mk).to(dev)
elif kind == "cocoproto":
    B, K = 2, 182
    net = spx.PPNetMultiScale(BB(256), 64, (2184, 64, 1, 1), [], 182, **mk).to(dev)
elif kind == "cocogroup":
    B, K = 2, 182
    net = GroupNet(BB(256), 64, (2184, 64, 1, 1), [], 182, num_groups=3, **mk).to(dev)
elif kind == "adeproto":
    B, K = 2, 150
    net = spx.PPNetMultiScale(BB(256), 64, (1800, 64, 1, 1), [], 150, **mk).to(dev)
else:
    net = spx.PPNetMultiScale(BB(256), 64, (228, 64, 1, 1), [], 19, **mk).to(dev)
x = torch.sigmoid(torch.randn(B, 256, H, W, device=dev)).bfloat16().requires_grad_(True)
net.add_on_layers = nn.Sequential()
gl = torch.randn(B, H, W, K, device=dev) * 1e-3
tgt = torch.randint(0, K + 1, (B, H, W), device=dev)
lossf = PixelWiseCrossEntropyLoss(ignore_index=-1)
params = [p for p in net.parameters() if p.requires_grad]
def step():
    x.grad = None
    for p in params: p.grad = None
    if ce:
        logits, dist = net.forward_from_conv_features(x, ce_target=tgt)
        lossf(logits, tgt).backward()
    else:
        logits, dist = net.forward_from_conv_features(x)
        torch.autograd.backward([logits], [gl])
for _ in range(5): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = [(e.key, e.device_time_total, e.count) for e in prof.key_averages() if e.device_time_total > 0 and e.device_type.name != "CPU"]
rows.sort(key=lambda r: -r[1])
tot = 0
for k, t, n in rows:
    print(f"{t:8.1f} us x{n:<3d} {k[:110]}"); tot += t
print(f"total device {tot:.1f} us, {sum(r[2] for r in rows)} launches")
