"""Probe: fwd+bwd step time THROUGH the drop-in modules at the reference's training shapes, eager and as a HIP graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
import scaleprotoseg_amd as spx
from scaleprotoseg_amd.graphs import capture_step
from scaleprotoseg_amd.loss import PixelWiseCrossEntropyLoss
from scaleprotoseg_amd.model_multiscale_group import PPNetMultiScale as GroupNet

class BB(nn.Module):
    def __init__(s, ch):
        super().__init__(); s.base = nn.Sequential(nn.Conv2d(3, ch, 1), nn.Conv2d(ch, ch, 1))
    def __repr__(s): return "MSC(standin)"
    def forward(s, x): return x

if os.environ.get("SPX_PROBE_HEAD_ROWS"):      # experiment: heads above this many rows take the product kernels instead of the fused head
    import scaleprotoseg_amd.model_multiscale as _mm
    _mm.MAX_FUSED_HEAD_ROWS = int(os.environ["SPX_PROBE_HEAD_ROWS"])
dev = torch.device("cuda:0")
ONLY = sys.argv[1] if len(sys.argv) > 1 else ""


def run(name, net, B, H, W, K, ce=False):
    if ONLY and ONLY not in name:
        return
    C = net.prototype_vectors.shape[1] * net.num_scales
    x = torch.sigmoid(torch.randn(B, C, H, W, device=dev)).bfloat16().requires_grad_(True)
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
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t) / 50 * 1e3
    g, _ = capture_step(step, warmup=2)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(100): g.replay()
    torch.cuda.synchronize(); gr = (time.perf_counter() - t) / 100 * 1e3
    print(f"{name}: eager {eager:.3f} ms, graph {gr:.3f} ms", flush=True)

torch.manual_seed(0)
mk = dict(add_on_layers_type="deeplab_simple", patch_classification=True, num_scales=4)
net = spx.PPNetMultiScale(BB(256), 64, (228, 64, 1, 1), [], 19, **mk).to(dev)
run("cityscapes prototype phase 10x65x65", net, 10, 65, 65, 19)
run("cityscapes prototype phase 10x65x65 + fused CE", net, 10, 65, 65, 19, ce=True)
net = GroupNet(BB(256), 64, (228, 64, 1, 1), [], 19, num_groups=3, **mk).to(dev)
run("cityscapes group phase 10x65x65", net, 10, 65, 65, 19)
run("cityscapes group phase 10x65x65 + fused CE", net, 10, 65, 65, 19, ce=True)
net = spx.PPNetMultiScale(BB(256), 64, (1800, 64, 1, 1), [], 150, **mk).to(dev)
run("ade prototype phase 2x65x65", net, 2, 65, 65, 150)
net = GroupNet(BB(256), 64, (1800, 64, 1, 1), [], 150, num_groups=3, **mk).to(dev)
run("ade group phase 2x65x65 (450 units: fp32 MFMA product kernels)", net, 2, 65, 65, 150)
