"""Probe: kernel-level profile of one KLDLoss forward + backward on gathered GPU planes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import scaleprotoseg_amd as spx
def _identity(P, K, S):
    ident = torch.zeros(P, K)
    per_scale, per_cs = P // S, P // K // S
    for s in range(S):
        for k in range(K):
            ident[s * per_scale + k * per_cs : s * per_scale + (k + 1) * per_cs, k] = 1
    return ident
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
P, K, S, H, W = 190, 19, 1, 1024, 2048
ident = _identity(P, K, S)
lay = spx.BankLayout(P, K, S, 256, ((0, P),))
keys, J, table = spx.class_gather_table(lay, ident, dev)
patches = torch.randint(0, K + 1, (1, H // 64, W // 64), device=dev)
target = patches.repeat_interleave(64, 1).repeat_interleave(64, 2)
vals = (torch.rand(1, J, H * W, device=dev) * 6).requires_grad_(True)
cd = spx.ClassDistances(vals, (target.reshape(1, -1) - 1).int(), table, (H, W))
loss_fn = spx.KLDLoss(ident, S, {0: (0, P)})
def step():
    vals.grad = None
    l = loss_fn(cd, target); l.backward(); return l
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=10, max_name_column_width=60))
