"""Probe: profile of one gathered-mode step with KLDLoss on the real distances."""
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
C, P, K, S, H, W = 256, 190, 19, 1, 1024, 2048
ident = _identity(P, K, S).to(dev)
lay = spx.BankLayout(P, K, S, C, ((0, P),))
keys, J, table = spx.class_gather_table(lay, ident, dev)
x = torch.sigmoid(torch.randn(1, C, H, W, device=dev)).bfloat16().requires_grad_(True)
bank = torch.rand(P, C, 1, 1, device=dev).requires_grad_(True)
head = torch.randn(K, P, device=dev).requires_grad_(True)
patches = torch.randint(0, K, (1, H // 64, W // 64), device=dev, dtype=torch.int32)
labels0 = patches.repeat_interleave(64, 1).repeat_interleave(64, 2).reshape(1, H * W).contiguous()
gather = spx.ClassGather(labels=labels0, keys=keys, width=J, table=table)
gl = torch.randn(H * W, K, device=dev) * 1e-3
kld = spx.KLDLoss(ident, S, {0: (0, P)})
target1 = (labels0 + 1).reshape(1, H, W)
def step():
    x.grad = bank.grad = head.grad = None
    logits, dmap, _ = spx.proto_head_forward(x, bank, head, lay, class_gather=gather)
    loss = kld(spx.ClassDistances(dmap, gather.labels, gather.table, (H, W)), target1)
    torch.autograd.backward([logits, loss], [gl, None])
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60))
