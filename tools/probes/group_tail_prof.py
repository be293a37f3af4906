import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import scaleprotoseg_amd as spx
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
P, S, Cs, K, G, H, W = 190, 1, 256, 19, 3, 1024, 2048
U = G * K
x = torch.sigmoid(torch.randn(1, Cs, H, W, device=dev)).bfloat16().requires_grad_(True)
bank = torch.rand(P, Cs, 1, 1, device=dev).requires_grad_(True)
wd = (torch.rand(U, P, device=dev) * 0.05).requires_grad_(True)
wg = torch.randn(K, U, device=dev).requires_grad_(True)
lay = spx.BankLayout(P, U, S, Cs, ((0, P),))
gl = torch.randn(H * W, K, device=dev) * 1e-3
gd = torch.randn(1, P, H, W, device=dev) * 1e-3
def step():
    x.grad = bank.grad = wd.grad = wg.grad = None
    logits, d, _, _ = spx.proto_head_forward(x, bank, wd, lay, group_tail=wg)
    torch.autograd.backward([logits, d], [gl, gd])
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=70))
