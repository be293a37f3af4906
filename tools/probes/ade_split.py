import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import scaleprotoseg_amd as spx
from scaleprotoseg_amd import functional as F_
dev = torch.device("cuda:0")
B, S, Cs, P, K, H, W = 1, 4, 64, 1800, 150, 512, 512
x = torch.sigmoid(torch.randn(B, S * Cs, H, W, device=dev)).bfloat16().requires_grad_(True)
bank = torch.rand(P, Cs, 1, 1, device=dev).requires_grad_(True)
head = (torch.randn(K, P, device=dev) * 0.1).requires_grad_(True)
per = P // S
lay = spx.BankLayout(P, K, S, Cs, tuple((s * per, (s + 1) * per) for s in range(S)))
gl = torch.randn(B * H * W, K, device=dev) * 1e-3
gd = torch.randn(B, P, H, W, device=dev) * 1e-3
def step():
    x.grad = bank.grad = head.grad = None
    logits, d, _ = F_.proto_head_forward(x, bank, head, lay)
    torch.autograd.backward([logits, d], [gl, gd])
for _ in range(2): step()
torch.cuda.synchronize()
prof = []
F_.set_profile(prof)
for _ in range(5): step()
torch.cuda.synchronize()
F_.set_profile(None)
per_op = {}
for n, e0, e1 in prof: per_op.setdefault(n, []).append(e0.elapsed_time(e1))
print({k: round(sum(v) / len(v), 3) for k, v in per_op.items()})
