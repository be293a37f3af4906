"""Probe: cost of the [pixel][P] activation output (and of its gradient input) on the 4-scale bank."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import scaleprotoseg_amd as spx
from scaleprotoseg_amd import functional as F_
dev = torch.device("cuda:0")
B, S, Cs, P, K, H, W = 1, 4, 64, 228, 19, 1024, 2048
x = torch.sigmoid(torch.randn(B, S * Cs, H, W, device=dev)).bfloat16().requires_grad_(True)
bank = torch.rand(P, Cs, 1, 1, device=dev).requires_grad_(True)
head = (torch.randn(K, P, device=dev) * 0.1).requires_grad_(True)
per = P // S
lay = spx.BankLayout(P, K, S, Cs, tuple((s * per, (s + 1) * per) for s in range(S)))
gl = torch.randn(B * H * W, K, device=dev) * 1e-3
gd = torch.randn(B, P, H, W, device=dev) * 1e-3
ga = torch.randn(B * H * W, P, device=dev) * 1e-3
for want_act in (False, True):
    def step():
        x.grad = bank.grad = head.grad = None
        logits, d, act = F_.proto_head_forward(x, bank, head, lay, want_activations=want_act)
        outs, gs = [logits, d], [gl, gd]
        if want_act:
            outs.append(act); gs.append(ga)
        torch.autograd.backward(outs, gs)
    for _ in range(2): step()
    torch.cuda.synchronize()
    prof = []
    F_.set_profile(prof)
    for _ in range(5): step()
    torch.cuda.synchronize()
    F_.set_profile(None)
    per_op = {}
    for n, e0, e1 in prof: per_op.setdefault(n, []).append(e0.elapsed_time(e1))
    print("activations out + dAct in" if want_act else "no activations", {k: round(sum(v) / len(v), 3) for k, v in per_op.items()})
