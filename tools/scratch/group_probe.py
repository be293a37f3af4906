"""Scratch: fwd+bwd time of the group-phase head shape (dense [G*K', P] head, 57 rows) vs the 19-row head."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import scaleprotoseg_amd as spx
from scaleprotoseg_amd.functional import proto_head_forward
dev = torch.device("cuda:0")
def run(P, S, Cs, K, B, H, W, tag):
    x = torch.sigmoid(torch.randn(B, S * Cs, H, W, device=dev)).bfloat16().requires_grad_(True)
    bank = torch.rand(P, Cs, 1, 1, device=dev).requires_grad_(True)
    head = torch.randn(K, P, device=dev).requires_grad_(True)
    per = P // S
    lay = spx.BankLayout(P, K, S, Cs, tuple((s * per, (s + 1) * per) for s in range(S)))
    gl = torch.randn(B * H * W, K, device=dev) * 1e-3
    def step():
        x.grad = bank.grad = head.grad = None
        logits, _, _ = proto_head_forward(x, bank, head, lay, want_distances=False)
        torch.autograd.backward([logits], [gl])
    for _ in range(3): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): step()
    e1.record(); torch.cuda.synchronize()
    print(f"{tag}: P={P} S={S} K={K} B={B} {H}x{W}: {e0.elapsed_time(e1)/10:.3f} ms/step")
run(228, 4, 64, 19, 1, 1024, 2048, "proto phase, big")
run(228, 4, 64, 57, 1, 1024, 2048, "group phase, big")
run(228, 4, 64, 19, 8, 129, 257, "proto phase, native x8")
run(228, 4, 64, 57, 8, 129, 257, "group phase, native x8")
