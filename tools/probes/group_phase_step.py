"""Scratch: group-phase step split: fused kernels vs the torch tail (exp + last_layer_group) around them."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
import scaleprotoseg_amd as spx
from scaleprotoseg_amd.functional import proto_head_forward
dev = torch.device("cuda:0")
P, S, Cs, U, K, B, H, W = 228, 4, 64, 57, 19, 1, 1024, 2048
x = torch.sigmoid(torch.randn(B, S * Cs, H, W, device=dev)).bfloat16().requires_grad_(True)
bank = torch.rand(P, Cs, 1, 1, device=dev).requires_grad_(True)
wd = (torch.rand(U, P, device=dev) * 0.05).requires_grad_(True)
wg = torch.randn(K, U, device=dev).requires_grad_(True)
per = P // S
lay = spx.BankLayout(P, U, S, Cs, tuple((s * per, (s + 1) * per) for s in range(S)))
gl = torch.randn(B * H * W, K, device=dev) * 1e-3
def step(tail):
    x.grad = bank.grad = wd.grad = wg.grad = None
    if tail == "fused":
        logits, _, _, _ = proto_head_forward(x, bank, wd, lay, want_distances=False, group_tail=wg)
        torch.autograd.backward([logits], [gl])
        return
    gpre, _, _ = proto_head_forward(x, bank, wd, lay, want_distances=False)
    if tail:
        logits = F.linear(torch.exp(gpre), wg)
        torch.autograd.backward([logits], [gl])
    else:
        torch.autograd.backward([gpre], [torch.ones_like(gpre) * 1e-3])
for tail in (False, True, "fused"):
    for _ in range(3): step(tail)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): step(tail)
    e1.record(); torch.cuda.synchronize()
    print(f"tail={tail}: {e0.elapsed_time(e1)/10:.3f} ms/step")
