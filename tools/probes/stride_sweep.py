"""Probe: does the power-of-two plane stride of the north-star shape (H*W*2 B = 4 MiB) cost anything?  fwd+bwd time per
pixel for neighbouring grid sizes (all multiples of 8 pixels: the same vector staging path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import scaleprotoseg_amd as spx
from scaleprotoseg_amd import functional as F_
from scaleprotoseg_amd.functional import proto_head_forward

dev = torch.device("cuda:0")
from scaleprotoseg_amd import _lib
MUL = int(sys.argv[1]) if len(sys.argv) > 1 else 0      # 0 = the library's automatic choice, 1 = identity
_lib.load().spx_diag_set_tile_mul(MUL)      # needs a library built with -DSPX_DIAG (tools/build_variant.sh)
print('tile multiplier', MUL)
C_, P, K = 256, 190, 19
lay = spx.BankLayout(P, K, 1, C_, ((0, P),))
bank = torch.rand(P, C_, 1, 1, device=dev).requires_grad_(True)
head = (torch.randn(K, P, device=dev) * 0.1).requires_grad_(True)
for H, W in ((1024, 2048), (1024, 2040), (1016, 2048), (1008, 2048), (1000, 2000), (960, 2048), (1024, 2304)):
    x = torch.sigmoid(torch.randn(1, C_, H, W, device=dev)).bfloat16().requires_grad_(True)
    gl = torch.randn(H * W, K, device=dev) * 1e-3
    gd = torch.randn(1, P, H, W, device=dev) * 1e-3

    def step():
        x.grad = bank.grad = head.grad = None
        logits, d, _ = proto_head_forward(x, bank, head, lay)
        torch.autograd.backward([logits, d], [gl, gd])

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    sink = []
    F_.set_profile(sink)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    F_.set_profile(None)
    per = {}
    for name, a, b in sink:
        per.setdefault(name, []).append(a.elapsed_time(b))
    norm = 2097152 / (H * W)
    parts = "  ".join(f"{k[4:]} {sum(v) / len(v) * norm:.3f}" for k, v in per.items())
    print(f"{H}x{W}: plane stride {H * W * 2:#x} B  step {ms * norm:.3f} ms per 2 Mpx | {parts}")
    del x, gl, gd
