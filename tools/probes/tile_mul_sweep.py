"""Probe (needs a -DSPX_DIAG build via SPX_LIB_OVERRIDE): forward / backward time of the north-star step over block -> tile multipliers."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import scaleprotoseg_amd as spx
from scaleprotoseg_amd import _lib
lib = _lib.load()
lib.spx_diag_set_tile_mul.argtypes = [C.c_int]; lib.spx_diag_set_tile_mul.restype = None
dev = torch.device("cuda:0")
Cc, P, K, H, W = 256, 190, 19, 1024, 2048
g = torch.Generator(device=dev).manual_seed(1)
x = torch.sigmoid(torch.randn(1, Cc, H, W, device=dev, generator=g)).bfloat16().requires_grad_(True)
bank = torch.rand(P, Cc, 1, 1, device=dev, generator=g).requires_grad_(True)
head = (torch.randn(K, P, device=dev, generator=g) * 0.1).requires_grad_(True)
lay = spx.BankLayout(P, K, 1, Cc, ((0, P),))
gl = torch.randn(H * W, K, device=dev, generator=g) * 1e-3
gd = torch.randn(1, P, H, W, device=dev, generator=g) * 1e-3
def step():
    x.grad = bank.grad = head.grad = None
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    logits, d, _ = spx.proto_head_forward(x, bank, head, lay)
    e[1].record()
    torch.autograd.backward([logits, d], [gl, gd])
    e[2].record()
    return e
for mul in (0, 1, 33, 65, 129, 257, 513, 1025, 2049, 4097, 0):
    lib.spx_diag_set_tile_mul(mul)
    for _ in range(3): step()
    fs = bs = 0.0
    for _ in range(10):
        e = step(); torch.cuda.synchronize()
        fs += e[0].elapsed_time(e[1]); bs += e[1].elapsed_time(e[2])
    print(f"mul {mul:5d}: fwd {fs/10:.3f} ms  bwd {bs/10:.3f} ms", flush=True)
