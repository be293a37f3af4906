"""Diagnostic (SPX_DIAG_STAMPS build): phase clocks of the fused persistent backward, cycles per workgroup and tile."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scaleprotoseg_amd as spx
from scaleprotoseg_amd import _lib
if os.environ.get('SPX_LIB_OVERRIDE'):
    _lib.LIB_PATH = os.path.abspath(os.environ['SPX_LIB_OVERRIDE'])
lib = _lib.load()
dev = torch.device("cuda:0")
C_, P, S, K, H, W = 256, 190, 1, 19, 1024, 2048
x = torch.sigmoid(torch.randn(1, C_, H, W, device=dev)).bfloat16().requires_grad_(True)
bank = torch.rand(P, C_, 1, 1, device=dev).requires_grad_(True)
head = torch.randn(K, P, device=dev).requires_grad_(True)
lay = spx.BankLayout(P, K, S, C_, ((0, P),))
gl = torch.randn(H * W, K, device=dev) * 1e-3
gd = torch.randn(1, P, H, W, device=dev) * 1e-3
nwg = 256
dbg = torch.zeros(nwg * 8 * 10 + 1024, dtype=torch.int64, device=dev)
for it in range(3):
    lib.spx_diag_set_debug_buffer(C.c_void_p(dbg.data_ptr()))
    logits, dist, _ = spx.proto_head_forward(x, bank, head, lay, want_distances=True)
    torch.cuda.synchronize()
    dbg.zero_()
    torch.autograd.backward([logits, dist], [gl, gd])
    torch.cuda.synchronize()
d = dbg[: nwg * 8 * 10].view(nwg, 8, 10).cpu().double()
tiles = H * W // 128 / nwg
names = ["A prologue+main", "B1 G/a", "B1 barrier", "B2 pack", "C dX", "D bank", "| main loop: issue", "compute (incl. issue)", "LDS write", "barrier"]
m = d.mean(dim=(0, 1)) / tiles
print("fused backward, cycles per tile (mean over waves):", "  ".join(f"{n} {v:.0f}" for n, v in zip(names, m.tolist())), f"  total {m[:6].sum().item():.0f}")
for w in range(8):
    print("  wave", w, "  ".join(f"{v:.0f}" for v in (d[:, w].mean(0) / tiles).tolist()))
