"""Diagnostic (SPX_DIAG_STAMPS build): per-workgroup phase clocks of the backward pixel kernel."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scaleprotoseg_amd as spx
from scaleprotoseg_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
C_, P, S, K, H, W = 256, 190, 1, 19, 1024, 2048
x = torch.sigmoid(torch.randn(1, C_, H, W, device=dev)).bfloat16().requires_grad_(True)
bank = torch.rand(P, C_, 1, 1, device=dev).requires_grad_(True)
head = torch.randn(K, P, device=dev).requires_grad_(True)
lay = spx.BankLayout(P, K, S, C_, ((0, P),))
gl = torch.randn(H * W, K, device=dev) * 1e-3
gd = torch.randn(1, P, H, W, device=dev) * 1e-3
ntiles = H * W // 128
dbg = torch.zeros(ntiles * 8, dtype=torch.int64, device=dev)
for it in range(2):
    lib.spx_diag_set_debug_buffer(C.c_void_p(dbg.data_ptr()))
    logits, dist, _ = spx.proto_head_forward(x, bank, head, lay, want_distances=True)
    torch.cuda.synchronize()
    dbg.zero_()
    torch.autograd.backward([logits, dist], [gl, gd])
    torch.cuda.synchronize()
d = dbg[: ntiles * 4].view(ntiles, 4).cpu().double()
print("bwd pixel kernel per-WG ticks: main %.0f  phase1 %.0f  phase2 %.0f  total %.0f" % (
    (d[:,1]-d[:,0]).mean(), (d[:,2]-d[:,1]).mean(), (d[:,3]-d[:,2]).mean(), (d[:,3]-d[:,0]).mean()))
