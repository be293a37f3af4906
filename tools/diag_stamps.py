"""Diagnostic (SPX_DIAG_STAMPS build): per-workgroup phase clocks of the forward kernel."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scaleprotoseg_amd as spx
from scaleprotoseg_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
C_, P, S, K, H, W = 256, 190, 1, 19, 1024, 2048
x = torch.sigmoid(torch.randn(1, C_, H, W, device=dev)).bfloat16()
bank = torch.rand(P, C_, 1, 1, device=dev)
head = torch.randn(K, P, device=dev)
lay = spx.BankLayout(P, K, S, C_, ((0, P),))
ntiles = H * W // 128
dbg = torch.zeros(ntiles * 8, dtype=torch.int64, device=dev)
want_d = "nodist" not in sys.argv
for it in range(3):
    lib.spx_diag_set_debug_buffer(C.c_void_p(dbg.data_ptr()))
    spx.proto_head_forward(x, bank, head, lay, want_distances=want_d)
    torch.cuda.synchronize()
d = dbg[: ntiles * 4].view(ntiles, 4).cpu().double()
e = dbg[ntiles * 4 :].view(ntiles, 4).cpu().double()
print("main loop split per WG (thread 0): issue+compute %.0f  wait+lds-write %.0f  barrier %.0f  (load issue alone %.0f)" % (e[:,0].mean(), e[:,1].mean(), e[:,2].mean(), e[:,3].mean()))
main = (d[:, 1] - d[:, 0]); epi = (d[:, 2] - d[:, 1]); tail = (d[:, 3] - d[:, 2]); tot = d[:, 3] - d[:, 0]
span = d[:, 3].max() - d[:, 0].min()
print("per-WG ticks: main loop %.0f  epilogue %.0f  logits %.0f  total %.0f ; kernel span %.0f ticks" % (main.mean(), epi.mean(), tail.mean(), tot.mean(), span))
print("fractions: main %.2f epi %.2f tail %.2f" % (main.mean()/tot.mean(), epi.mean()/tot.mean(), tail.mean()/tot.mean()))
print("tiles*total/span = concurrency %.1f WGs" % (tot.sum()/span))
