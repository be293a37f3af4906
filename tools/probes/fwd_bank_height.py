"""Scratch: forward time vs bank height (register footprint / occupancy probe)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import scaleprotoseg_amd as spx
from scaleprotoseg_amd.functional import proto_head_forward
dev = torch.device("cuda:0")
Cs, K, H, W = 256, 19, 1024, 2048
x = torch.sigmoid(torch.randn(1, Cs, H, W, device=dev)).bfloat16()
for P in (57, 64, 114, 128, 190):
    bank = torch.rand(P, Cs, 1, 1, device=dev)
    Wl = torch.randn(K, P, device=dev)
    lay = spx.BankLayout(P, K, 1, Cs, ((0, P),))
    for want_d in (True, False):
        for _ in range(3): proto_head_forward(x, bank, Wl, lay, want_distances=want_d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): proto_head_forward(x, bank, Wl, lay, want_distances=want_d)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        by = (512 + (4 * P if want_d else 0) + 76) * H * W
        print(f"P={P:4d} dist={want_d}: {ms:.3f} ms  {by/ms/1e9:.2f} TB/s algorithmic (floor at 5.6 TB/s {by/5.6e9:.3f} ms)")
