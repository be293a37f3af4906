"""Probe: class-masked push argmin over a full-size distance map (spx_push_argmin), achieved HBM rate."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import scaleprotoseg_amd as spx
dev = torch.device("cuda:0")
for (B, P, K, H, W) in ((1, 190, 19, 1024, 2048), (10, 228, 19, 65, 65), (1, 228, 19, 129, 257)):
    d = torch.rand(B, P, H, W, device=dev) * 50
    lab = torch.randint(0, K + 1, (B, H, W), device=dev)
    ident = torch.zeros(P, K, device=dev)
    for p in range(P):
        ident[p, p % K] = 1
    for _ in range(3): spx.push_masked_argmin(d, lab, ident)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): spx.push_masked_argmin(d, lab, ident)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"push argmin B={B} P={P} {H}x{W}: {ms:.3f} ms  ({d.numel() * 4 / ms / 1e9:.2f} TB/s on the map bytes)")
