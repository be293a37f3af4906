"""Debug probe: fused backward vs oracle, error pattern per (prototype block, channel block)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from oracle import ppnet_oracle as O
from test_gpu_parity import _problem, _layout
from test_gpu_fused_bwd import _run

dev = torch.device("cuda:0")
shape = tuple(int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,1,256,190,19,16,64".split(",")))
B, S, Cs, P, K, H, W = shape
conv, bank, Wl, ident, ranges = _problem(*shape, seed=11)
layout = _layout(P, K, S, Cs, ranges)
g = torch.Generator().manual_seed(5)
g_logits = torch.randn(B, H, W, K, generator=g) * 1e-3
g_dist = torch.randn(B, P, H, W, generator=g) * 1e-3
for name, gl, gd in (("both", g_logits, g_dist), ("dist only", None, g_dist), ("logits only", g_logits, None)):
    _, _, _, dx_ref, dp_ref, dw_ref = O.fwd_bwd_reference(conv, bank, ranges, S, Wl, gl if gl is not None else torch.zeros_like(g_logits),
                                                          gd if gd is not None else torch.zeros_like(g_dist))
    dx, dp, dw = _run(conv, bank, Wl, layout, gl, gd, torch.float32, dev, fused=True)
    dx2, dp2, dw2 = _run(conv, bank, Wl, layout, gl, gd, torch.float32, dev, fused=False)
    dp, dp2, dpr = dp.cpu().reshape(P, Cs), dp2.cpu().reshape(P, Cs), dp_ref.reshape(P, Cs)
    sc = dpr.abs().max().item()
    print(name, "dX err", ((dx.cpu() - dx_ref).abs().max() / dx_ref.abs().max()).item(), "two-kernel dP err", ((dp2 - dpr).abs().max() / sc).item())
    for pb in range((P + 31) // 32):
        row = []
        for cb in range((Cs + 31) // 32):
            e = (dp[pb * 32:(pb + 1) * 32, cb * 32:(cb + 1) * 32] - dpr[pb * 32:(pb + 1) * 32, cb * 32:(cb + 1) * 32]).abs().max().item() / sc
            row.append(f"{e:8.1e}")
        print("  pb", pb, " ".join(row))
    if dw is not None:
        print("  dW err", ((dw.cpu() - dw_ref).abs().max() / dw_ref.abs().max()).item())
