"""Debug probe: one-hot distance gradient -> where does the fused d_bank put it?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import ppnet_oracle as O
from test_gpu_parity import _problem, _layout
from test_gpu_fused_bwd import _run

dev = torch.device("cuda:0")
shape = (1, 1, 256, 190, 19, 1, 128)
B, S, Cs, P, K, H, W = shape
conv, bank, Wl, ident, ranges = _problem(*shape, seed=11)
layout = _layout(P, K, S, Cs, ranges)
for (p_, px_) in ((0, 0), (1, 0), (5, 3), (37, 70), (100, 127), (33, 17)):
    g_dist = torch.zeros(B, P, H, W)
    g_dist[0, p_, 0, px_] = 1.0
    _, _, _, dx_ref, dp_ref, _ = O.fwd_bwd_reference(conv, bank, ranges, S, Wl, torch.zeros(B, H, W, K), g_dist)
    dx, dp, _ = _run(conv, bank, Wl, layout, None, g_dist, torch.float32, dev, fused=True, freeze=("head",))
    dp, dpr = dp.cpu().reshape(P, Cs), dp_ref.reshape(P, Cs)
    rows = (dp.abs().sum(1) > 0).nonzero().flatten().tolist()
    print(f"p={p_} px={px_}: ref rows {(dpr.abs().sum(1) > 0).nonzero().flatten().tolist()} got rows {rows}")
    x = conv[0, :, 0, :]          # [C, W]
    for r_ in rows[:4]:
        # which pixel's x explains the row?  dp[r] = 2 g (p[r] - x[:, px])  -> x_est = p[r] - dp[r] / (2 g)
        gval = 1.0
        xest = bank[r_, :, 0, 0] - dp[r_] / 2.0
        d = (x - xest[:, None]).abs().max(0).values
        print(f"   row {r_}: best px {d.argmin().item()} (err {d.min().item():.2e}); ref err {(dp[r_] - dpr[r_]).abs().max().item():.2e}; first vals got {dp[r_, :3].tolist()} ref {dpr[r_, :3].tolist()}")
