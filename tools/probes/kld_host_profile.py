"""Host-side profile (cProfile) of the eager KLDLoss step on training-crop sized inputs: where the Python time goes."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import scaleprotoseg_amd as spx
from kld_loss_time import _identity

dev = torch.device("cuda:0")
B, P, K, S, H, W = 10, 228, 19, 4, 65, 65
ident = _identity(P, K, S)
per = P // S
lay = spx.BankLayout(P, K, S, 64, tuple((s * per, (s + 1) * per) for s in range(S)))
keys, J, table = spx.class_gather_table(lay, ident, dev)
patches = torch.randint(0, K + 1, (B, 5, 5), device=dev)
target = patches.repeat_interleave(16, 1).repeat_interleave(16, 2)[:, :H, :W].contiguous()
vals = (torch.rand(B, J, H * W, device=dev) * 6).requires_grad_(True)
cd = spx.ClassDistances(vals, (target.reshape(B, -1) - 1).int(), table, (H, W))
loss_fn = spx.KLDLoss(ident, S, {s: lay.scale_ranges[s] for s in range(S)})


def step():
    vals.grad = None
    loss = loss_fn(cd, target)
    loss.backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
