"""Driver for profiling the kernels beside the distance path at their documented shapes: the push argmin
(190 x 1024 x 2048 map), the KLD loss kernels (class-gathered planes of a 2 Mpx image, forward + backward), the
class-gathered forward that feeds them and the fused eval-time upsample + argmin (228 x 129 x 257 -> 1024 x 2048).
    rocprofv3 --kernel-trace --stats -- python3 tools/probes/aux_kernels.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import scaleprotoseg_amd as spx

dev = torch.device("cuda:0")
REPS = 5
g = torch.Generator(device=dev).manual_seed(1)
# push argmin
P, K, H, W = 190, 19, 1024, 2048
d = torch.rand(1, P, H, W, device=dev, generator=g) * 50
lab = torch.randint(0, K + 1, (1, H // 64, W // 64), device=dev, generator=g).repeat_interleave(64, 1).repeat_interleave(64, 2)
ident = torch.zeros(P, K, device=dev)
for p in range(P):
    ident[p, p // (P // K)] = 1
for _ in range(REPS):
    spx.push_masked_argmin(d, lab, ident, void_class=0)
del d
# KLD on the class-gathered planes of one 2 Mpx image
J = 10
vals = (torch.rand(1, J, H * W, device=dev, generator=g) * 20).requires_grad_(True)
labels0 = (lab.reshape(1, -1) - 1).to(torch.int32)
table = torch.arange(K * J, device=dev).reshape(K, J)
kld = spx.KLDLoss(ident, 1, {0: (0, P)})
for _ in range(REPS):
    vals.grad = None
    kld(spx.ClassDistances(vals, labels0, table, (H, W)), lab).backward()
# eval-time map
src = torch.rand(1, 228, 129, 257, device=dev, generator=g) * 10
for _ in range(REPS):
    spx.upsample_argext(src, (1024, 2048))
    spx.upsample_argext(src[:, :19], (1024, 2048), largest=True)
torch.cuda.synchronize()
print("aux kernels done")
