"""One prototype-phase training iteration of the path at the reference's crop shape (10 x 65 x 65 latent, P = 228, S = 4,
K = 19), through the modules: forward_from_conv_features with the class-gathered distances and the fused cross entropy, the
KLD loss, backward, Adam step on the prototypes + last layer.  Irregular label regions.  Eager ms per iteration, the host's
share (cProfile), and - under rocprofv3 - the kernels.   python tools/probes/train_step_real.py [profile]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn as nn

import scaleprotoseg_amd as spx
from scaleprotoseg_amd.loss import KLDLoss, PixelWiseCrossEntropyLoss


class _Backbone(nn.Module):
    """Stand-in for the DeepLab backbone (out of scope): str() starts with MSC and .base holds two Conv2d."""

    def __init__(self, c):
        super().__init__()
        self.base = nn.Sequential(nn.Conv2d(3, c, 1), nn.Conv2d(c, c, 1))

    def __repr__(self):
        return "MSC(standin)"

    def forward(self, x):
        return x


def main():
    dev = torch.device("cuda:0")
    B, S, Cs, K, H, W = 10, 4, 64, 19, 65, 65
    P = 228
    torch.manual_seed(0)
    net = spx.PPNetMultiScale(_Backbone(S * Cs), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                              patch_classification=True, num_scales=S).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.sigmoid(torch.randn(B, S * Cs, H, W, device=dev, generator=g)).to(torch.bfloat16)
    coarse = torch.randn(B, K + 1, 5, 5, device=dev, generator=g)
    target = torch.nn.functional.interpolate(coarse, size=(H, W), mode="bicubic", align_corners=False).argmax(dim=1)
    ce = PixelWiseCrossEntropyLoss(ignore_index=-1)
    kld = KLDLoss(net.prototype_class_identity, S, net.scale_num_prototypes)
    opt = torch.optim.Adam([net.prototype_vectors, net.last_layer.weight], lr=1e-3)

    def step():
        opt.zero_grad(set_to_none=True)
        xin = x.detach().requires_grad_(True)
        logits, cd = net.forward_from_conv_features(xin, target_labels=target, ce_target=target)
        loss = ce(logits, target) + 0.25 * kld(cd, target)
        loss.backward()
        opt.step()
        return loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        loss = step()
    t_host = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n * 1e3
    print(f"training iteration 10x65x65 P=228 S=4: {t_all:.3f} ms per iteration (host issue time {t_host:.3f} ms), loss {loss.item():.4f}", flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "profile":
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(100):
            step()
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
