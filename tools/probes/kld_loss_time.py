"""KLDLoss forward + backward on class-gathered planes, eager: the 2 Mpx north-star grid and the reference's training crops.
python tools/probes/kld_loss_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import scaleprotoseg_amd as spx


def _identity(P, K, S):
    ident = torch.zeros(P, K)
    per_scale, per_cs = P // S, P // K // S
    for s in range(S):
        for k in range(K):
            ident[s * per_scale + k * per_cs: s * per_scale + (k + 1) * per_cs, k] = 1
    return ident


def run(name, B, P, K, S, H, W, patch):
    dev = torch.device("cuda:0")
    ident = _identity(P, K, S)
    per = P // S
    lay = spx.BankLayout(P, K, S, 64, tuple((s * per, (s + 1) * per) for s in range(S)))
    keys, J, table = spx.class_gather_table(lay, ident, dev)
    if patch > 0:
        patches = torch.randint(0, K + 1, (B, -(-H // patch), -(-W // patch)), device=dev)
        target = patches.repeat_interleave(patch, 1).repeat_interleave(patch, 2)[:, :H, :W].contiguous()
    else:
        # irregular regions with curved boundaries: argmax over K + 1 smooth random fields (a coarse field, upsampled)
        g = torch.Generator(device=dev).manual_seed(3)
        coarse = torch.randn(B, K + 1, max(2, H // -patch), max(2, W // -patch), device=dev, generator=g)
        target = torch.nn.functional.interpolate(coarse, size=(H, W), mode="bicubic", align_corners=False).argmax(dim=1)
        mixed = 0
        for r0 in range(0, H - 3, 4):
            blk = target[:, r0:r0 + 4, : (W // 16) * 16].reshape(B, 4, W // 16, 16)
            mixed += int((blk.amax(dim=(1, 3)) != blk.amin(dim=(1, 3))).sum())
        print(f"  ({name}: {mixed / max(1, B * (H // 4) * (W // 16)):.2f} of the 16x4 blocks hold more than one class)")
    vals = (torch.rand(B, J, H * W, device=dev) * 6).requires_grad_(True)
    cd = spx.ClassDistances(vals, (target.reshape(B, -1) - 1).int(), table, (H, W))
    loss_fn = spx.KLDLoss(ident, S, {s: lay.scale_ranges[s] for s in range(S)})

    def step():
        vals.grad = None
        loss = loss_fn(cd, target)
        loss.backward()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        loss = step()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: KLDLoss fwd+bwd on gathered [{B},{J},{H * W}]: {e0.elapsed_time(e1) / n:.3f} ms, loss {loss.item():.5f}", flush=True)


if __name__ == "__main__":
    cases = [("north star 1024x2048 P=190 S=1", 1, 190, 19, 1, 1024, 2048, 64),
             ("cityscapes crops 10x65x65 P=228 S=4", 10, 228, 19, 4, 65, 65, 16),
             ("native 129x257 P=228 S=4", 1, 228, 19, 4, 129, 257, 16),
             ("north star, 16-px label patches", 1, 190, 19, 1, 1024, 2048, 16),
             ("north star, irregular regions ~24 px", 1, 190, 19, 1, 1024, 2048, -24),
             ("cityscapes crops 10x65x65, irregular regions ~12 px", 10, 228, 19, 4, 65, 65, -12)]
    sel = [int(a) for a in sys.argv[1:]] or range(len(cases))
    for i in sel:
        run(*cases[i])
