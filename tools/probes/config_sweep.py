"""Probe: fwd+bwd time of the BASELINE.json configurations (SURVEY 8d shapes), distances + logits out."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import scaleprotoseg_amd as spx
from scaleprotoseg_amd.functional import proto_head_forward
dev = torch.device("cuda:0")
def run(tag, B, S, Cs, P, K, H, W):
    x = torch.sigmoid(torch.randn(B, S * Cs, H, W, device=dev)).bfloat16().requires_grad_(True)
    bank = torch.rand(P, Cs, 1, 1, device=dev).requires_grad_(True)
    head = (torch.randn(K, P, device=dev) * 0.1).requires_grad_(True)
    per = P // S
    lay = spx.BankLayout(P, K, S, Cs, tuple((s * per, (s + 1) * per) for s in range(S)))
    gl = torch.randn(B * H * W, K, device=dev) * 1e-3
    gd = torch.randn(B, P, H, W, device=dev) * 1e-3
    def step():
        x.grad = bank.grad = head.grad = None
        logits, d, _ = proto_head_forward(x, bank, head, lay)
        torch.autograd.backward([logits, d], [gl, gd])
    for _ in range(3): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{tag:38s} B={B} {H}x{W} P={P} S={S} K={K}: {ms:.3f} ms/step  {B*H*W/ms/1e3:.1f} Mpix/s")
run("EM literal", 1, 1, 64, 10, 2, 512, 512)
run("EM gin", 1, 4, 64, 24, 2, 512, 512)
run("Pascal baseline", 2, 1, 64, 210, 21, 65, 65)
run("Pascal ScaleProtoSeg", 2, 4, 64, 252, 21, 65, 65)
run("Cityscapes native", 1, 4, 64, 228, 19, 129, 257)
run("Cityscapes train crops x10", 10, 4, 64, 228, 19, 65, 65)
run("ADE literal", 2, 1, 64, 1500, 150, 65, 65)
run("ADE gin", 2, 4, 64, 1800, 150, 65, 65)
run("ADE gin, 512x512 latent", 1, 4, 64, 1800, 150, 512, 512)
