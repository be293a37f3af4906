"""Probe: host-side cost of one eager fwd+bwd at the reference's launch-bound training shapes (wall clock per step
vs the GPU's busy time) and where the Python time goes (cProfile, top cumulative entries)."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import scaleprotoseg_amd as spx
from scaleprotoseg_amd.functional import proto_head_forward
dev = torch.device("cuda:0")
def make(B, S, Cs, P, K, H, W):
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
    return step
for tag, cfg in (("Cityscapes train crops x10", (10, 4, 64, 228, 19, 65, 65)), ("Pascal ScaleProtoSeg", (2, 4, 64, 252, 21, 65, 65))):
    step = make(*cfg)
    for _ in range(20): step()
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n): step()
    t_cpu = time.perf_counter() - t0          # host time to ENQUEUE n steps
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{tag}: host enqueue {t_cpu / n * 1e3:.3f} ms/step, wall {t_all / n * 1e3:.3f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100): step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:40]))
