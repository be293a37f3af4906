"""Probe: the fwd+bwd sequence under torch.cuda.CUDAGraph (= hipGraph) at a reference-native shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import scaleprotoseg_amd as spx
from scaleprotoseg_amd.functional import proto_head_forward
dev = torch.device("cuda:0")
import ast
B, S, Cs, P, K, H, W = ast.literal_eval(sys.argv[1]) if len(sys.argv) > 1 else (10, 4, 64, 228, 19, 65, 65)
x = torch.sigmoid(torch.randn(B, S * Cs, H, W, device=dev)).bfloat16().requires_grad_(True)
bank = torch.rand(P, Cs, 1, 1, device=dev).requires_grad_(True)
head = (torch.randn(K, P, device=dev) * 0.1).requires_grad_(True)
per = P // S
lay = spx.BankLayout(P, K, S, Cs, tuple((s * per, (s + 1) * per) for s in range(S)))
gl = torch.randn(B * H * W, K, device=dev) * 1e-3
gd = torch.randn(B, P, H, W, device=dev) * 1e-3
def step():
    logits, d, _ = proto_head_forward(x, bank, head, lay)
    return torch.autograd.grad([logits, d], [x, bank, head], [gl, gd])
def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
eager = bench(step)
ref = [g.clone() for g in step()]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    outs = step()
graphed = bench(g.replay)
g.replay(); torch.cuda.synchronize()
print(f"eager {eager:.3f} ms/step, graph replay {graphed:.3f} ms/step; identical results: {all(torch.equal(a, b) for a, b in zip(outs, ref))}")
