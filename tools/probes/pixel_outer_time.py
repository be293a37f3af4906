import torch, time
dev = torch.device("cuda:0")
M, K2, U = 2097152, 19, 57
gl = torch.randn(M, K2, device=dev); g = torch.rand(M, U, device=dev)
def t(f, name):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): r = f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1)/5:.3f} ms")
    return r
a = t(lambda: gl.t() @ g, "gl.t() @ g")
def bmm():
    c = 4096
    return torch.bmm(gl.view(M // c, c, K2).transpose(1, 2), g.view(M // c, c, U)).sum(0)
b = t(bmm, "chunked bmm + sum")
print((a - b).abs().max().item(), a.abs().max().item())
wg = torch.randn(K2, U, device=dev)
t(lambda: torch.nn.functional.linear(g, wg), "linear fwd")
t(lambda: torch.exp(g), "exp")
t(lambda: gl @ wg, "dG = gl @ wg")
