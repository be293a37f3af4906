"""Split sweep of the bf16x3 product kernel (diagnostic build: spx_diag_set_gemm forces kernel 3 = bf16x3, 1/2 = fp32 pipe).
SPX_LIB_OVERRIDE=<diag lib> python tools/probes/gemm3_sweep.py"""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from scaleprotoseg_amd import _lib, functional as F_

def timed(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

dev = torch.device("cuda:0")
lib = _lib.load()
lib.spx_diag_set_gemm.argtypes = [C.c_int, C.c_int]; lib.spx_diag_set_gemm.restype = None
g = torch.Generator(device=dev).manual_seed(1)
for name, M, N, K in (("ade units", 8450, 450, 1800), ("coco classes", 8450, 182, 2184), ("ade tail", 8450, 150, 450), ("coco units", 16900, 546, 2184)):
    a = torch.randn(M, K, device=dev, generator=g); w = torch.randn(N, K, device=dev, generator=g); go = torch.randn(M, N, device=dev, generator=g)
    cases = (("y", lambda: F_._rows_gemm(a, (K, 1), w, (K, 1), M, N, K)),
             ("d_a", lambda: F_._rows_gemm(go, (N, 1), w, (1, K), M, K, N)),
             ("d_w", lambda: F_._rows_gemm(go, (1, N), a, (1, K), N, K, M)))
    for what, fn in cases:
        row = []
        for wm, sps in ((3, (1, 2, 3, 4, 5, 6, 8, 12, 16)), (1, (1, 2, 4, 8))):
            for sp in sps:
                lib.spx_diag_set_gemm(wm, sp)
                row.append(f"k{wm}/s{sp}: {timed(fn):6.1f}")
        lib.spx_diag_set_gemm(0, 0)
        row.append(f"auto: {timed(fn):6.1f}")
        print(f"{name} {what}: " + "  ".join(row), flush=True)
