"""Tile / split sweep of spx_rows_gemm on one shape and operand layout (experiments: spx_diag_set_gemm forces the choice).
python tools/probes/gemm_sweep.py"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from scaleprotoseg_amd import _lib  # noqa: E402
from scaleprotoseg_amd import functional as F_  # noqa: E402


def timed(fn, n=20, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    lib.spx_diag_set_gemm.argtypes = [C.c_int, C.c_int]
    lib.spx_diag_set_gemm.restype = None
    g = torch.Generator(device=dev).manual_seed(1)
    for name, M, N, K in (("ade units", 8450, 450, 1800), ("ade units, N = 448", 8450, 448, 1800), ("coco classes", 8450, 182, 2184)):
        a = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g)
        go = torch.randn(M, N, device=dev, generator=g)
        cases = (("y", lambda: F_._rows_gemm(a, (K, 1), w, (K, 1), M, N, K)),
                 ("d_a", lambda: F_._rows_gemm(go, (N, 1), w, (1, K), M, K, N)),
                 ("d_w", lambda: F_._rows_gemm(go, (1, N), a, (1, K), N, K, M)))
        for what, fn in cases:
            row = []
            for wm in (2, 1):
                for sp in (1, 2, 4, 8, 16, 32):
                    if what != "d_w" and sp > 4:
                        continue
                    lib.spx_diag_set_gemm(wm, sp)
                    row.append(f"{64 * wm}/s{sp}: {timed(fn):6.1f}")
            lib.spx_diag_set_gemm(0, 0)
            row.append(f"auto: {timed(fn):6.1f}")
            for wm, sp in ((1, -1), (1, -2), (1, -4), (1, -16)):
                if what != "d_w" and sp < -4:
                    continue
                lib.spx_diag_set_gemm(wm, sp)
                row.append(f"linear 64/s{-sp}: {timed(fn):6.1f}")
            lib.spx_diag_set_gemm(0, 0)
            print(f"{name} {what}: " + "  ".join(row), flush=True)


if __name__ == "__main__":
    main()
