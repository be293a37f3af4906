"""Times of the fp32 MFMA product kernels (spx_rows_gemm) on the wide-head shapes, next to torch.mm (rocBLAS) on the same
operands - the library is only the yardstick here, the product path does not call it.
python tools/probes/gemm_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from scaleprotoseg_amd import functional as F_  # noqa: E402


def timed(fn, n=20, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    for name, M, N, K in (("ade group units 2x65x65", 8450, 450, 1800), ("coco classes 2x65x65", 8450, 182, 2184),
                          ("ade tail 2x65x65", 8450, 150, 450), ("coco group units 4x65x65", 16900, 546, 2184)):
        a = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g)
        go = torch.randn(M, N, device=dev, generator=g)
        fl = 2.0 * M * N * K
        rows = []
        for what, ours, lib in (
            ("y = a.w^T", lambda: F_._rows_gemm(a, (K, 1), w, (K, 1), M, N, K), lambda: a @ w.t()),
            ("d_a = g.w", lambda: F_._rows_gemm(go, (N, 1), w, (1, K), M, K, N), lambda: go @ w),
            ("d_w = g^T.a", lambda: F_._rows_gemm(go, (1, N), a, (1, K), N, K, M), lambda: go.t() @ a),
        ):
            t1, t2 = timed(ours), timed(lib)
            rows.append(f"{what}: {t1 * 1e3:7.1f} us = {fl / t1 / 1e9:6.1f} TFLOP/s (torch.mm {t2 * 1e3:7.1f} us = {fl / t2 / 1e9:6.1f})")
        print(f"{name}  M={M} N={N} K={K}\n  " + "\n  ".join(rows), flush=True)


if __name__ == "__main__":
    main()
