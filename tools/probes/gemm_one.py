"""Probe: the three products of ONE wide-head shape a few times each (for rocprofv3 passes). usage: gemm_one.py [M N K]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from scaleprotoseg_amd import functional as F_
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8450, 450, 1800)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
a = torch.randn(M, K, device=dev, generator=g); w = torch.randn(N, K, device=dev, generator=g); go = torch.randn(M, N, device=dev, generator=g)
for _ in range(5):
    F_._rows_gemm(a, (K, 1), w, (K, 1), M, N, K)
    F_._rows_gemm(go, (N, 1), w, (1, K), M, K, N)
    F_._rows_gemm(go, (1, N), a, (1, K), N, K, M)
torch.cuda.synchronize()
