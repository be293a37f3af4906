"""GPU parity of the product kernels behind the wide heads (csrc/spx_gemm.hip, include/spx_hip.h: spx_rows_gemm): the three
products of the reference's nn.Linear (segmentation/model/model_multiscale.py:243-244 and its autograd,
model_multiscale_group.py:303-308 with the exponential) against float64 on the CPU.

Two kernels share the entry point: fp32 MFMAs (small or mostly-padding shapes) and the bf16x3 split (three bf16 planes per
fp32 operand, six bf16 MFMAs per k-step, fp32 accumulation; the dropped cross terms are <= 2^-23 of a product).  Both are held
to the SAME bound: |C - C_ref| <= 2e-6 * (|A| . |B|) element-wise - fp32 operands and accumulation up to summation order."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_parity import _dev


def _ref(A, B, flags=0, E=None):
    a, b = A.double(), B.double()
    if flags & 1:
        a = a.float().exp().double()
    if flags & 2:
        b = b.float().exp().double()
    c = a @ b.t()
    bound = a.abs() @ b.abs().t()
    if flags & 4:
        e = E.float().exp().double()
        c, bound = c * e, bound * e
    return c, bound


def _check(out, ref, bound, what):
    err = (out.double().cpu() - ref).abs()
    lim = 2e-6 * bound + 1e-30
    assert (err <= lim).all(), f"{what}: worst ratio {(err / lim).max().item():.3f}"


SHAPES = [
    # M,   N,    K
    (130, 182, 190),       # coco class head on a 10 x 13 grid: ragged everywhere
    (845, 450, 1800),      # ADE group units (a tenth of the 2 x 65 x 65 crop)
    (1, 1, 1),
    (127, 129, 17),
    (256, 128, 16),
    (300, 33, 1030),
    # shapes the bf16x3 kernel takes (>= 8 tiles of 128 x 128, little padding):
    (1000, 300, 203),      # ragged k (203 = 8 * 25 + 3), ragged rows and columns
    (640, 384, 1030),      # whole tiles, k % 8 = 6, split into 8 slabs
    (1290, 520, 64),       # one chunk pair only
]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_rows_gemm_three_layouts(M, N, K):
    from scaleprotoseg_amd import functional as F_

    dev = _dev()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g)
    # y = a . w^T (both operands k-contiguous)
    y = F_._rows_gemm(a.to(dev), (K, 1), w.to(dev), (K, 1), M, N, K)
    _check(y, *_ref(a, w), "a . w^T")
    # d_a = g . w (B with the k index on rows)
    go = torch.randn(M, N, generator=g)
    da = F_._rows_gemm(go.to(dev), (N, 1), w.to(dev), (1, K), M, K, N)
    _check(da, *_ref(go, w.t().contiguous()), "g . w")
    # d_w = g^T . a (both operands with the pixel index on rows: a long contraction, split over workgroups)
    dw = F_._rows_gemm(go.to(dev), (1, N), a.to(dev), (1, K), N, K, M)
    _check(dw, *_ref(go.t().contiguous(), a.t().contiguous()), "g^T . a")


def test_rows_gemm_long_contraction_is_split_and_deterministic():
    from scaleprotoseg_amd import _lib
    from scaleprotoseg_amd import functional as F_

    dev = _dev()
    M, n1, n2 = 40000, 182, 190
    assert _lib.load().spx_rows_gemm_workspace_bytes(n1, n2, M, 0) > 0           # few tiles, long k: slabs
    g = torch.Generator().manual_seed(3)
    a = torch.randn(M, n1, generator=g)
    b = torch.randn(M, n2, generator=g)
    o1 = F_._pixel_outer(a.to(dev), b.to(dev))
    o2 = F_._pixel_outer(a.to(dev), b.to(dev))
    assert torch.equal(o1, o2)
    _check(o1, *_ref(a.t().contiguous(), b.t().contiguous()), "a^T . b")


def test_wide_group_tail_forward_backward():
    """logits = exp(units) . W_g^T and its gradients (model_multiscale_group.py:303-308 through autograd)."""
    from scaleprotoseg_amd import functional as F_

    dev = _dev()
    M, U, K2 = 333, 150, 50
    g = torch.Generator().manual_seed(4)
    units = torch.randn(M, U, generator=g)
    wg = torch.randn(K2, U, generator=g) * 0.1
    go = torch.randn(M, K2, generator=g)
    u0 = units.double().requires_grad_(True)
    w0 = wg.double().requires_grad_(True)
    ref = torch.exp(u0) @ w0.t()
    ref.backward(go.double())
    u1 = units.to(dev).requires_grad_(True)
    w1 = wg.to(dev).requires_grad_(True)
    out = F_.wide_group_tail(u1, w1)
    out.backward(go.to(dev))
    for got, want, what in ((out, ref.detach(), "logits"), (u1.grad, u0.grad, "d_units"), (w1.grad, w0.grad, "d_W_g")):
        err = (got.double().cpu() - want).abs().max().item()
        assert err <= 1e-5 * want.abs().max().item(), f"{what}: {err:.3e} of {want.abs().max().item():.3e}"


def test_wide_group_tail_on_the_bf16x3_kernel():
    """The same three products at a size the bf16x3 kernel takes, with the exponential on either operand and in the epilogue
    (flags 1 / 2 / 4 of spx_rows_gemm)."""
    from scaleprotoseg_amd import functional as F_

    dev = _dev()
    M, U, K2 = 1024, 450, 256
    g = torch.Generator().manual_seed(5)
    units = torch.randn(M, U, generator=g)
    wg = torch.randn(K2, U, generator=g) * 0.1
    go = torch.randn(M, K2, generator=g)
    u0 = units.double().requires_grad_(True)
    w0 = wg.double().requires_grad_(True)
    ref = torch.exp(u0) @ w0.t()
    ref.backward(go.double())
    u1 = units.to(dev).requires_grad_(True)
    w1 = wg.to(dev).requires_grad_(True)
    out = F_.wide_group_tail(u1, w1)
    out.backward(go.to(dev))
    for got, want, what in ((out, ref.detach(), "logits"), (u1.grad, u0.grad, "d_units"), (w1.grad, w0.grad, "d_W_g")):
        err = (got.double().cpu() - want).abs().max().item()
        assert err <= 1e-5 * want.abs().max().item(), f"{what}: {err:.3e} of {want.abs().max().item():.3e}"


def test_rows_gemm_rejects_bad_arguments():
    from scaleprotoseg_amd import SpxError
    from scaleprotoseg_amd import functional as F_

    dev = _dev()
    a = torch.zeros(8, 8, device=dev)
    with pytest.raises(SpxError):
        F_._rows_gemm(a, (2, 2), a, (8, 1), 4, 4, 4)                # no unit stride
    with pytest.raises(SpxError):
        F_._rows_gemm(a, (8, 1), a, (8, 1), 4, 4, 4, flags=4)       # flag 4 without E
    with pytest.raises(SpxError):
        F_._rows_gemm(a.cpu(), (8, 1), a.cpu(), (8, 1), 4, 4, 4)    # no CPU path
