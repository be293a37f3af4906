"""GPU parity of the fused persistent backward (spx_dist_bwd_fused: dX and dPrototypes from one kernel, csrc/spx_bwdf_impl.h)
against the CPU oracle (autograd through the reference's op sequence, segmentation/model/model_multiscale.py:255-281,
:324-330, :243-244) and against the two-kernel backward on the same inputs.

Tolerances as everywhere (tests/test_gpu_parity.py): every gradient max|g - g_ref| <= 1e-3 max|g_ref|; dX returned in bf16
(bf16 features) 4e-3 (one output rounding).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ppnet_oracle as O

from test_gpu_parity import BF16_DX_TOL, GRAD_TOL, _dev, _grad_close, _layout, _problem


def _run(conv, bank, Wl, layout, g_logits, g_dist, x_dtype, dev, fused, freeze=()):
    from scaleprotoseg_amd import functional as F_

    old = F_.FUSED_BACKWARD
    F_.FUSED_BACKWARD = fused
    try:
        x = conv.to(dev, x_dtype).requires_grad_("x" not in freeze)
        pv = bank.to(dev).requires_grad_("bank" not in freeze)
        w = Wl.to(dev).requires_grad_("head" not in freeze) if Wl is not None else None
        logits, dist, _ = F_.proto_head_forward(x, pv, w, layout, want_distances=True)
        outs, gouts = [], []
        if g_logits is not None:
            outs.append(logits); gouts.append(g_logits.reshape(logits.shape).to(dev))
        if g_dist is not None:
            outs.append(dist); gouts.append(g_dist.to(dev))
        torch.autograd.backward(outs, gouts)
        torch.cuda.synchronize()
        return x.grad, pv.grad, (w.grad if w is not None else None)
    finally:
        F_.FUSED_BACKWARD = old


FUSED_SHAPES = [
    # B, S, Cs,  P,   K,  H,  W
    (1, 1, 256, 190, 19, 16, 64),     # north-star bank, 8 full tiles
    (2, 1, 256, 190, 19, 17, 19),     # odd grid: ragged image ends, unaligned rows
    (1, 1, 64, 190, 19, 9, 16),       # 64 channels: two channel blocks only
    (1, 1, 128, 170, 7, 24, 40),      # 170 prototypes: a partial last block
    (3, 1, 256, 192, 32, 10, 50),     # exactly 192 prototypes, 32 head rows
    (1, 1, 48, 161, 3, 5, 7),         # Cs = 48 (a 16-channel tail), one partial tile
    (1, 1, 256, 190, 19, 96, 512),    # 384 tiles: more tiles than workgroups (several tiles per workgroup)
]


@pytest.mark.parametrize("shape", FUSED_SHAPES)
@pytest.mark.parametrize("x_dtype", [torch.float32, torch.bfloat16])
def test_fused_backward_matches_oracle(shape, x_dtype):
    from scaleprotoseg_amd import _lib

    dev = _dev()
    B, S, Cs, P, K, H, W = shape
    conv, bank, Wl, ident, ranges = _problem(*shape, seed=11)
    layout = _layout(P, K, S, Cs, ranges)
    import ctypes as C
    assert _lib.load().spx_bwd_fused_supported(C.byref(layout.plan())) == 1
    g = torch.Generator().manual_seed(5)
    g_logits = torch.randn(B, H, W, K, generator=g) * 1e-3
    g_dist = torch.randn(B, P, H, W, generator=g) * 1e-3
    _, _, _, dx_ref, dp_ref, dw_ref = O.fwd_bwd_reference(conv, bank, ranges, S, Wl, g_logits, g_dist)
    dx, dp, dw = _run(conv, bank, Wl, layout, g_logits, g_dist, x_dtype, dev, fused=True)
    assert dx.dtype == x_dtype and dx.shape == conv.shape
    _grad_close(dx, dx_ref, "dX", tol=GRAD_TOL if x_dtype == torch.float32 else BF16_DX_TOL)
    _grad_close(dp, dp_ref, "dPrototypes")
    _grad_close(dw, dw_ref, "dLastLayer")
    # ... and the two-kernel backward on the same inputs stays within the same bounds of it
    dx2, dp2, dw2 = _run(conv, bank, Wl, layout, g_logits, g_dist, x_dtype, dev, fused=False)
    _grad_close(dx, dx2, "dX vs two-kernel", tol=2 * (GRAD_TOL if x_dtype == torch.float32 else BF16_DX_TOL))
    _grad_close(dp, dp2, "dPrototypes vs two-kernel", tol=2 * GRAD_TOL)
    _grad_close(dw, dw2, "dLastLayer vs two-kernel", tol=1e-5)     # the same blob; only the chunk walk of the d_W instance differs


def test_fused_backward_gradient_magnitudes_vary_across_tiles():
    """The G16 scale is per tile and the persistent accumulators follow it: gradients whose magnitude differs by many
    orders between image regions (and tiles that carry no gradient at all) must still sum to the oracle's d_bank."""
    dev = _dev()
    shape = (1, 1, 256, 190, 19, 80, 512)            # 320 tiles on <= 256 workgroups
    B, S, Cs, P, K, H, W = shape
    conv, bank, Wl, ident, ranges = _problem(*shape, seed=12)
    layout = _layout(P, K, S, Cs, ranges)
    g = torch.Generator().manual_seed(6)
    ramp = torch.logspace(-9, 0, H).reshape(1, 1, H, 1)               # row-wise: 1e-9 ... 1
    ramp[:, :, 10:14] = 0.0                                            # rows (whole tiles) without any gradient
    g_dist = torch.randn(B, P, H, W, generator=g) * ramp
    g_logits = torch.randn(B, H, W, K, generator=g) * ramp.reshape(1, H, 1, 1)
    _, _, _, dx_ref, dp_ref, dw_ref = O.fwd_bwd_reference(conv, bank, ranges, S, Wl, g_logits, g_dist)
    dx, dp, dw = _run(conv, bank, Wl, layout, g_logits, g_dist, torch.float32, dev, fused=True)
    _grad_close(dp, dp_ref, "dPrototypes")
    _grad_close(dw, dw_ref, "dLastLayer")
    # dX: per row band against the band's own maximum (a max-normalised bound over the whole image would only see the last rows)
    for lo in range(0, H, 8):
        ref = dx_ref[:, :, lo:lo + 8]
        if ref.abs().max() > 0:
            _grad_close(dx[:, :, lo:lo + 8], ref, f"dX rows {lo}..", tol=GRAD_TOL)
        else:
            assert dx[:, :, lo:lo + 8].abs().max().item() == 0.0


def test_fused_backward_partial_inputs_and_frozen_parameters():
    dev = _dev()
    shape = (2, 1, 256, 190, 19, 12, 40)
    B, S, Cs, P, K, H, W = shape
    conv, bank, Wl, ident, ranges = _problem(*shape, seed=13)
    layout = _layout(P, K, S, Cs, ranges)
    g = torch.Generator().manual_seed(8)
    g_logits = torch.randn(B, H, W, K, generator=g) * 1e-3
    g_dist = torch.randn(B, P, H, W, generator=g) * 1e-3
    # logits only
    _, _, _, dx_ref, dp_ref, dw_ref = O.fwd_bwd_reference(conv, bank, ranges, S, Wl, g_logits, torch.zeros(B, P, H, W))
    dx, dp, dw = _run(conv, bank, Wl, layout, g_logits, None, torch.float32, dev, fused=True)
    _grad_close(dx, dx_ref, "dX (logits only)")
    _grad_close(dp, dp_ref, "dP (logits only)")
    _grad_close(dw, dw_ref, "dW (logits only)")
    # distances only, head frozen; bank frozen; x frozen
    _, _, _, dx_ref, dp_ref, _ = O.fwd_bwd_reference(conv, bank, ranges, S, Wl, torch.zeros(B, H, W, K), g_dist)
    dx, dp, dw = _run(conv, bank, Wl, layout, None, g_dist, torch.float32, dev, fused=True, freeze=("head",))
    assert dw is None
    _grad_close(dx, dx_ref, "dX (distances only)")
    _grad_close(dp, dp_ref, "dP (distances only)")
    dx, dp, _ = _run(conv, bank, Wl, layout, None, g_dist, torch.float32, dev, fused=True, freeze=("bank", "head"))
    assert dp is None
    _grad_close(dx, dx_ref, "dX (bank frozen)")
    dx, dp, _ = _run(conv, bank, Wl, layout, None, g_dist, torch.float32, dev, fused=True, freeze=("x", "head"))
    assert dx is None
    _grad_close(dp, dp_ref, "dP (x frozen)")


def test_fused_backward_is_run_to_run_identical():
    dev = _dev()
    shape = (1, 1, 256, 190, 19, 64, 256)
    B, S, Cs, P, K, H, W = shape
    conv, bank, Wl, ident, ranges = _problem(*shape, seed=14)
    layout = _layout(P, K, S, Cs, ranges)
    g = torch.Generator().manual_seed(9)
    g_logits = torch.randn(B, H, W, K, generator=g) * 1e-3
    g_dist = torch.randn(B, P, H, W, generator=g) * 1e-3
    a = _run(conv, bank, Wl, layout, g_logits, g_dist, torch.bfloat16, dev, fused=True)
    b = _run(conv, bank, Wl, layout, g_logits, g_dist, torch.bfloat16, dev, fused=True)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_pack_cache_follows_every_parameter_edit():
    """The packed operands are reused while the parameters are unchanged and rebuilt after each kind of edit the reference makes
    (optimizer step, push commit, prune / re-assignment, simplex projection's ``weight.data = ...``)."""
    from scaleprotoseg_amd import functional as F_

    dev = _dev()
    shape = (1, 4, 64, 228, 19, 9, 13)
    B, S, Cs, P, K, H, W = shape
    conv, bank, Wl, ident, ranges = _problem(*shape, seed=21)
    layout = _layout(P, K, S, Cs, ranges)
    x = conv.to(dev)
    pv = torch.nn.Parameter(bank.to(dev))
    w = torch.nn.Parameter(Wl.to(dev))
    F_.invalidate_pack_cache()
    st = F_.PACK_CACHE_STATS

    def fwd():
        with torch.no_grad():
            return F_.proto_head_forward(x, pv, w, layout, want_distances=True)[1].clone()

    h0, m0 = st["hits"], st["misses"]
    d0 = fwd()
    d1 = fwd()
    assert (st["hits"], st["misses"]) == (h0 + 1, m0 + 1) and torch.equal(d0, d1)
    # an optimizer step (in place through the parameter: its version counter moves)
    opt = torch.optim.SGD([pv, w], lr=0.1)
    pv.grad = torch.ones_like(pv)
    w.grad = torch.zeros_like(w)
    opt.step()
    d2 = fwd()
    assert st["misses"] == m0 + 2 and not torch.equal(d2, d0)
    ref = O.forward_from_conv_features(conv, pv.detach().cpu(), ranges, S, w.detach().cpu())[1]
    assert ((d2.cpu() - ref).abs() <= 2e-2 * (1 + ref)).all()          # (the stepped bank is no longer bf16-representable)
    # new storage (the simplex projection's ``weight.data = ...``)
    w.data = w.data.clone() * 0.5
    fwd()
    assert st["misses"] == m0 + 3
    # an in-place write through .data is invisible to the version counter: the package's own sites invalidate explicitly
    pv.data.copy_(bank.to(dev))
    F_.invalidate_pack_cache()
    d3 = fwd()
    assert st["misses"] == m0 + 4 and torch.equal(d3, d0)
