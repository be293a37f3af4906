"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Stated tolerances (inputs are bf16-representable fp32, SURVEY.md 8d "identical inputs"):
  distances   |d - d_ref| <= 1e-4 * (1 + d_ref)
  activations |a - a_ref| <= 2e-4 * (1 + |a_ref|)
  logits      |l - l_ref| <= 1e-4 * max(1, max|l_ref|)      (1e-4 relative, north star)
  gradients   max|g - g_ref| <= 1e-3 * max|g_ref| for dX (fp32 features), dPrototypes, dLastLayer, dGroupProjection and
              dLastLayerGroup everywhere (G enters dX = 2(rs x - P^T G) and crosses to the parameter kernel as ONE fp16
              plane with a power-of-two scale per tile; the activations cross as block-scaled int16 rebuilt as an exact bf16
              hi + lo pair; dLogits as split bf16); 4e-3 for dX RETURNED in bf16 (bf16 features): the output rounding alone
              is 2^-9
  push        indices bit-exact, values bit-exact given the same distance map
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import loss_oracle as LO
from oracle import ppnet_oracle as O


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _problem(B, S, Cs, P, K, H, W, seed=0, x_dtype=torch.float32):
    g = torch.Generator().manual_seed(20220227 + seed)
    conv = O.bf16_representable(torch.sigmoid(torch.randn(B, S * Cs, H, W, generator=g)))
    bank = O.bf16_representable(torch.rand(P, Cs, 1, 1, generator=g))
    ident = O.default_class_identity(P, K, S)
    Wl = O.last_layer_init(ident) + 0.05 * torch.randn(K, P, generator=g)
    return conv, bank, Wl, ident, O.default_scale_ranges(P, S)


def _layout(P, K, S, Cs, ranges):
    from scaleprotoseg_amd.functional import BankLayout

    return BankLayout(P, K, S, Cs, tuple(ranges[s] for s in range(S)))


def _assert_fwd(logits, dist, act, ref_logits, ref_dist, ref_act):
    if dist is not None:
        err = (dist.cpu() - ref_dist).abs()
        assert (err <= 1e-4 * (1 + ref_dist)).all(), f"distance err {err.max().item()}"
    if act is not None:
        err = (act.cpu() - ref_act).abs()
        assert (err <= 2e-4 * (1 + ref_act.abs())).all(), f"activation err {err.max().item()}"
    if logits is not None:
        rl = ref_logits.reshape(-1, ref_logits.shape[-1])
        d = (logits.cpu().reshape(rl.shape) - rl).abs()
        err = d.max().item()
        assert err <= 1e-4 * max(1.0, rl.abs().max().item()), f"logit err {err}"
        # ... and per element, relative with an absolute floor (a logit is a signed sum of P terms that passes through zero):
        # |err| <= 1.1e-4 (|l| + 0.2 max|l|).  What bounds it is the operand precision of the head product - activations and head
        # weights enter the MFMA as bf16 hi + lo pairs, 2^-17 of each TERM |a_p w_kp| (a <= 9.2) - so an element's error scales
        # with the sum of its terms' magnitudes, not with the element.  700 fuzzed configurations peak at 1.002e-4 of this
        # measure (seed 378 of test_random_gather_and_tail, in the suite below); 1e-4 with a 0.1 floor would take 24-bit
        # operands (a third bf16 plane of W and a: twice the head MFMAs) - DESIGN.md 4
        floor = 0.2 * max(1.0, rl.abs().max().item())
        assert (d <= 1.1e-4 * (rl.abs() + floor)).all(), f"per-element logit err {(d / (rl.abs() + floor)).max().item()}"


SHAPES = [
    # B, S, Cs,  P,   K,  H,  W
    (2, 4, 64, 228, 19, 17, 19),     # Cityscapes ScaleProtoSeg bank, odd grid (scalar-load path)
    (1, 1, 256, 190, 19, 16, 64),    # north-star bank, 16-B aligned rows (vector-load path)
    (1, 1, 64, 210, 21, 13, 11),     # Pascal baseline bank (2 panels of 105)
    (2, 4, 64, 252, 21, 8, 16),      # Pascal ScaleProtoSeg
    (1, 3, 64, 171, 19, 9, 16),      # 3-scale variant
    (1, 2, 16, 16, 3, 5, 7),         # Cs = 16 (kc = 16), floor semantics (unassigned prototypes)
    (1, 4, 64, 1800, 150, 9, 8),     # ADE bank: 3 panels per scale, 5 class blocks
    (1, 1, 64, 1500, 150, 6, 8),     # ADE literal 150 x 10
    (1, 4, 64, 228, 57, 9, 13),      # 57 head rows (grouping head): 2 class blocks
    (1, 1, 256, 190, 64, 8, 16),     # exactly 64 head rows, 6-block panel
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("x_dtype", [torch.float32, torch.bfloat16])
def test_forward_matches_oracle(shape, x_dtype):
    from scaleprotoseg_amd.functional import proto_head_forward

    dev = _dev()
    B, S, Cs, P, K, H, W = shape
    conv, bank, Wl, ident, ranges = _problem(*shape)
    ref_logits, ref_dist, ref_act = O.forward_from_conv_features(conv, bank, ranges, S, Wl)
    logits, dist, act = proto_head_forward(
        conv.to(dev, x_dtype), bank.to(dev), Wl.to(dev), _layout(P, K, S, Cs, ranges),
        want_distances=True, want_activations=True,
    )
    torch.cuda.synchronize()
    assert dist.shape == (B, P, H, W) and act.shape == (B * H * W, P) and logits.shape == (B * H * W, K)
    _assert_fwd(logits, dist, act, ref_logits, ref_dist, ref_act)


@pytest.mark.parametrize("name", ["proto_ms_city", "proto_s1_wide", "proto_s3", "proto_floor", "proto_ms_small"])
def test_forward_matches_golden(golden, name):
    """Fixtures generated from the reference's own classes (oracle/gen_golden.py)."""
    from scaleprotoseg_amd.functional import proto_head_forward

    dev = _dev()
    g = golden(name)
    S = int(g["num_scales"])
    conv = torch.from_numpy(g["conv"])
    bank = torch.from_numpy(g["prototype_vectors"])
    Wl = torch.from_numpy(g["last_layer_weight"])
    P, Cs = bank.shape[0], bank.shape[1]
    ranges = {s: (int(lo), int(hi)) for s, (lo, hi) in enumerate(g["scale_ranges"])}
    logits, dist, act = proto_head_forward(
        conv.to(dev), bank.to(dev), Wl.to(dev), _layout(P, Wl.shape[0], S, Cs, ranges),
        want_distances=True, want_activations=True,
    )
    _assert_fwd(logits, dist, act, torch.from_numpy(g["logits"]), torch.from_numpy(g["distances"]),
                torch.from_numpy(g["activations"]))


GRAD_TOL = 1e-3        # gradients rel 1e-3 (max-normalised)
BF16_DX_TOL = 4e-3     # dX returned in bf16 (bf16 features): ONE output rounding is half a bf16 ulp = 2^-8 of the element


def _dx_tol(x_dtype, ranges=None):
    """fp32 features: 1e-3.  bf16 features: dX is returned in bf16 - ONE output rounding, also for scales of more than 192
    prototypes (several panels: their shares are summed in an fp32 scratch, round 4; through the bf16 buffer it was one
    rounding per panel)."""
    return GRAD_TOL if x_dtype == torch.float32 else BF16_DX_TOL


def _grad_close(got, ref, what, tol=GRAD_TOL):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"{what}: err {err:.3e} vs scale {scale:.3e}"


BWD_SHAPES = [
    (1, 4, 64, 228, 57, 9, 13),      # dense grouping head of the Cityscapes group phase (3 x 19 rows: 2 class blocks)
    (1, 1, 64, 210, 63, 8, 16),      # Pascal group phase (3 x 21 rows), 4-block panels
    (2, 4, 64, 228, 19, 17, 19),
    (1, 1, 256, 190, 19, 16, 64),
    (1, 1, 64, 210, 21, 13, 11),
    (1, 2, 16, 16, 3, 5, 7),
    (1, 4, 64, 1800, 150, 5, 8),
    # bf16 features, H*W a multiple of 64, scales wider than 64 channels, one class block: the LDS-DMA parameter kernel
    (2, 1, 96, 100, 7, 8, 24),       # 3 units per image (the last tile half empty), 96 channels (zero rows), 4-block panel
    (1, 2, 128, 600, 19, 16, 40),    # two scales x two panels of 150
    (3, 1, 80, 40, 5, 8, 8),         # one unit per image, 2-block panel
]


@pytest.mark.parametrize("shape", BWD_SHAPES)
@pytest.mark.parametrize("x_dtype", [torch.float32, torch.bfloat16])
def test_backward_matches_oracle(shape, x_dtype):
    from scaleprotoseg_amd.functional import proto_head_forward

    dev = _dev()
    B, S, Cs, P, K, H, W = shape
    conv, bank, Wl, ident, ranges = _problem(*shape, seed=1)
    g = torch.Generator().manual_seed(7)
    g_logits = torch.randn(B, H, W, K, generator=g) * 1e-3
    g_dist = torch.randn(B, P, H, W, generator=g) * 1e-3
    g_act = torch.randn(B * H * W, P, generator=g) * 1e-3
    _, _, _, dx_ref, dp_ref, dw_ref = O.fwd_bwd_reference(conv, bank, ranges, S, Wl, g_logits, g_dist, g_act)

    x = conv.to(dev, x_dtype).requires_grad_(True)
    pv = bank.to(dev).requires_grad_(True)
    w = Wl.to(dev).requires_grad_(True)
    logits, dist, act = proto_head_forward(x, pv, w, _layout(P, K, S, Cs, ranges), want_distances=True, want_activations=True)
    loss = (logits * g_logits.reshape(-1, K).to(dev)).sum() + (dist * g_dist.to(dev)).sum() + (act * g_act.to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()
    assert x.grad.dtype == x_dtype and x.grad.shape == x.shape
    _grad_close(x.grad, dx_ref, "dX", tol=_dx_tol(x_dtype, ranges))
    _grad_close(pv.grad, dp_ref, "dPrototypes")
    _grad_close(w.grad, dw_ref, "dLastLayer")


@pytest.mark.parametrize("name", ["proto_ms_city", "proto_s1_wide"])
def test_backward_matches_golden(golden, name):
    from scaleprotoseg_amd.functional import proto_head_forward

    dev = _dev()
    g = golden(name)
    S = int(g["num_scales"])
    bank = torch.from_numpy(g["prototype_vectors"])
    Wl = torch.from_numpy(g["last_layer_weight"])
    P, Cs, K = bank.shape[0], bank.shape[1], Wl.shape[0]
    ranges = {s: (int(lo), int(hi)) for s, (lo, hi) in enumerate(g["scale_ranges"])}
    x = torch.from_numpy(g["conv"]).to(dev).requires_grad_(True)
    pv = bank.to(dev).requires_grad_(True)
    w = Wl.to(dev).requires_grad_(True)
    logits, dist, act = proto_head_forward(x, pv, w, _layout(P, K, S, Cs, ranges), want_distances=True, want_activations=True)
    loss = (
        (logits * torch.from_numpy(g["g_logits"]).reshape(-1, K).to(dev)).sum()
        + (dist * torch.from_numpy(g["g_dist"]).to(dev)).sum()
        + (act * torch.from_numpy(g["g_act"]).to(dev)).sum()
    )
    loss.backward()
    _grad_close(x.grad, torch.from_numpy(g["d_conv"]), "dX")
    _grad_close(pv.grad, torch.from_numpy(g["d_prototypes"]), "dPrototypes")
    _grad_close(w.grad, torch.from_numpy(g["d_last_layer"]), "dLastLayer")


def test_backward_partial_inputs():
    """Only logits consumed, bank frozen: dX and dW still flow; distances-only: dX and dP flow."""
    from scaleprotoseg_amd.functional import proto_head_forward

    dev = _dev()
    shape = (1, 4, 64, 228, 19, 9, 16)
    B, S, Cs, P, K, H, W = shape
    conv, bank, Wl, ident, ranges = _problem(*shape, seed=2)
    g = torch.Generator().manual_seed(3)
    g_logits = torch.randn(B, H, W, K, generator=g) * 1e-3
    zeros_d = torch.zeros(B, P, H, W)
    _, _, _, dx_ref, dp_ref, dw_ref = O.fwd_bwd_reference(conv, bank, ranges, S, Wl, g_logits, zeros_d)
    x = conv.to(dev).requires_grad_(True)
    w = Wl.to(dev).requires_grad_(True)
    logits, dist, _ = proto_head_forward(x, bank.to(dev), w, _layout(P, K, S, Cs, ranges))
    (logits * g_logits.reshape(-1, K).to(dev)).sum().backward()
    _grad_close(x.grad, dx_ref, "dX (logits only)")
    _grad_close(w.grad, dw_ref, "dW (logits only)")

    g_dist = torch.randn(B, P, H, W, generator=g) * 1e-3
    _, _, _, dx_ref, dp_ref, _ = O.fwd_bwd_reference(conv, bank, ranges, S, Wl, torch.zeros(B, H, W, K), g_dist)
    x = conv.to(dev).requires_grad_(True)
    pv = bank.to(dev).requires_grad_(True)
    _, dist, _ = proto_head_forward(x, pv, None, _layout(P, 1, S, Cs, ranges))
    (dist * g_dist.to(dev)).sum().backward()
    _grad_close(x.grad, dx_ref, "dX (distances only)")
    _grad_close(pv.grad, dp_ref, "dP (distances only)")


def test_push_argmin_golden(golden):
    from scaleprotoseg_amd.functional import push_masked_argmin

    dev = _dev()
    g = golden("push_argmin")
    idx, val = push_masked_argmin(
        torch.from_numpy(g["distances"]).to(dev), torch.from_numpy(g["target"])[None].to(dev),
        torch.from_numpy(g["class_identity"]), void_class=0,
    )
    np.testing.assert_array_equal(idx.cpu().numpy(), g["indices"])
    np.testing.assert_array_equal(val.cpu().numpy(), g["values"])


@pytest.mark.parametrize("hw", [(129, 257), (65, 65), (96, 512)])
def test_push_argmin_random(hw):
    """Bit-exact indices incl. absent classes, exact ties and multi-chunk rows."""
    from scaleprotoseg_amd.functional import push_masked_argmin

    dev = _dev()
    H, W = hw
    B, K, S, r = 2, 19, 4, 3
    P = K * S * r
    g = torch.Generator().manual_seed(11)
    dist = torch.rand(B, P, H, W, generator=g) * 50
    # coarse quantisation -> many exact ties; lowest flat index must win
    dist[:, ::3] = torch.floor(dist[:, ::3] * 4) / 4
    target = torch.randint(0, K + 1, (B, H, W), generator=g)
    target[target == 5] = 1       # class 4 absent everywhere
    target[1][target[1] == 9] = 0  # class 8 absent in image 1
    ident = O.default_class_identity(P, K, S)
    ref_idx, ref_val = O.push_masked_argmin(dist, target, ident, K, void_class=0)
    idx, val = push_masked_argmin(dist.to(dev), target.to(dev), ident, void_class=0)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref_idx.numpy())
    np.testing.assert_array_equal(val.cpu().numpy(), ref_val.numpy())
    assert (ref_val == 1e10).any()


PUSH_FUSED_SHAPES = [
    # B, S, Cs,  P,   K,  H,   W
    (2, 4, 64, 228, 19, 33, 65),      # the gin's bank, odd grid (ragged last tile, unaligned rows)
    (1, 1, 256, 190, 19, 64, 128),    # north-star bank, 64 full tiles
    (3, 2, 32, 44, 11, 9, 13),        # tiny: one partial tile per image
    (1, 4, 64, 1800, 150, 17, 19),    # ADE bank: several panels per scale
]


@pytest.mark.parametrize("shape", PUSH_FUSED_SHAPES)
@pytest.mark.parametrize("x_dtype", [torch.float32, torch.bfloat16])
def test_fused_push_minimum_equals_the_two_step_push(shape, x_dtype):
    """spx_dist_push_min (minimum taken inside the distance kernel, no map) against (a) the two-step path - the map
    spx_dist_fwd writes reduced by spx_push_argmin - bit for bit, and (b) the oracle's push on that same map
    (push_multiscale_optimization.py:74-91).  Absent classes, void pixels, duplicated prototypes (exact ties between rows)
    and a prototype copied from a pixel (distance 0) are all in."""
    from scaleprotoseg_amd.functional import proto_head_forward, push_masked_argmin, push_min_from_features

    dev = _dev()
    B, S, Cs, P, K, H, W = shape
    conv, bank, Wl, ident, ranges = _problem(*shape, seed=31)
    conv[:, :, 2:4, :] = conv[:, :, 0:2, :]            # repeated rows of pixels: exact ties along a prototype row
    cv = conv[0].view(S, Cs, H, W)
    p_sel = next(p for p in range(P // S) if int(ident[p].argmax()) != 2)
    bank[p_sel, :, 0, 0] = cv[0, :, 1, 2]                  # a pushed prototype: distance 0 at (1, 2) - and at its copy (3, 2)
    bank[p_sel + 2] = bank[p_sel + 1]
    g = torch.Generator().manual_seed(32)
    target = torch.randint(0, K + 1, (B, H, W), generator=g)
    target[target == 3] = 1                                # class 2 absent everywhere
    target[0, 1, 2] = target[0, 3, 2] = int(ident[p_sel].argmax()) + 1
    layout = _layout(P, 1, S, Cs, ranges)
    x = conv.to(dev, x_dtype)
    _, dist, _ = proto_head_forward(x, bank.to(dev), None, layout)
    idx2, val2 = push_masked_argmin(dist, target.to(dev), ident, void_class=0)
    idx, val = push_min_from_features(x, bank.to(dev), layout, target.to(dev), ident, void_class=0)
    np.testing.assert_array_equal(idx.cpu().numpy(), idx2.cpu().numpy())
    np.testing.assert_array_equal(val.cpu().numpy(), val2.cpu().numpy())
    ref_idx, ref_val = O.push_masked_argmin(dist.cpu(), target, ident, K, void_class=0)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref_idx.numpy())
    np.testing.assert_array_equal(val.cpu().numpy(), ref_val.numpy())
    assert (ref_val == 1e10).any() and (ref_val < 1e-3).any()
    # labels without a void value (void_class=None: 0..K-1)
    idx3, val3 = push_min_from_features(x, bank.to(dev), layout, (target - 1).clamp(min=0).to(dev), ident, void_class=None)
    r3 = O.push_masked_argmin(dist.cpu(), (target - 1).clamp(min=0), ident, K, void_class=None)
    np.testing.assert_array_equal(idx3.cpu().numpy(), r3[0].numpy())
    np.testing.assert_array_equal(val3.cpu().numpy(), r3[1].numpy())


def test_fused_push_rejects_a_fractional_identity():
    from scaleprotoseg_amd import SpxError
    from scaleprotoseg_amd.functional import push_min_from_features

    dev = _dev()
    shape = (1, 1, 32, 20, 5, 4, 8)
    B, S, Cs, P, K, H, W = shape
    conv, bank, Wl, ident, ranges = _problem(*shape, seed=33)
    ident = ident.clone()
    ident[0, 0] = 0.5
    with pytest.raises(SpxError):
        push_min_from_features(conv.to(dev), bank.to(dev), _layout(P, 1, S, Cs, ranges), torch.zeros(B, H, W, dtype=torch.long), ident)


def test_argmin_over_images():
    from scaleprotoseg_amd.functional import argmin_over_images

    dev = _dev()
    g = torch.Generator().manual_seed(5)
    vals = torch.floor(torch.rand(37, 228, generator=g) * 8)  # ties -> lowest image index
    best = argmin_over_images(vals.to(dev))
    np.testing.assert_array_equal(best.cpu().numpy(), vals.argmin(dim=0).numpy())


def test_prototype_equal_to_pixel():
    """After a push a prototype IS a latent pixel: its distance there must be ~0 and win the argmin."""
    from scaleprotoseg_amd.functional import proto_head_forward, push_masked_argmin

    dev = _dev()
    shape = (1, 4, 64, 228, 19, 12, 16)
    B, S, Cs, P, K, H, W = shape
    conv, bank, Wl, ident, ranges = _problem(*shape, seed=4)
    cv = conv.view(S, Cs, H, W)
    for p in range(0, P, 7):
        s = p // (P // S)
        bank[p, :, 0, 0] = cv[s, :, (p * 5) % H, (p * 3) % W]
    _, dist, _ = proto_head_forward(conv.to(dev), bank.to(dev), None, _layout(P, 1, S, Cs, ranges))
    for p in range(0, P, 7):
        assert dist[0, p, (p * 5) % H, (p * 3) % W].item() <= 1e-4
    ref = O.scale_l2_convolution(conv, bank, ranges, S)
    assert ((dist.cpu() - ref).abs() <= 1e-4 * (1 + ref)).all()


def test_large_shape_properties():
    """North-star bank at a multi-megapixel grid: tile-independent properties instead of a CPU reference.
    (a) a pixel's outputs do not depend on where it sits in the grid, (b) logits are linear in the head."""
    from scaleprotoseg_amd.functional import proto_head_forward

    dev = _dev()
    S, Cs, P, K = 1, 256, 190, 19
    H, W = 256, 512
    conv, bank, Wl, ident, ranges = _problem(1, S, Cs, P, K, 8, 16, seed=6)
    small = conv.to(dev, torch.bfloat16)
    big = small.repeat(1, 1, H // 8, W // 16).contiguous()
    lay = _layout(P, K, S, Cs, ranges)
    l_s, d_s, _ = proto_head_forward(small, bank.to(dev), Wl.to(dev), lay)
    l_b, d_b, _ = proto_head_forward(big, bank.to(dev), Wl.to(dev), lay)
    torch.cuda.synchronize()
    assert torch.equal(d_b, d_s.repeat(1, 1, H // 8, W // 16))
    assert torch.equal(l_b.view(H, W, K), l_s.view(8, 16, K).repeat(H // 8, W // 16, 1))
    l2, _, _ = proto_head_forward(big, bank.to(dev), (2 * Wl).to(dev), lay)
    assert torch.allclose(l2, 2 * l_b, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------------
# class-gathered distances: spx_dist_fwd_cls / spx_dist_bwd_cls
# ------------------------------------------------------------------------------------------------
GATHER_SHAPES = [
    (2, 4, 64, 228, 19, 17, 19),     # 3 prototypes per (class, scale), odd grid
    (1, 1, 256, 190, 19, 16, 64),    # north-star bank: 10 prototypes per class
    (1, 2, 16, 16, 3, 5, 7),         # floor semantics: prototypes without a class
    (2, 4, 64, 252, 21, 8, 16),
]


def _labels(B, H, W, K, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.randint(0, K + 1, (B, H, W), generator=g)      # reference convention: 0 = void, 1..K
    t[t == K] = 0                                            # last class absent ...
    t[0, 0, 0] = K                                           # ... but for one pixel
    return t


@pytest.mark.parametrize("shape", GATHER_SHAPES)
@pytest.mark.parametrize("x_dtype", [torch.float32, torch.bfloat16])
def test_class_gathered_forward_backward(shape, x_dtype):
    """The gathered output equals the oracle's gather of the reference distance map (same tolerance as the map),
    and gradients through it equal autograd through the full map with the scattered gradient."""
    from scaleprotoseg_amd.functional import ClassGather, class_gather_table, proto_head_forward

    dev = _dev()
    B, S, Cs, P, K, H, W = shape
    conv, bank, Wl, ident, ranges = _problem(*shape, seed=11)
    target = _labels(B, H, W, K, seed=5)
    lab0 = (target.reshape(B, -1) - 1)
    lay = _layout(P, K, S, Cs, ranges)
    keys, J, table = class_gather_table(lay, ident, dev)
    assert torch.equal(table.cpu(), O.class_slot_table(ident))
    gather = ClassGather(labels=lab0.to(dev, torch.int32).contiguous(), keys=keys, width=J, table=table)

    g = torch.Generator().manual_seed(9)
    g_logits = torch.randn(B, H, W, K, generator=g) * 1e-3
    g_cls = torch.randn(B, H * W, J, generator=g) * 1e-3        # oracle layout [B, HW, J]; the kernels use [B, J, HW]

    # oracle: full map -> gather; loss on the gathered entries
    c0 = conv.clone().requires_grad_(True)
    p0 = bank.clone().requires_grad_(True)
    w0 = Wl.clone().requires_grad_(True)
    l_ref, d_ref, _ = O.forward_from_conv_features(c0, p0, ranges, S, w0)
    cd_ref = O.gather_class_distances(d_ref, lab0, ident)
    ((l_ref * g_logits).sum() + (cd_ref * g_cls).sum()).backward()

    x = conv.to(dev, x_dtype).requires_grad_(True)
    pv = bank.to(dev).requires_grad_(True)
    w = Wl.to(dev).requires_grad_(True)
    logits, cd, _ = proto_head_forward(x, pv, w, lay, class_gather=gather)
    torch.cuda.synchronize()
    assert tuple(cd.shape) == (B, J, H * W)
    ref = cd_ref.detach()
    got = cd.detach().cpu().permute(0, 2, 1)
    err = (got - ref).abs()
    assert (err <= 1e-4 * (1 + ref)).all(), f"class distance err {err.max().item()}"
    # entries no prototype maps to / void pixels are exactly 0
    assert (got[ref == 0] == 0).all()
    _assert_fwd(logits, None, None, l_ref.detach(), None, None)
    ((logits * g_logits.reshape(-1, K).to(dev)).sum() + (cd * g_cls.permute(0, 2, 1).contiguous().to(dev)).sum()).backward()
    torch.cuda.synchronize()
    _grad_close(x.grad, c0.grad, "dX", tol=_dx_tol(x_dtype, ranges))
    _grad_close(pv.grad, p0.grad, "dPrototypes")
    _grad_close(w.grad, w0.grad, "dLastLayer")


def test_kld_through_the_module(golden):
    """PPNetMultiScale.forward_from_conv_features(target_labels=...) + KLDLoss on the gathered output
    == the oracle's KLD (pinned to the reference's KLDLoss) on the full distance map, value and gradients."""
    import scaleprotoseg_amd as spx

    dev = _dev()
    B, S, Cs, P, K, H, W = 2, 4, 16, 40, 5, 9, 11
    conv, bank, Wl, ident, ranges = _problem(B, S, Cs, P, K, H, W, seed=3)
    target = _labels(B, H, W, K, seed=2)

    class _Feat(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.base = torch.nn.Sequential(torch.nn.Conv2d(3, S * Cs, 1), torch.nn.Conv2d(S * Cs, S * Cs, 1))

        def __repr__(self):
            return "MSC(standin)"

        def forward(self, x):
            return x

    net = spx.PPNetMultiScale(_Feat(), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                              patch_classification=True, num_scales=S).to(dev)
    net.prototype_class_identity = net.prototype_class_identity.to(dev)
    with torch.no_grad():
        net.prototype_vectors.copy_(bank.to(dev))
        net.last_layer.weight.copy_(Wl.to(dev))
    x = conv.to(dev).requires_grad_(True)
    logits, cd = net.forward_from_conv_features(x, target_labels=target.to(dev))
    assert isinstance(cd, spx.ClassDistances)
    kld = spx.KLDLoss(net.prototype_class_identity, S, net.scale_num_prototypes)(cd, target.to(dev))
    kld.backward()
    torch.cuda.synchronize()

    c0 = conv.clone().requires_grad_(True)
    p0 = bank.clone().requires_grad_(True)
    _, d_ref, _ = O.forward_from_conv_features(c0, p0, ranges, S, Wl)
    k_ref = O.kld_loss(d_ref, target, ident, S, ranges)
    k_ref.backward()
    assert abs(kld.item() - k_ref.item()) <= 1e-4 * max(1.0, abs(k_ref.item())), (kld.item(), k_ref.item())
    _grad_close(x.grad, c0.grad, "dX via KLD")
    _grad_close(net.prototype_vectors.grad, p0.grad, "dPrototypes via KLD")
    # same module, same loss class, full-map input: identical value
    logits2, dist2 = net.forward_from_conv_features(conv.to(dev))
    k2 = spx.KLDLoss(net.prototype_class_identity, S, net.scale_num_prototypes)(dist2, target.to(dev))
    assert abs(k2.item() - kld.item()) <= 1e-5 * max(1.0, abs(kld.item()))


def test_north_star_size_properties():
    """BASELINE.json's full size (one 1024x2048 latent grid, C=256, P=190): properties that need no CPU reference.
    (a) forward outputs of a pixel do not depend on its position (a periodic image gives periodic outputs, bit for
    bit); (b) dX of a periodic image under periodic upstream gradients is periodic, bit for bit; (c) dPrototypes and
    dLastLayer equal (number of periods) x the one-period gradients up to fp32 summation order."""
    from scaleprotoseg_amd.functional import proto_head_forward

    dev = _dev()
    S, Cs, P, K = 1, 256, 190, 19
    H, W, h, w = 1024, 2048, 8, 64
    conv, bank, Wl, ident, ranges = _problem(1, S, Cs, P, K, h, w, seed=21)
    lay = _layout(P, K, S, Cs, ranges)
    g = torch.Generator().manual_seed(3)
    gl_s = (torch.randn(h * w, K, generator=g) * 1e-3).to(dev)
    gd_s = (torch.randn(1, P, h, w, generator=g) * 1e-3).to(dev)
    reps = (H // h) * (W // w)

    def run(x, gl, gd):
        x = x.requires_grad_(True)
        pv = bank.to(dev).requires_grad_(True)
        wl = Wl.to(dev).requires_grad_(True)
        logits, dist, _ = proto_head_forward(x, pv, wl, lay)
        torch.autograd.backward([logits, dist], [gl, gd])
        torch.cuda.synchronize()
        return logits.detach(), dist.detach(), x.grad, pv.grad, wl.grad

    small = conv.to(dev, torch.bfloat16)
    l_s, d_s, dx_s, dp_s, dw_s = run(small.clone(), gl_s, gd_s)
    big = small.repeat(1, 1, H // h, W // w).contiguous()
    gl_b = gl_s.view(h, w, K).repeat(H // h, W // w, 1).reshape(H * W, K).contiguous()
    gd_b = gd_s.repeat(1, 1, H // h, W // w).contiguous()
    l_b, d_b, dx_b, dp_b, dw_b = run(big, gl_b, gd_b)
    assert torch.equal(d_b, d_s.repeat(1, 1, H // h, W // w))
    assert torch.equal(l_b.view(H, W, K), l_s.view(h, w, K).repeat(H // h, W // w, 1))
    assert torch.equal(dx_b, dx_s.repeat(1, 1, H // h, W // w))
    _grad_close(dp_b, reps * dp_s, "dPrototypes (full size)")
    _grad_close(dw_b, reps * dw_s, "dLastLayer (full size)")


def test_argument_errors():
    """Error behaviour of the boundary: bad shapes raise SpxError (the C ABI returns non-zero with a message), nothing
    is launched."""
    import scaleprotoseg_amd as spx
    from scaleprotoseg_amd.functional import proto_head_forward

    dev = _dev()
    lay = _layout(40, 5, 4, 16, O.default_scale_ranges(40, 4))
    bank = torch.rand(40, 16, 1, 1, device=dev)
    head = torch.rand(5, 40, device=dev)
    with pytest.raises(spx.SpxError, match="channels"):
        proto_head_forward(torch.rand(1, 48, 4, 4, device=dev), bank, head, lay)
    with pytest.raises(spx.SpxError, match="bfloat16 or float32"):
        proto_head_forward(torch.rand(1, 64, 4, 4, device=dev).half(), bank, head, lay)
    with pytest.raises(spx.SpxError, match="head matrix"):
        proto_head_forward(torch.rand(1, 64, 4, 4, device=dev), bank, torch.rand(6, 40, device=dev), lay)
    with pytest.raises(spx.SpxError, match="empty input"):
        proto_head_forward(torch.rand(0, 64, 4, 4, device=dev), bank, head, lay)
    # maximum size: P * H * W must stay below 2^29 (32-bit byte offsets inside one image of the distance map)
    big = _layout(190, 19, 1, 16, {0: (0, 190)})
    with pytest.raises(spx.SpxError, match="too large"):
        proto_head_forward(torch.zeros(1, 16, 1700, 1700, device=dev, dtype=torch.bfloat16), torch.rand(190, 16, 1, 1, device=dev),
                           torch.rand(19, 190, device=dev), big)
    # P % num_scales != 0 with the reference's own scale table (Ps = P // S: 189 of 190 rows covered,
    # model_multiscale.py:146-149): its forward raises at F.linear, the plan is rejected here
    with pytest.raises(spx.SpxError):
        _layout(190, 19, 3, 16, {0: (0, 63), 1: (63, 126), 2: (126, 189)}).plan()


@pytest.mark.parametrize("shape", [(2, 4, 64, 228, 19, 9, 13, 3), (1, 1, 64, 210, 21, 8, 16, 3), (1, 4, 16, 40, 5, 9, 11, 3)])
@pytest.mark.parametrize("x_dtype", [torch.float32, torch.bfloat16])
def test_fused_group_tail(shape, x_dtype):
    """Grouping head with exp + last_layer_group inside the kernels (spx_dist_fwd_group / spx_dist_bwd_group) against
    the oracle's compute_group + last_layer_group (model_multiscale_group.py:283-308) and its autograd."""
    from scaleprotoseg_amd.functional import proto_head_forward

    dev = _dev()
    B, S, Cs, P, K, H, W, G = shape
    conv, bank, _, ident, ranges = _problem(B, S, Cs, P, K, H, W, seed=31)
    g = torch.Generator().manual_seed(17)
    idx = O.class_prototype_index(ident)
    idx = [i for i in idx if len(i) > 0]
    gw = [O.projection_simplex_sort(torch.rand(G, len(i), generator=g)) for i in idx]        # rows on the simplex
    gci = O.group_class_identity(ident, G)
    wg = (gci.t() - 0.5 * (1 - gci.t())) + 0.05 * torch.randn(gci.shape[1], gci.shape[0], generator=g)   # [K, G*K']
    U = G * len(idx)
    g_logits = torch.randn(B, H, W, K, generator=g) * 1e-3
    g_dist = torch.randn(B, P, H, W, generator=g) * 1e-3

    # oracle
    c0 = conv.clone().requires_grad_(True)
    p0 = bank.clone().requires_grad_(True)
    gw0 = [w.clone().requires_grad_(True) for w in gw]
    wg0 = wg.clone().requires_grad_(True)
    d_ref = O.scale_l2_convolution(c0, p0, ranges, S)
    act_ref = O.distance_2_similarity(d_ref).permute(0, 2, 3, 1).reshape(-1, P)
    units = torch.cat(O.compute_group(act_ref, ident, gw0), dim=-1)                           # exp(...) list -> [M, U]
    l_ref = torch.nn.functional.linear(units, wg0)
    ((l_ref * g_logits.reshape(-1, K)).sum() + (d_ref * g_dist).sum()).backward()

    # dense head [U, P] (zeros outside each class's prototype columns), as the module builds it
    wd = torch.zeros(U, P)
    for k, i in enumerate(idx):
        wd[k * G:(k + 1) * G, i] = gw[k]
    x = conv.to(dev, x_dtype).requires_grad_(True)
    pv = bank.to(dev).requires_grad_(True)
    wdd = wd.to(dev).requires_grad_(True)
    wgd = wg.to(dev).requires_grad_(True)
    logits, dist, _, gact = proto_head_forward(x, pv, wdd, _layout(P, U, S, Cs, ranges), group_tail=wgd)
    torch.cuda.synchronize()
    rl = l_ref.detach()
    err = (logits.detach().cpu() - rl).abs().max().item()
    assert err <= 1e-4 * max(1.0, rl.abs().max().item()), f"group logits err {err}"
    ug = units.detach()
    assert ((gact.cpu() - ug).abs() <= 2e-4 * (1 + ug.abs())).all(), "group activations"
    ((logits * g_logits.reshape(-1, K).to(dev)).sum() + (dist * g_dist.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    _grad_close(x.grad, c0.grad, "dX", tol=_dx_tol(x_dtype, ranges))
    _grad_close(pv.grad, p0.grad, "dPrototypes")
    _grad_close(wgd.grad, wg0.grad, "dLastLayerGroup")
    dwd_ref = torch.zeros(U, P)
    for k, i in enumerate(idx):
        dwd_ref[k * G:(k + 1) * G, i] = gw0[k].grad
    mask = torch.zeros(U, P, dtype=torch.bool)
    for k, i in enumerate(idx):
        mask[k * G:(k + 1) * G, i] = True
    _grad_close(wdd.grad.cpu() * mask, dwd_ref, "dGroupProjection")


# the reference's own evaluation shapes (eval_valid_multiscale.py:229-234: latent 129 x 257 -> 1024 x 2048 for Cityscapes, 65 x 65
# -> 513 x 513 for Pascal; a few channels of the 228 / 21 keep the CPU oracle small), then small and down-sampling maps
@pytest.mark.parametrize("case", [(1, 6, 129, 257, 1024, 2048, False), (2, 19, 65, 65, 513, 513, True), (1, 228, 17, 33, 129, 257, False),
                                  (2, 19, 9, 11, 70, 90, True), (1, 5, 4, 5, 4, 5, False), (1, 40, 33, 65, 17, 20, False),
                                  (1, 3, 16, 16, 64, 64, False)])
def test_upsample_argext(case):
    """Fused bilinear upsample + argmin/argmax against torch's F.interpolate + reduction on the CPU.
    The kernel follows, operation by operation, the arithmetic torch's CPU kernel runs at the reference's evaluation shapes
    (oracle::upsample_bilinear_restated, pinned to F.interpolate bit for bit in tests/test_oracle_golden.py): there - and
    wherever else the host kernel takes that code path - values AND indices are bit-identical, near-ties included.  For
    maps small enough that the host's TensorIterator picks another loop instantiation (its compiler contracted that one
    differently; the results differ in the last bit) the extremum is held to 1e-5 relative and the index must match
    wherever the runner-up is not within that tolerance."""
    from scaleprotoseg_amd.functional import upsample_argext

    dev = _dev()
    N, C, h, w, H, W, largest = case
    g = torch.Generator().manual_seed(77)
    src = torch.rand(N, C, h, w, generator=g) * 10
    idx_ref, val_ref, up = O.upsample_argext(src, (H, W), largest)
    idx, val = upsample_argext(src.to(dev), (H, W), largest)
    torch.cuda.synchronize()
    idx, val = idx.cpu(), val.cpu()
    same_path = torch.equal(up, O.upsample_bilinear_restated(src, (H, W)))
    if H * W >= 513 * 513:
        assert same_path, "this host's torch does not run the pinned arithmetic at the reference's evaluation shapes"
    if same_path:
        assert torch.equal(val, val_ref)
        assert torch.equal(idx, idx_ref)
        return
    tol = 1e-5 * (1 + val_ref.abs())
    assert ((val - val_ref).abs() <= tol).all(), (val - val_ref).abs().max().item()
    picked = torch.gather(up, 1, idx.unsqueeze(1)).squeeze(1)
    assert ((picked - val_ref).abs() <= 2 * tol).all()
    srt = torch.sort(up, dim=1, descending=largest).values
    clear = (srt[:, 1] - srt[:, 0]).abs() > 4 * tol if C > 1 else torch.ones_like(val_ref, dtype=torch.bool)
    assert torch.equal(idx[clear], idx_ref[clear])
    assert clear.float().mean() > 0.99


def _random_case(rng):
    S = int(rng.choice([1, 2, 3, 4]))
    Cs = int(rng.choice([16, 32, 48, 64, 80, 128]))
    K = int(rng.choice([2, 3, 5, 19, 21, 40, 64, 70]))
    per_scale = [int(rng.integers(1, 7)) * max(1, K // int(rng.choice([1, 2, 4]))) for _ in range(S)]
    per_scale = [min(p, 230) for p in per_scale]
    if rng.random() < 0.4:                       # unequal scales, as after the push's de-duplication
        per_scale = [max(1, p - int(rng.integers(0, 5))) for p in per_scale]
    B = int(rng.integers(1, 4))
    H, W = int(rng.integers(1, 24)), int(rng.integers(1, 40))
    if rng.random() < 0.3:
        W = 8 * int(rng.integers(1, 12))         # aligned rows: vector staging path
    if rng.random() < 0.2:                       # (drawn last: the other seeds keep their round-3 configurations)
        # H*W a multiple of 64 and scales wider than 64 channels: the LDS-DMA parameter kernel (bf16 features, K <= 32)
        H, W, Cs = 8 * int(rng.integers(1, 4)), 8 * int(rng.integers(1, 6)), int(rng.choice([80, 96, 128]))
    return B, S, Cs, per_scale, K, H, W


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_configurations(seed):
    """Randomised shapes (scale counts, unequal per-scale banks, channel widths with 16-channel tails, 1/2/5-block
    heads, odd and aligned grids, partial tiles, both feature dtypes): forward outputs and all gradients against the
    oracle, with the stated tolerances."""
    from scaleprotoseg_amd.functional import BankLayout, proto_head_forward

    dev = _dev()
    rng = np.random.default_rng(1000 + seed)
    B, S, Cs, per_scale, K, H, W = _random_case(rng)
    P = sum(per_scale)
    ranges, lo = {}, 0
    for s, n in enumerate(per_scale):
        ranges[s] = (lo, lo + n)
        lo += n
    g = torch.Generator().manual_seed(seed)
    conv = O.bf16_representable(torch.sigmoid(torch.randn(B, S * Cs, H, W, generator=g)))
    bank = O.bf16_representable(torch.rand(P, Cs, 1, 1, generator=g))
    Wl = torch.randn(K, P, generator=g) * 0.3
    x_dtype = torch.bfloat16 if seed % 2 else torch.float32
    g_logits = torch.randn(B, H, W, K, generator=g) * 1e-3
    g_dist = torch.randn(B, P, H, W, generator=g) * 1e-3
    g_act = torch.randn(B * H * W, P, generator=g) * 1e-3
    l_ref, d_ref, a_ref, dx_ref, dp_ref, dw_ref = O.fwd_bwd_reference(conv, bank, ranges, S, Wl, g_logits, g_dist, g_act)

    x = conv.to(dev, x_dtype).requires_grad_(True)
    pv = bank.to(dev).requires_grad_(True)
    w = Wl.to(dev).requires_grad_(True)
    lay = BankLayout(P, K, S, Cs, tuple(ranges[s] for s in range(S)))
    logits, dist, act = proto_head_forward(x, pv, w, lay, want_distances=True, want_activations=True)
    torch.cuda.synchronize()
    _assert_fwd(logits, dist, act, l_ref, d_ref, a_ref)
    ((logits * g_logits.reshape(-1, K).to(dev)).sum() + (dist * g_dist.to(dev)).sum() + (act * g_act.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    tag = f"B{B} S{S} Cs{Cs} P{per_scale} K{K} {H}x{W} {x_dtype}"
    _grad_close(x.grad, dx_ref, "dX " + tag, tol=_dx_tol(x_dtype, ranges))
    # G enters dX and dPrototypes as ONE fp16 plane scaled per tile by a power of two (11 significant bits per element)
    _grad_close(pv.grad, dp_ref, "dPrototypes " + tag)
    # d_W = dLogits^T . a: the activations cross HBM as block-scaled int16 codes and enter the MFMA as an exact bf16 hi + lo
    # pair, dLogits as split bf16
    _grad_close(w.grad, dw_ref, "dLastLayer " + tag)


# (40, 48, 84: the configurations that missed a 0.1 max|l| per-element floor in the round-2 fuzz run; 378: the worst of the
# round-3 run, 1.002e-4 of the per-element measure - all fixed regression cases of the bound stated in _assert_fwd)
@pytest.mark.parametrize("seed", list(range(12)) + [40, 48, 84, 378])
def test_random_gather_and_tail(seed):
    """Randomised shapes for the two extended modes: class-gathered distances (even seeds) and the fused grouping
    tail (odd seeds), forward + all gradients against the oracle."""
    from scaleprotoseg_amd.functional import BankLayout, ClassGather, class_gather_table, proto_head_forward

    dev = _dev()
    rng = np.random.default_rng(5000 + seed)
    S = int(rng.choice([1, 2, 4]))
    Cs = int(rng.choice([16, 32, 64]))
    K = int(rng.choice([2, 3, 5, 7, 19]))
    r = int(rng.integers(1, 4))
    P = K * S * r
    B, H, W = int(rng.integers(1, 3)), int(rng.integers(2, 14)), int(rng.integers(2, 30))
    x_dtype = torch.bfloat16 if (seed // 2) % 2 else torch.float32
    conv, bank, Wl, ident, ranges = _problem(B, S, Cs, P, K, H, W, seed=100 + seed)
    lay_k = BankLayout(P, K, S, Cs, tuple(ranges[s] for s in range(S)))
    g = torch.Generator().manual_seed(seed)
    g_logits = torch.randn(B, H, W, K, generator=g) * 1e-3
    x = conv.to(dev, x_dtype).requires_grad_(True)
    pv = bank.to(dev).requires_grad_(True)
    c0 = conv.clone().requires_grad_(True)
    p0 = bank.clone().requires_grad_(True)
    # tiny banks (P as small as 4): dX is a sum of a handful of signed terms G_p (x - p), each carrying G's bf16
    # rounding (2^-8); with cancellation between them the error relative to max|dX| can pass the 4e-3 of the
    # realistic shapes, so these toy cases get 6e-3
    dx_tol = _dx_tol(x_dtype, ranges)
    if seed % 2 == 0:
        target = _labels(B, H, W, K, seed=seed)
        lab0 = target.reshape(B, -1) - 1
        keys, J, table = class_gather_table(lay_k, ident, dev)
        gather = ClassGather(labels=lab0.to(dev, torch.int32).contiguous(), keys=keys, width=J, table=table)
        g_cls = torch.randn(B, H * W, J, generator=g) * 1e-3
        w0 = Wl.clone().requires_grad_(True)
        l_ref, d_ref, _ = O.forward_from_conv_features(c0, p0, ranges, S, w0)
        cd_ref = O.gather_class_distances(d_ref, lab0, ident)
        ((l_ref * g_logits).sum() + (cd_ref * g_cls).sum()).backward()
        w = Wl.to(dev).requires_grad_(True)
        logits, cd, _ = proto_head_forward(x, pv, w, lay_k, class_gather=gather)
        got = cd.detach().cpu().permute(0, 2, 1)
        assert ((got - cd_ref.detach()).abs() <= 1e-4 * (1 + cd_ref.detach())).all()
        _assert_fwd(logits, None, None, l_ref.detach(), None, None)
        ((logits * g_logits.reshape(-1, K).to(dev)).sum() + (cd * g_cls.permute(0, 2, 1).contiguous().to(dev)).sum()).backward()
        torch.cuda.synchronize()
        _grad_close(w.grad, w0.grad, "dLastLayer")
    else:
        G = int(rng.integers(2, 4))
        idx = [i for i in O.class_prototype_index(ident) if len(i) > 0]
        gw = [O.projection_simplex_sort(torch.rand(G, len(i), generator=g)) for i in idx]
        U = G * len(idx)
        wg = torch.randn(K, U, generator=g) * 0.5
        gw0 = [t.clone().requires_grad_(True) for t in gw]
        wg0 = wg.clone().requires_grad_(True)
        d_ref = O.scale_l2_convolution(c0, p0, ranges, S)
        act_ref = O.distance_2_similarity(d_ref).permute(0, 2, 3, 1).reshape(-1, P)
        units = torch.cat(O.compute_group(act_ref, ident, gw0), dim=-1)
        l_ref = torch.nn.functional.linear(units, wg0)
        (l_ref * g_logits.reshape(-1, K)).sum().backward()
        wd = torch.zeros(U, P)
        for k, i in enumerate(idx):
            wd[k * G:(k + 1) * G, i] = gw[k]
        wdd = wd.to(dev).requires_grad_(True)
        wgd = wg.to(dev).requires_grad_(True)
        logits, _, _, gact = proto_head_forward(x, pv, wdd, BankLayout(P, U, S, Cs, tuple(ranges[s] for s in range(S))),
                                                want_distances=False, group_tail=wgd)
        rl = l_ref.detach()
        err = (logits.detach().cpu() - rl).abs().max().item()
        assert err <= 1e-4 * max(1.0, rl.abs().max().item()), f"group logits err {err}"
        (logits * g_logits.reshape(-1, K).to(dev)).sum().backward()
        torch.cuda.synchronize()
        _grad_close(wgd.grad, wg0.grad, "dLastLayerGroup")
    _grad_close(x.grad, c0.grad, "dX", tol=dx_tol)
    _grad_close(pv.grad, p0.grad, "dPrototypes")



@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_kld_kernels_match_reference_golden(golden, tag):
    """KLDLoss on class-gathered GPU planes (csrc/spx_kld.hip: integer-atomic segment reductions + per-pixel gradient)
    against the reference's KLDLoss value and gradient (tests/golden/kld_loss.npz), and run-to-run bit-identical."""
    import scaleprotoseg_amd as spx
    from scaleprotoseg_amd.loss import class_slot_table, gather_class_distances

    dev = _dev()
    g = golden("kld_loss")
    t = torch.from_numpy(g[f"{tag}_target"])
    ident = torch.from_numpy(g[f"{tag}_ident"])
    S = int(g[f"{tag}_S"])
    ranges = {s: tuple(int(v) for v in g[f"{tag}_ranges"][s]) for s in range(S)}
    d_full = torch.from_numpy(g[f"{tag}_dist"])
    B, P, H, W = d_full.shape
    table = class_slot_table(ident)
    lab0 = t.reshape(B, -1) - 1
    planes = gather_class_distances(d_full, lab0, table).permute(0, 2, 1).contiguous()     # [B, J, HW]
    loss_fn = spx.KLDLoss(ident, S, ranges)

    def run():
        v = planes.to(dev).requires_grad_(True)
        cd = spx.ClassDistances(v, lab0.to(dev).int(), table.to(dev), (H, W))
        loss = loss_fn(cd, t.to(dev))
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().cpu(), v.grad.detach().cpu()

    l1, g1 = run()
    l2, g2 = run()
    assert torch.equal(l1, l2) and torch.equal(g1, g2)                     # integer atomics: order-independent
    assert abs(l1.item() - float(g[f"{tag}_loss"])) <= 2e-6, (l1.item(), float(g[f"{tag}_loss"]))
    # reference gradient lives on the full map: gather it the same way (other entries are exactly 0 there)
    gref_full = torch.from_numpy(g[f"{tag}_grad"])
    gref = gather_class_distances(gref_full, lab0, table).permute(0, 2, 1)
    assert float((gref_full.abs().sum() - gref.abs().sum()).abs()) <= 1e-6 * float(gref_full.abs().sum())
    scale = gref.abs().max().item()
    assert (g1 - gref).abs().max().item() <= 2e-5 * scale + 1e-9, ((g1 - gref).abs().max().item(), scale)


def test_kld_kernels_large_random():
    """2048 x 1024 pixels, random (non-uniform) and patchy labels: kernels vs the torch formulation of the same loss."""
    import scaleprotoseg_amd as spx

    dev = _dev()
    P, K, S, H, W = 190, 19, 1, 512, 1024
    ident = O.default_class_identity(P, K, S)
    lay = _layout(P, K, S, 16, O.default_scale_ranges(P, S))
    keys, J, table = spx.class_gather_table(lay, ident, dev)
    gen = torch.Generator(device=dev).manual_seed(5)
    for patchy in (False, True):
        if patchy:
            t = torch.randint(0, K + 1, (1, H // 32, W // 32), device=dev, generator=gen).repeat_interleave(32, 1).repeat_interleave(32, 2)
        else:
            t = torch.randint(0, K + 1, (1, H, W), device=dev, generator=gen)
        base = torch.rand(1, J, H * W, device=dev, generator=gen) * 40
        loss_fn = spx.KLDLoss(ident, S, {0: (0, P)})
        v1 = base.clone().requires_grad_(True)
        l1 = loss_fn(spx.ClassDistances(v1, (t.reshape(1, -1) - 1).int(), table, (H, W)), t)
        l1.backward()
        v2 = base.double().clone().requires_grad_(True)                      # fp64 -> the oracle's torch restatement
        with pytest.raises(spx.SpxError):
            loss_fn(spx.ClassDistances(v2, (t.reshape(1, -1) - 1).int(), table, (H, W)), t)      # the product has no other backend
        l2 = LO.kld_loss(loss_fn, spx.ClassDistances(v2, (t.reshape(1, -1) - 1).int(), table, (H, W)), t)
        l2.backward()
        torch.cuda.synchronize()
        assert abs(l1.item() - l2.item()) <= 1e-5 * max(1.0, abs(l2.item())), (l1.item(), l2.item())
        s = v2.grad.abs().max().item()
        assert (v1.grad.double() - v2.grad).abs().max().item() <= 1e-4 * s + 1e-12


@pytest.mark.parametrize("K,per", [(150, 12), (182, 12)])
def test_kld_kernels_with_the_ade_and_coco_class_counts(K, per):
    """scaleproto_ade.gin (150 classes x 12 prototypes) / scaleproto_coco.gin (182 x 12) with loss_weight_kld = 0.25: the
    [K, J, J] tables of the pair and gradient passes exceed the LDS and are tiled over class blocks (the loss had no
    backend there).  Value and gradient against the oracle's torch restatement in fp64, on the crops those configs train on."""
    import scaleprotoseg_amd as spx

    dev = _dev()
    P, S, B, H, W = K * per, 4, 2, 33, 65
    ident = O.default_class_identity(P, K, S)
    ranges = O.default_scale_ranges(P, S)
    lay = _layout(P, K if K <= 160 else 1, S, 16, ranges)       # (more than 160 classes: the head is not in the plan, as in the module)
    keys, J, table = spx.class_gather_table(lay, ident, dev)
    assert J == per
    gen = torch.Generator(device=dev).manual_seed(11)
    coarse = torch.randn(B, K + 1, 3, 5, device=dev, generator=gen)
    t = torch.nn.functional.interpolate(coarse, size=(H, W), mode="bilinear", align_corners=False).argmax(dim=1)     # irregular regions
    t[0, :4] = torch.randint(0, K + 1, (4, W), device=dev, generator=gen)                                            # and some salt
    base = torch.rand(B, J, H * W, device=dev, generator=gen) * 20
    loss_fn = spx.KLDLoss(ident, S, ranges)
    lab0 = (t.reshape(B, -1) - 1).int()
    v1 = base.clone().requires_grad_(True)
    l1 = loss_fn(spx.ClassDistances(v1, lab0, table, (H, W)), t)
    l1.backward()
    v2 = base.double().clone().requires_grad_(True)
    l2 = LO.kld_loss(loss_fn, spx.ClassDistances(v2, lab0, table, (H, W)), t)
    l2.backward()
    torch.cuda.synchronize()
    assert abs(l1.item() - l2.item()) <= 1e-5 * max(1.0, abs(l2.item())), (l1.item(), l2.item())
    sc = v2.grad.abs().max().item()
    assert sc > 0 and (v1.grad.double() - v2.grad).abs().max().item() <= 1e-4 * sc + 1e-12


def test_kld_group_kernels_match_reference_golden(golden):
    """KLDLossGroup on the GPU (class-gathered group activations through csrc/spx_kld.hip) against the reference's
    KLDLossGroup value and gradients (segmentation/model/loss.py:461-545), run-to-run bit-identical."""
    import scaleprotoseg_amd as spx

    dev = _dev()
    g = golden("kld_loss")
    n = int(g["grp_n"])
    t = torch.from_numpy(g["grp_target"]).to(dev)
    m = spx.KLDLossGroup(torch.from_numpy(g["grp_ident"]), torch.from_numpy(g["grp_gci"]), int(g["grp_G"]))

    def run():
        acts = [torch.from_numpy(g[f"grp_act{i}"]).to(dev).requires_grad_(True) for i in range(n)]
        loss = m(acts, t)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().cpu(), [a.grad.cpu() for a in acts]

    l1, g1 = run()
    l2, g2 = run()
    assert torch.equal(l1, l2) and all(torch.equal(a, b) for a, b in zip(g1, g2))
    assert abs(l1.item() - float(g["grp_loss"])) <= 2e-6, (l1.item(), float(g["grp_loss"]))
    for i, a in enumerate(g1):
        ref = torch.from_numpy(g[f"grp_grad{i}"])
        assert (a - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-9


@pytest.mark.parametrize("H,W", [(67, 333), (5, 64), (130, 257), (1, 1000)])
def test_kld_kernels_ragged_grids(H, W):
    """Pair-sum kernel on grids that do not fill its 256-column x 16-row strips (and the linear walk, W unknown):
    against the fp64 torch formulation of the same loss, value and gradient."""
    import scaleprotoseg_amd as spx
    from scaleprotoseg_amd import loss as L

    dev = _dev()
    P, K, S = 60, 6, 1
    ident = O.default_class_identity(P, K, S)
    lay = _layout(P, K, S, 16, O.default_scale_ranges(P, S))
    keys, J, table = spx.class_gather_table(lay, ident, dev)
    gen = torch.Generator(device=dev).manual_seed(H * 1000 + W)
    t = torch.randint(0, K + 1, (2, (H + 7) // 8, (W + 15) // 16), device=dev, generator=gen).repeat_interleave(8, 1).repeat_interleave(16, 2)[:, :H, :W].contiguous()
    base = torch.rand(2, J, H * W, device=dev, generator=gen) * 30
    loss_fn = spx.KLDLoss(ident, S, {0: (0, P)})
    lab = (t.reshape(2, -1) - 1).int()
    v2 = base.double().clone().requires_grad_(True)
    l2 = LO.kld_loss(loss_fn, spx.ClassDistances(v2, lab, table, (H, W)), t)
    l2.backward()
    for grid in ((H, W), (1, H * W)):               # column strips / one row (= the linear order)
        v1 = base.clone().requires_grad_(True)
        l1 = loss_fn(spx.ClassDistances(v1, lab, table, grid), t)
        l1.backward()
        torch.cuda.synchronize()
        assert abs(l1.item() - l2.item()) <= 1e-5 * max(1.0, abs(l2.item())), (grid, l1.item(), l2.item())
        s = v2.grad.abs().max().item()
        assert (v1.grad.double() - v2.grad).abs().max().item() <= 1e-4 * s + 1e-12
    # W = 0 through the C ABI gives the same sums as the column walk up to fp32 rounding of partial sums
    A_w = L.segment_pair_sums(base, lab, K, W)
    A_0 = L.segment_pair_sums(base, lab, K, 0)
    assert (A_w - A_0).abs().max().item() <= 1e-5 * (1.0 + A_0.abs().max().item())


@pytest.mark.parametrize("B,P,K,H,W,void", [(1, 3, 2, 1, 4, 0), (2, 8, 4, 2, 6, None), (3, 21, 7, 10, 36, 3), (1, 190, 19, 64, 260, 0), (2, 9, 3, 130, 64, None)])
def test_push_argmin_vector_path_edges(B, P, K, H, W, void):
    """H*W % 4 == 0 (the 4-pixel x 8-prototype kernel): fewer prototypes than a block, pixel ranges that end inside a
    workgroup's span, no void class, a void class in the middle, exact ties - bit-exact against the oracle."""
    from scaleprotoseg_amd.functional import push_masked_argmin

    dev = _dev()
    g = torch.Generator().manual_seed(B * 1000 + P)
    dist = torch.floor(torch.rand(B, P, H, W, generator=g) * 64) / 8          # many exact ties
    nlab = K if void is None else K + 1
    target = torch.randint(0, nlab, (B, H, W), generator=g)
    ident = torch.zeros(P, K)
    ident[torch.arange(P), torch.arange(P) % K] = 1
    if K > 2:
        ident[:, K - 1] = 0                                                   # a class no prototype belongs to
    ref_idx, ref_val = O.push_masked_argmin(dist, target, ident, K, void_class=void)
    idx, val = push_masked_argmin(dist.to(dev), target.to(dev), ident, void_class=void)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref_idx.numpy())
    np.testing.assert_array_equal(val.cpu().numpy(), ref_val.numpy())


def test_push_argmin_full_size_against_torch():
    """North-star map size (190 x 1024 x 2048): the kernel against torch's masked min on the GPU (same fp32 arithmetic)."""
    from scaleprotoseg_amd.functional import push_masked_argmin

    dev = _dev()
    P, K, H, W = 190, 19, 1024, 2048
    gen = torch.Generator(device=dev).manual_seed(3)
    dist = torch.rand(1, P, H, W, device=dev, generator=gen) * 40
    target = torch.randint(0, K + 1, (1, H, W), device=dev, generator=gen)
    ident = O.default_class_identity(P, K, 1).to(dev)
    idx, val = push_masked_argmin(dist, target, ident, void_class=0)
    cls = ident.argmax(dim=1)                                                 # one class per prototype here
    for p0 in range(0, P, 38):
        sl = slice(p0, p0 + 38)
        mask = (target.reshape(1, 1, -1) - 1 == cls[sl].view(1, -1, 1)).float()
        masked = dist[:, sl].reshape(1, 38, -1) + 1e10 * (1 - mask)
        ref = masked.min(dim=-1)
        assert torch.equal(val[:, sl], ref.values)
        # torch's GPU min does not promise the first index among exact ties: check the value at ours and minimality
        got = torch.gather(masked, 2, idx[:, sl].unsqueeze(-1)).squeeze(-1)
        assert torch.equal(got, ref.values)
        first = (masked == ref.values.unsqueeze(-1)).float().argmax(dim=-1)
        assert torch.equal(idx[:, sl], first)


class _Guarded:
    """A device buffer with guard bands on both sides: kernels get a pointer into the middle, the bands must stay intact."""
    GUARD = 64 * 1024          # bytes each side

    def __init__(self, nbytes, dev):
        self.n = int(nbytes)
        self.buf = torch.full((self.n + 2 * self.GUARD,), 0xA5, dtype=torch.uint8, device=dev)

    @property
    def ptr(self):
        return self.buf.data_ptr() + self.GUARD

    def body(self, dtype):
        return self.buf[self.GUARD:self.GUARD + self.n].view(dtype)

    def intact(self):
        g = self.GUARD
        return bool((self.buf[:g] == 0xA5).all()) and bool((self.buf[g + self.n:] == 0xA5).all())


@pytest.mark.parametrize("seed", list(range(16)))
def test_no_write_outside_the_output_buffers(seed):
    """Every output of the forward / pixel-side / bank-side backward / push through the C ABI, placed between guard
    bands, on random ragged shapes (partial tiles, odd H*W, 16-channel tails, padded prototype and class blocks):
    the kernels predicate with out-of-range buffer offsets, so nothing may land outside [ptr, ptr + size)."""
    import ctypes as C_
    from scaleprotoseg_amd import _lib
    from scaleprotoseg_amd.functional import _Packs, BankLayout

    dev = _dev()
    lib = _lib.load()
    rng = np.random.default_rng(3000 + seed)
    B, S, Cs, per_scale, K, H, W = _random_case(rng)
    P, HW, C = sum(per_scale), H * W, S * Cs
    ranges, lo = [], 0
    for n in per_scale:
        ranges.append((lo, lo + n))
        lo += n
    lay = BankLayout(P, K, S, Cs, tuple(ranges))
    plan = lay.plan()
    pp = C_.byref(plan)
    g = torch.Generator(device=dev).manual_seed(seed)
    xf32 = seed % 2 == 0
    x = torch.rand(B, C, HW, device=dev, generator=g)
    x = x if xf32 else x.bfloat16()
    esz = 4 if xf32 else 2
    bank = torch.rand(P, Cs, device=dev, generator=g)
    head = torch.randn(K, P, device=dev, generator=g) * 0.1
    packs = _Packs(plan, bank, head, True)
    s = _lib.stream_ptr()
    dist, act, logits = _Guarded(B * P * HW * 4, dev), _Guarded(B * HW * P * 4, dev), _Guarded(B * HW * K * 4, dev)
    _lib.check(lib.spx_dist_fwd(pp, _lib.ptr(x), 1 if xf32 else 0, B, HW, _lib.ptr(packs.bank), _lib.ptr(packs.p2), _lib.ptr(packs.head),
                                dist.ptr, act.ptr, logits.ptr, 1e-4, 0, s))
    scr = lib.spx_bwd_scratch_bytes(pp, B, HW)
    dx, gs, as_ = _Guarded(B * C * HW * esz, dev), _Guarded(scr, dev), _Guarded(lib.spx_bwd_head_scratch_bytes(pp, B, HW), dev)
    n_acc = lib.spx_bwd_dx_scratch_bytes(pp, 1 if xf32 else 0, B, HW)
    dacc = _Guarded(n_acc, dev) if n_acc else None
    gd = torch.randn(B, P, HW, device=dev, generator=g) * 1e-3
    ga = torch.randn(B * HW, P, device=dev, generator=g) * 1e-3
    gl = torch.randn(B * HW, K, device=dev, generator=g) * 1e-3
    _lib.check(lib.spx_dist_bwd(pp, _lib.ptr(x), 1 if xf32 else 0, B, HW, _lib.ptr(packs.bank), _lib.ptr(packs.bankT), _lib.ptr(packs.p2),
                                _lib.ptr(packs.headT), _lib.ptr(gd), _lib.ptr(ga), _lib.ptr(gl), dx.ptr, dacc.ptr if dacc else None, gs.ptr,
                                as_.ptr, 1e-4, 0, s))
    ws = _Guarded(lib.spx_bank_bwd_workspace_bytes(pp, B, HW), dev)
    d_bank, d_head = _Guarded(P * Cs * 4, dev), _Guarded(K * P * 4, dev)
    _lib.check(lib.spx_bank_bwd(pp, _lib.ptr(x), 1 if xf32 else 0, B, HW, _lib.ptr(bank), gs.ptr, as_.ptr, _lib.ptr(gl), d_bank.ptr,
                                d_head.ptr, ws.ptr, s))
    idx, val, scratch = _Guarded(B * P * 8, dev), _Guarded(B * P * 4, dev), _Guarded(B * P * 8, dev)
    labels = torch.randint(0, K + 1, (B, HW), device=dev, generator=g, dtype=torch.int32)
    ident = torch.zeros(P, K, device=dev)
    ident[torch.arange(P), torch.arange(P) % K] = 1
    _lib.check(lib.spx_push_argmin(dist.ptr, _lib.ptr(labels), _lib.ptr(ident), B, P, K, HW, 0, 1e10, idx.ptr, val.ptr, scratch.ptr, s))
    torch.cuda.synchronize()
    tag = f"B{B} S{S} Cs{Cs} P{per_scale} K{K} {H}x{W} {'f32' if xf32 else 'bf16'}"
    for name, gb in (("distances", dist), ("activations", act), ("logits", logits), ("dX", dx), ("G blob", gs), ("head scratch", as_),
                     ("bank workspace", ws), ("dBank", d_bank), ("dHead", d_head), ("push idx", idx), ("push val", val),
                     ("push scratch", scratch)) + ((("dX fp32 partials", dacc),) if dacc else ()):
        assert gb.intact(), f"{name}: write outside the buffer ({tag})"
    # and the buffers themselves were fully produced (no 0xA5 pattern left in the dense outputs)
    for name, gb in (("distances", dist), ("logits", logits), ("dBank", d_bank), ("dHead", d_head)):
        assert not bool((gb.body(torch.int32) == -1515870811).any()), f"{name}: unwritten words ({tag})"     # 0xA5A5A5A5


@pytest.mark.parametrize("seed", list(range(10)))
def test_no_write_outside_the_output_buffers_extended_ops(seed):
    """Guard bands around the outputs of the class-gathered forward / backward, the KLD passes and the fused
    evaluation map, on random ragged shapes."""
    import ctypes as C_
    from scaleprotoseg_amd import _lib
    from scaleprotoseg_amd.functional import _Packs, BankLayout, class_gather_table

    dev = _dev()
    lib = _lib.load()
    rng = np.random.default_rng(4000 + seed)
    S = int(rng.choice([1, 2, 4]))
    Cs = int(rng.choice([16, 32, 48, 64]))
    K = int(rng.choice([2, 3, 5, 19, 21]))
    r = int(rng.integers(1, 4))
    P = K * S * r
    B, H, W = int(rng.integers(1, 3)), int(rng.integers(1, 40)), int(rng.integers(1, 70))
    HW, C = H * W, S * Cs
    ranges = O.default_scale_ranges(P, S)
    lay = BankLayout(P, K, S, Cs, tuple(ranges[s] for s in range(S)))
    plan = lay.plan()
    pp = C_.byref(plan)
    ident = O.default_class_identity(P, K, S)
    keys, J, table = class_gather_table(lay, ident, dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.rand(B, C, HW, device=dev, generator=g).bfloat16()
    bank = torch.rand(P, Cs, device=dev, generator=g)
    head = torch.randn(K, P, device=dev, generator=g) * 0.1
    packs = _Packs(plan, bank, head, True)
    labels = torch.randint(-1, K, (B, HW), device=dev, generator=g, dtype=torch.int32)
    s = _lib.stream_ptr()
    cd, logits = _Guarded(B * J * HW * 4, dev), _Guarded(B * HW * K * 4, dev)
    cd.body(torch.float32).zero_()
    _lib.check(lib.spx_dist_fwd_cls(pp, _lib.ptr(x), 0, B, HW, _lib.ptr(packs.bank), _lib.ptr(packs.p2), _lib.ptr(packs.head),
                                    _lib.ptr(labels), _lib.ptr(keys), J, cd.ptr, None, logits.ptr, 1e-4, 0, s))
    scr = lib.spx_bwd_scratch_bytes(pp, B, HW)
    dx, gs, as_ = _Guarded(B * C * HW * 2, dev), _Guarded(scr, dev), _Guarded(lib.spx_bwd_head_scratch_bytes(pp, B, HW), dev)
    n_acc = lib.spx_bwd_dx_scratch_bytes(pp, 0, B, HW)
    dacc = _Guarded(n_acc, dev) if n_acc else None
    gcd = torch.randn(B, J, HW, device=dev, generator=g) * 1e-3
    gl = torch.randn(B * HW, K, device=dev, generator=g) * 1e-3
    _lib.check(lib.spx_dist_bwd_cls(pp, _lib.ptr(x), 0, B, HW, _lib.ptr(packs.bank), _lib.ptr(packs.bankT), _lib.ptr(packs.p2),
                                    _lib.ptr(packs.headT), _lib.ptr(labels), _lib.ptr(keys), J, _lib.ptr(gcd), None, _lib.ptr(gl),
                                    dx.ptr, dacc.ptr if dacc else None, gs.ptr, as_.ptr, 1e-4, 0, s))
    # KLD passes on the gathered planes
    kk, cnt, ssum = _Guarded(B * K * J * 4, dev), _Guarded(B * K * 4, dev), _Guarded(B * K * J * 8, dev)
    lse, afx, grad = _Guarded(B * K * J * 4, dev), _Guarded(B * K * J * J * 8, dev), _Guarded(B * J * HW * 4, dev)
    for z in (kk, cnt, ssum, afx):
        z.body(torch.uint8).zero_()
    Wk = W if seed % 2 else 0
    rngk, scale = _Guarded(8, dev), _Guarded(8, dev)
    rngk.body(torch.uint8).zero_()
    _lib.check(lib.spx_kld_segment_max(cd.ptr, _lib.ptr(labels), B, J, HW, Wk, K, kk.ptr, cnt.ptr, rngk.ptr, s))
    _lib.check(lib.spx_kld_segment_sumexp(cd.ptr, _lib.ptr(labels), B, J, HW, Wk, K, kk.ptr, ssum.ptr, s))
    _lib.check(lib.spx_kld_segment_lse(kk.ptr, ssum.ptr, B * K * J, lse.ptr, rngk.ptr, HW, scale.ptr, s))
    _lib.check(lib.spx_kld_pair_sums(cd.ptr, _lib.ptr(labels), B, J, HW, Wk, K, lse.ptr, scale.ptr, afx.ptr, s))
    A, E, Cf, kloss = (_Guarded(B * K * J * J * 4, dev), _Guarded(B * K * 2 * 8, dev), _Guarded(B * K * J * J * 4, dev),
                       _Guarded(8, dev))
    pair_ok = torch.triu(torch.ones(J, J, dtype=torch.uint8, device=dev), diagonal=1).repeat(K, 1, 1).contiguous()
    _lib.check(lib.spx_kld_gram_loss(afx.ptr, scale.ptr, cnt.ptr, _lib.ptr(pair_ok), B * K, K, J, A.ptr, Cf.ptr, E.ptr, kloss.ptr, s))
    _lib.check(lib.spx_kld_backward(cd.ptr, _lib.ptr(labels), B, J, HW, K, lse.ptr, A.ptr, Cf.ptr, kloss.ptr + 4, grad.ptr, s))
    # evaluation map: upsample the small map by a non-integer factor
    Ho, Wo = int(rng.integers(H, 5 * H + 3)), int(rng.integers(W, 5 * W + 3))
    src = torch.rand(B, P, H, W, device=dev, generator=g)
    eidx, eval_ = _Guarded(B * Ho * Wo * 8, dev), _Guarded(B * Ho * Wo * 4, dev)
    _lib.check(lib.spx_upsample_argext(_lib.ptr(src), B, P, H, W, Ho, Wo, 0, eidx.ptr, eval_.ptr, s))
    torch.cuda.synchronize()
    tag = f"B{B} S{S} Cs{Cs} P{P} K{K} J{J} {H}x{W} -> {Ho}x{Wo}"
    for name, gb in (("class distances", cd), ("logits", logits), ("dX", dx), ("G blob", gs), ("head scratch", as_), ("kld keys", kk),
                     ("kld counts", cnt), ("kld sums", ssum), ("kld lse", lse), ("kld pair sums", afx), ("kld grad", grad),
                     ("kld range keys", rngk), ("kld scale", scale), ("kld A", A), ("kld partials", E), ("kld Cf", Cf), ("kld loss", kloss),
                     ("eval idx", eidx), ("eval val", eval_)) + ((("dX fp32 partials", dacc),) if dacc else ()):
        assert gb.intact(), f"{name}: write outside the buffer ({tag})"


def _graph_problem(dev):
    from scaleprotoseg_amd.functional import BankLayout

    B, S, Cs, P, K, H, W = 2, 4, 64, 228, 19, 33, 33
    g = torch.Generator().manual_seed(31)
    x = torch.sigmoid(torch.randn(B, S * Cs, H, W, generator=g)).to(dev, torch.bfloat16).requires_grad_(True)
    bank = torch.rand(P, Cs, 1, 1, generator=g).to(dev).requires_grad_(True)
    head = (torch.randn(K, P, generator=g) * 0.1).to(dev).requires_grad_(True)
    per = P // S
    lay = BankLayout(P, K, S, Cs, tuple((s * per, (s + 1) * per) for s in range(S)))
    gl = (torch.randn(B * H * W, K, generator=g) * 1e-3).to(dev)
    gd = (torch.randn(B, P, H, W, generator=g) * 1e-3).to(dev)
    return x, bank, head, lay, gl, gd


def test_hip_graph_capture_replay():
    """The whole step (packs, forward, K1, K2, K3, .grad accumulation) captured into a HIP graph replays bit-identically,
    also after the static inputs were refilled in place.  Round 1's capture_end segfault (gpurun_out/seg.log) was the
    legacy default stream being pulled into the capture by AccumulateGrad nodes of a still-alive eager graph - stock
    torch ops alone reproduce it (tools/probes/capture_repro.py); capture_step captures on one side stream and refuses
    to start with such a graph alive."""
    from scaleprotoseg_amd.functional import proto_head_forward
    from scaleprotoseg_amd.graphs import capture_step

    dev = _dev()
    x, bank, head, lay, gl, gd = _graph_problem(dev)
    leaves = (x, bank, head)

    def step():
        for t in leaves:
            t.grad = None
        logits, d, _ = proto_head_forward(x, bank, head, lay)
        torch.autograd.backward([logits, d], [gl, gd])
        return logits.detach(), d.detach()

    def eager():
        lo, di = step()
        res = [lo.clone(), di.clone()] + [t.grad.clone() for t in leaves]
        for t in leaves:
            t.grad = None
        return res

    ref = eager()
    # nothing of the eager graph is kept alive: step() returns detached tensors and eager() drops the grads
    graph, (lo, di) = capture_step(step, warmup=2)
    for t in leaves:
        t.grad.zero_()
    graph.replay()
    torch.cuda.synchronize()
    got = [lo, di] + [t.grad for t in leaves]
    assert all(torch.equal(a, b) for a, b in zip(got, ref))
    # new data in the static buffers: replay == eager on the new data
    with torch.no_grad():
        x.copy_(torch.sigmoid(torch.randn_like(x, dtype=torch.float32)).to(x.dtype))
        bank.mul_(0.5)
    graph.replay()
    torch.cuda.synchronize()
    got = [t.clone() for t in [lo, di] + [t.grad for t in leaves]]
    del graph
    ref2 = eager()
    assert all(torch.equal(a, b) for a, b in zip(got, ref2))


@pytest.mark.parametrize("hw", [(8, 16), (5, 7)])       # the LDS-DMA and the register-staged parameter kernel
def test_features_beyond_the_fp16_range_give_finite_prototype_gradients(hw):
    """The d_bank product runs in fp16: a feature of 1e5 (representable in bf16) must saturate, not become inf (0 * inf = NaN
    in every prototype gradient)."""
    from scaleprotoseg_amd.functional import proto_head_forward

    dev = _dev()
    B, S, Cs, P, K = 1, 1, 128, 40, 5
    H, W = hw
    conv, bank, Wl, ident, ranges = _problem(B, S, Cs, P, K, H, W, seed=4)
    conv[0, 3, 1, 2] = 1.0e5
    conv[0, 70, 0, 0] = -3.0e5
    x = conv.to(dev, torch.bfloat16)
    pv = bank.to(dev).requires_grad_(True)
    w = Wl.to(dev).requires_grad_(True)
    logits, dist, _ = proto_head_forward(x, pv, w, _layout(P, K, S, Cs, ranges), want_distances=True)
    g = torch.Generator().manual_seed(3)
    ((logits * (torch.randn(B * H * W, K, generator=g) * 1e-3).to(dev)).sum() + (dist * 1e-9).sum()).backward()
    assert torch.isfinite(pv.grad).all() and torch.isfinite(w.grad).all()


def test_eager_forward_after_graph_replays_sees_the_updated_parameters():
    """A captured step with the optimizer inside edits bank and head in place (no version bump, same storage): the pack cache
    must not serve an eager forward the packs of the parameters as they were BEFORE the replays."""
    from scaleprotoseg_amd.functional import proto_head_forward
    from scaleprotoseg_amd.graphs import capture_step

    dev = _dev()
    B, S, Cs, P, K, H, W = 1, 1, 32, 40, 5, 8, 16
    conv, bank, Wl, ident, ranges = _problem(B, S, Cs, P, K, H, W, seed=9)
    lay = _layout(P, K, S, Cs, ranges)
    x = conv.to(dev)
    pv = bank.to(dev).requires_grad_(True)
    w = Wl.to(dev).requires_grad_(True)
    opt = torch.optim.SGD([pv, w], lr=0.5)
    g_logits = (torch.randn(B * H * W, K, generator=torch.Generator().manual_seed(1)) * 1e-2).to(dev)

    def step():
        opt.zero_grad(set_to_none=False)
        logits, _, _ = proto_head_forward(x, pv, w, lay, want_distances=False)
        (logits * g_logits).sum().backward()
        opt.step()

    with torch.no_grad():
        proto_head_forward(x, pv, w, lay, want_distances=True)       # an eager forward: fills the pack cache
    graph, _ = capture_step(step, warmup=1)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        logits, dist, _ = proto_head_forward(x, pv, w, lay, want_distances=True)
    # the oracle on the parameters AS THEY ARE NOW (rounded to bf16 as the kernels take the bank)
    ref_logits, ref_dist, _ = O.forward_from_conv_features(conv, O.bf16_representable(pv.detach().cpu()), ranges, S, w.detach().cpu())
    assert not torch.equal(pv.detach().cpu(), bank)                   # the optimizer did move them
    assert ((dist.cpu() - ref_dist).abs() <= 1e-4 * (1 + ref_dist)).all()
    rl = ref_logits.reshape(-1, K)
    assert (logits.cpu() - rl).abs().max() <= 1e-4 * max(1.0, rl.abs().max().item())


def test_capture_step_refuses_a_stale_default_stream_graph():
    """With an eager graph (default stream) alive, capture_step raises before any capture begins instead of letting
    hipStreamEndCapture crash the process."""
    from scaleprotoseg_amd import SpxError
    from scaleprotoseg_amd.functional import proto_head_forward
    from scaleprotoseg_amd.graphs import capture_step

    dev = _dev()
    x, bank, head, lay, gl, gd = _graph_problem(dev)

    def step():
        for t in (x, bank, head):
            t.grad = None
        logits, d, _ = proto_head_forward(x, bank, head, lay)
        torch.autograd.backward([logits, d], [gl, gd])

    keep = proto_head_forward(x, bank, head, lay)        # eager graph on the default stream, kept alive
    with pytest.raises(SpxError, match="still alive"):
        capture_step(step, warmup=1)
    # ... and a SECOND time in the same process (torch emits that warning through TORCH_WARN_ONCE: capture_step must not
    # depend on being the first to see it), with the global "warn always" switch left as it was
    before = torch.is_warn_always_enabled()
    with pytest.raises(SpxError, match="still alive"):
        capture_step(step, warmup=1)
    assert torch.is_warn_always_enabled() == before
    del keep
    graph, _ = capture_step(step, warmup=1)              # and with the graph gone the same step captures
    graph.replay()
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,n1,n2", [(1, 1, 1), (33, 19, 57), (42250, 19, 57), (100003, 32, 160), (5000, 5, 15), (70000, 150, 54)])
def test_pixel_outer_kernel(M, n1, n2):
    """spx_pixel_outer (d W_g = d_logits^T . g of the grouping tail) against an fp64 product; run-to-run identical."""
    from scaleprotoseg_amd.functional import _pixel_outer

    dev = _dev()
    g = torch.Generator().manual_seed(M + n1)
    a = torch.randn(M, n1, generator=g).to(dev)
    b = torch.rand(M, n2, generator=g).to(dev) * 3
    out = _pixel_outer(a, b)
    ref = (a.double().t() @ b.double()).float()
    scale = ref.abs().max().item() + 1e-30
    assert (out - ref).abs().max().item() <= 2e-5 * scale + 1e-6 * (M ** 0.5)
    assert torch.equal(out, _pixel_outer(a, b))


@pytest.mark.parametrize("H,W", [(2, 2), (1, 7), (3, 3), (1, 8), (5, 13)])
@pytest.mark.parametrize("x_dtype", [torch.float32, torch.bfloat16])
def test_tiny_and_ragged_grids(H, W, x_dtype):
    """Images of fewer than 8 pixels (element-wise staging), exactly one piece, and ragged ends below one tile."""
    from scaleprotoseg_amd.functional import proto_head_forward

    dev = _dev()
    shape = (2, 4, 16, 40, 5, H, W)
    B, S, Cs, P, K, _, _ = shape
    conv, bank, Wl, ident, ranges = _problem(*shape, seed=5)
    g = torch.Generator().manual_seed(9)
    g_logits = torch.randn(B, H, W, K, generator=g) * 1e-3
    g_dist = torch.randn(B, P, H, W, generator=g) * 1e-3
    rl, rd, _, dx_ref, dp_ref, dw_ref = O.fwd_bwd_reference(conv, bank, ranges, S, Wl, g_logits, g_dist)
    x = conv.to(dev, x_dtype).requires_grad_(True)
    pv = bank.to(dev).requires_grad_(True)
    w = Wl.to(dev).requires_grad_(True)
    logits, dist, _ = proto_head_forward(x, pv, w, _layout(P, K, S, Cs, ranges))
    _assert_fwd(logits, dist, None, rl, rd, None)
    torch.autograd.backward([logits, dist], [g_logits.reshape(-1, K).to(dev), g_dist.to(dev)])
    _grad_close(x.grad, dx_ref, "dX", tol=_dx_tol(x_dtype, ranges))
    _grad_close(pv.grad, dp_ref, "dPrototypes")
    _grad_close(w.grad, dw_ref, "dLastLayer")


@pytest.mark.parametrize("B,H,W", [(300, 3, 5), (140, 9, 11), (37, 13, 21)])
def test_many_small_images(B, H, W):
    """More images than the parameter-side kernel has slabs: its chunk walk then steps over whole images (step > chunks per
    image; the position advances incrementally, spx_bank.hip) and every slab sums chunks of several images."""
    from scaleprotoseg_amd.functional import proto_head_forward

    dev = _dev()
    shape = (B, 2, 16, 12, 3, H, W)
    _, S, Cs, P, K, _, _ = shape
    conv, bank, Wl, ident, ranges = _problem(*shape, seed=8)
    g = torch.Generator().manual_seed(10)
    g_logits = torch.randn(B, H, W, K, generator=g) * 1e-3
    g_dist = torch.randn(B, P, H, W, generator=g) * 1e-3
    rl, rd, _, dx_ref, dp_ref, dw_ref = O.fwd_bwd_reference(conv, bank, ranges, S, Wl, g_logits, g_dist)
    x = conv.to(dev, torch.bfloat16).requires_grad_(True)
    pv = bank.to(dev).requires_grad_(True)
    w = Wl.to(dev).requires_grad_(True)
    logits, dist, _ = proto_head_forward(x, pv, w, _layout(P, K, S, Cs, ranges))
    _assert_fwd(logits, dist, None, rl, rd, None)
    torch.autograd.backward([logits, dist], [g_logits.reshape(-1, K).to(dev), g_dist.to(dev)])
    _grad_close(x.grad, dx_ref, "dX", tol=_dx_tol(torch.bfloat16, ranges))
    _grad_close(pv.grad, dp_ref, "dPrototypes")
    _grad_close(w.grad, dw_ref, "dLastLayer")


@pytest.mark.parametrize("x_dtype,offset", [(torch.bfloat16, 1), (torch.bfloat16, 3), (torch.float32, 1)])
def test_features_at_an_unaligned_address(x_dtype, offset):
    """A feature tensor that starts at an odd element of its storage (2- / 4-byte aligned base): the vector staging path
    takes any alignment (gfx950 serves misaligned 16-B buffer accesses)."""
    from scaleprotoseg_amd.functional import proto_head_forward

    dev = _dev()
    shape = (1, 1, 64, 210, 21, 8, 16)
    B, S, Cs, P, K, H, W = shape
    conv, bank, Wl, ident, ranges = _problem(*shape, seed=6)
    g = torch.Generator().manual_seed(4)
    g_logits = torch.randn(B, H, W, K, generator=g) * 1e-3
    g_dist = torch.randn(B, P, H, W, generator=g) * 1e-3
    rl, rd, _, dx_ref, dp_ref, dw_ref = O.fwd_bwd_reference(conv, bank, ranges, S, Wl, g_logits, g_dist)
    store = torch.zeros(conv.numel() + 8, dtype=x_dtype, device=dev)
    store[offset:offset + conv.numel()] = conv.to(dev, x_dtype).reshape(-1)
    x = store[offset:offset + conv.numel()].view(conv.shape).requires_grad_(True)
    assert x.data_ptr() % 16 != 0
    pv = bank.to(dev).requires_grad_(True)
    w = Wl.to(dev).requires_grad_(True)
    logits, dist, _ = proto_head_forward(x, pv, w, _layout(P, K, S, Cs, ranges))
    _assert_fwd(logits, dist, None, rl, rd, None)
    gx, gp, gw = torch.autograd.grad([logits, dist], [x, pv, w], [g_logits.reshape(-1, K).to(dev), g_dist.to(dev)])
    _grad_close(gx, dx_ref, "dX", tol=_dx_tol(x_dtype, ranges))
    _grad_close(gp, dp_ref, "dPrototypes")
    _grad_close(gw, dw_ref, "dLastLayer")


@pytest.mark.parametrize("P,K,S,Cs,K2", [(190, 19, 1, 256, 0), (228, 57, 4, 64, 19), (1800, 150, 4, 64, 0), (37, 9, 2, 48, 5), (64, 0, 1, 64, 0)])
def test_pack_all_equals_the_single_pack_calls(P, K, S, Cs, K2):
    """spx_pack_all (one launch per step) leaves byte for byte what spx_pack_bank / spx_pack_head / spx_pack_group_tail /
    spx_pack_headT_units leave (include/spx_hip.h); K = 0: no head, K2 = 0: no grouping tail."""
    import ctypes as C
    from scaleprotoseg_amd import _lib
    from scaleprotoseg_amd.functional import BankLayout

    dev = _dev()
    lib = _lib.load()
    g = torch.Generator().manual_seed(P + K)
    per = P // S
    ranges = tuple((s * per, P if s == S - 1 else (s + 1) * per) for s in range(S))
    plan = BankLayout(P, max(K, 1), S, Cs, ranges).plan()
    pp = C.byref(plan)
    bank = torch.rand(P, Cs, generator=g).to(dev)
    W = torch.randn(K, P, generator=g).to(dev) if K else None
    Wg = torch.randn(K2, K, generator=g).to(dev) if K2 else None
    s = _lib.stream_ptr()

    def bufs():
        u8 = dict(dtype=torch.uint8, device=dev)
        mk = lambda n: torch.full((n,), 0xA5, **u8)
        out = dict(bank=mk(lib.spx_packed_bank_bytes(pp)), bankT=mk(lib.spx_packed_bankT_bytes(pp)), p2=mk(lib.spx_packed_p2_bytes(pp)))
        if K:
            out.update(head=mk(lib.spx_packed_head_bytes(pp)), headT=mk(lib.spx_packed_headT_bytes(pp)))
        if K2:
            out.update(tail=mk(lib.spx_packed_tail_bytes(pp)), tailT=mk(lib.spx_packed_tail_bytes(pp)))
        return out

    a, b = bufs(), bufs()
    _lib.check(lib.spx_pack_bank(pp, _lib.ptr(bank), _lib.ptr(a["bank"]), _lib.ptr(a["bankT"]), _lib.ptr(a["p2"]), s))
    if K:
        _lib.check(lib.spx_pack_head(pp, _lib.ptr(W), _lib.ptr(a["head"]), _lib.ptr(a["headT"]), s))
    if K2:
        _lib.check(lib.spx_pack_group_tail(pp, _lib.ptr(Wg), K2, _lib.ptr(a["tail"]), _lib.ptr(a["tailT"]), s))
        _lib.check(lib.spx_pack_headT_units(pp, _lib.ptr(W), _lib.ptr(a["headT"]), s))
    _lib.check(lib.spx_pack_all(pp, _lib.ptr(bank), _lib.ptr(W), _lib.ptr(Wg), K2, _lib.ptr(b["bank"]), _lib.ptr(b["bankT"]),
                                _lib.ptr(b["p2"]), _lib.ptr(b.get("head")), _lib.ptr(b.get("headT")), _lib.ptr(b.get("tail")),
                                _lib.ptr(b.get("tailT")), s))
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # forward-only form: the backward operands may be NULL
    c = bufs()
    _lib.check(lib.spx_pack_all(pp, _lib.ptr(bank), _lib.ptr(W), _lib.ptr(Wg), K2, _lib.ptr(c["bank"]), None, _lib.ptr(c["p2"]),
                                _lib.ptr(c.get("head")), None, _lib.ptr(c.get("tail")), None, s))
    torch.cuda.synchronize()
    for k in ("bank", "p2", "head", "tail"):
        if k in c:
            assert torch.equal(a[k], c[k]), k
    assert bool((c["bankT"] == 0xA5).all())


@pytest.mark.parametrize("S,H,W", [(1, 17, 40), (4, 33, 33), (2, 1, 5)])
def test_gathered_forward_writes_every_slot_of_uninitialised_planes(S, H, W):
    """spx_dist_fwd_cls on NaN-filled planes (include/spx_hip.h: the planes may arrive uninitialised): the slots past a class's
    prototype count and every slot of a pixel without a class read exactly 0 afterwards, the mapped ones the oracle's value.
    Classes own 3, 2, 1 and 0 prototypes per scale, so J = 3 S and most classes leave slots unmapped."""
    import ctypes as C_
    from scaleprotoseg_amd import _lib
    from scaleprotoseg_amd.functional import BankLayout, _Packs, class_gather_table

    dev = _dev()
    lib = _lib.load()
    K, Cs, B = 4, 32, 2
    per_scale = [0, 0, 0, 1, 1, 2]                       # classes of a scale's 6 prototypes: class 3 has none
    P = S * len(per_scale)
    ident = torch.zeros(P, K)
    for s_ in range(S):
        for i, c in enumerate(per_scale):
            ident[s_ * len(per_scale) + i, c] = 1
    ranges = O.default_scale_ranges(P, S)
    lay = BankLayout(P, K, S, Cs, tuple(ranges[s_] for s_ in range(S)))
    plan = lay.plan()
    pp = C_.byref(plan)
    keys, J, table = class_gather_table(lay, ident, dev)
    assert J == 3 * S
    g = torch.Generator().manual_seed(S * 100 + H)
    HW = H * W
    x = torch.rand(B, S * Cs, HW, generator=g).bfloat16()
    bank = torch.rand(P, Cs, generator=g).bfloat16().float()
    head = torch.randn(K, P, generator=g) * 0.1
    labels = torch.randint(-1, K + 1, (B, HW), generator=g, dtype=torch.int32)       # -1 and K: no class
    packs = _Packs(plan, bank.to(dev), head.to(dev), False)
    cd = torch.full((B, J, HW), float("nan"), device=dev)
    logits = torch.empty(B * HW, K, device=dev)
    _lib.check(lib.spx_dist_fwd_cls(pp, _lib.ptr(x.to(dev)), 0, B, HW, _lib.ptr(packs.bank), _lib.ptr(packs.p2), _lib.ptr(packs.head),
                                    _lib.ptr(labels.to(dev)), _lib.ptr(keys), J, _lib.ptr(cd), None, _lib.ptr(logits), 1e-4, 0,
                                    _lib.stream_ptr()))
    torch.cuda.synchronize()
    _, d_ref, _ = O.forward_from_conv_features(x.float().reshape(B, S * Cs, H, W), bank.reshape(P, Cs, 1, 1), ranges, S, head)
    ref = O.gather_class_distances(d_ref, labels.long(), ident)                       # [B, HW, J], zeros where nothing maps
    got = cd.cpu().permute(0, 2, 1)
    assert not torch.isnan(got).any(), "a slot was left unwritten"
    mapped = torch.zeros_like(ref, dtype=torch.bool)
    tab = O.class_slot_table(ident)
    for c in range(K):
        mapped[(labels == c)] = (tab[c] >= 0)
    assert (got[~mapped] == 0).all()
    err = (got - ref).abs()
    assert (err <= 1e-4 * (1 + ref)).all(), f"class distance err {err.max().item()}"


@pytest.mark.parametrize("P,S,Cs,U,K2,B,H,W", [(160, 1, 64, 160, 32, 1, 37, 41), (96, 2, 32, 150, 31, 2, 20, 33)])
def test_group_tail_with_cross_entropy_at_the_widest_heads(P, S, Cs, U, K2, B, H, W):
    """The grouping tail kernel with the fused cross entropy at its limits (160 units, 32 classes: 73 KiB of LDS for W_g^T, the
    group activations of 64 pixels and the logits): logits and loss against an fp32 torch restatement of
    model_multiscale_group.py:303-308 + loss.py:9-48 on the same bf16-representable inputs; every gradient finite."""
    from scaleprotoseg_amd.functional import BankLayout, proto_head_forward

    dev = _dev()
    torch.manual_seed(P + U)
    x = torch.sigmoid(torch.randn(B, S * Cs, H, W, device=dev)).bfloat16().requires_grad_(True)
    bank = torch.rand(P, Cs, 1, 1, device=dev).bfloat16().float().requires_grad_(True)
    wd = (torch.rand(U, P, device=dev) * 0.02).requires_grad_(True)
    wg = (torch.randn(K2, U, device=dev) * 0.1).requires_grad_(True)
    per = P // S
    lay = BankLayout(P, U, S, Cs, tuple((s * per, (s + 1) * per) for s in range(S)))
    lab = torch.randint(-1, K2, (B, H * W), device=dev, dtype=torch.int32)
    out = proto_head_forward(x, bank, wd, lay, want_distances=True, group_tail=wg, ce_labels=lab)
    logits, fce = out[0], out[3]
    xf, d = x.detach().float(), []
    for s in range(S):
        xs = xf[:, s * Cs:(s + 1) * Cs].reshape(B, Cs, -1)
        p = bank.detach()[s * per:(s + 1) * per].reshape(per, Cs)
        d.append(((xs * xs).sum(1, keepdim=True) - 2 * torch.einsum("pc,bcm->bpm", p, xs) + (p * p).sum(1)[None, :, None]).clamp_min(0))
    d = torch.cat(d, 1)
    act = torch.log((d + 1) / (d + 1e-4)).permute(0, 2, 1).reshape(B * H * W, P)
    ref = torch.exp(act @ wd.detach().t()) @ wg.detach().t()
    assert (logits - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    valid = lab.reshape(-1) >= 0
    ce_ref = torch.nn.functional.cross_entropy(ref[valid], lab.reshape(-1)[valid].long())
    assert abs(float(fce.loss.detach()) - float(ce_ref)) <= 1e-5 * max(1.0, abs(float(ce_ref)))
    fce.loss.backward()
    assert all(torch.isfinite(t.grad).all().item() for t in (x, bank, wd, wg))
