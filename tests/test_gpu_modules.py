"""GPU parity THROUGH the drop-in classes (not the bare operators): the nn.Module surface of
segmentation/model/model_multiscale.py, model_multiscale_group.py, model.py and the push routine of
segmentation/push_multiscale_optimization.py, against the reference-generated fixtures (tests/golden/*.npz,
oracle/gen_golden.py) and the CPU oracle.  Tolerances as in tests/test_gpu_parity.py."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import ppnet_oracle as O

GRAD_TOL = 1e-3          # gradients rel 1e-3 (max-normalised)
BF16_DX_TOL = 4e-3     # dX returned in bf16 (bf16 features): ONE output rounding is half a bf16 ulp = 2^-8 of the element


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


class _RoundBf16(nn.Module):
    """Test stand-in: makes the add-on output bf16-representable (SURVEY.md 8d 'identical inputs')."""

    def forward(self, x):
        return x.to(torch.bfloat16).to(torch.float32)


class _Backbone(nn.Module):
    """Stand-in for the DeepLab backbone (out of scope): ``str()`` starts with MSC and ``.base`` holds two Conv2d, which
    is all model_multiscale.py:153-171 asks of it.  ``mode``: identity (features are fed directly), pool (image ->
    features for the push tests) or list (MSC training output: a list of feature maps)."""

    def __init__(self, ch, mode="identity", stride=4):
        super().__init__()
        self.base = nn.Sequential(nn.Conv2d(3, ch, 1), nn.Conv2d(ch, ch, 1))
        self.pool = nn.AvgPool2d(stride)
        self.mode = mode

    def __repr__(self):
        return "MSC(standin)"

    def forward(self, x):
        if self.mode == "identity":
            return x
        if self.mode == "list":
            return [x, x[..., ::2, ::2].contiguous()]
        return self.base(self.pool(x))


def _close_fwd(got, ref, kind):
    got, ref = got.detach().float().cpu(), torch.as_tensor(ref)
    assert got.shape == ref.shape, f"{kind}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    err = (got - ref).abs()
    if kind == "distances":
        assert (err <= 1e-4 * (1 + ref)).all(), f"distance err {err.max().item()}"
    elif kind == "activations":
        assert (err <= 2e-4 * (1 + ref.abs())).all(), f"activation err {err.max().item()}"
    else:
        assert err.max().item() <= 1e-4 * max(1.0, ref.abs().max().item()), f"{kind} err {err.max().item()}"
        if kind == "logits":    # per element: 1e-4 relative with an absolute floor (a logit is a signed sum through zero)
            floor = 0.2 * max(1.0, ref.abs().max().item())
            assert (err <= 1e-4 * (ref.abs() + floor)).all(), f"per-element logit err {(err / (ref.abs() + floor)).max().item()}"


def _grad_close(got, ref, what, tol=GRAD_TOL):
    got, ref = got.detach().float().cpu(), torch.as_tensor(ref).float()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"{what}: err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.2e})"


def _proto_net(g, dev, cls=None, **kw):
    import scaleprotoseg_amd as spx

    P, K = g["class_identity"].shape
    Cs = g["prototype_vectors"].shape[1]
    S = int(g["num_scales"]) if "num_scales" in g.files else 1
    cls = cls or spx.PPNetMultiScale
    if cls is spx.PPNet:
        net = cls(_Backbone(Cs), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple", patch_classification=True, **kw)
    else:
        net = cls(_Backbone(Cs * S), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                  patch_classification=True, num_scales=S, **kw)
    with torch.no_grad():
        net.prototype_vectors.copy_(torch.from_numpy(g["prototype_vectors"]))
        if "last_layer_weight" in g.files:
            net.last_layer.weight.copy_(torch.from_numpy(g["last_layer_weight"]))
    return net.to(dev)


# ------------------------------------------------------------------------------------------------------------------
# (i) PPNetMultiScale.forward_from_conv_features: the three tuple modes + the MSC list recursion
#     (segmentation/model/model_multiscale.py:340-388)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["proto_ms_city", "proto_ms_small", "proto_s3", "proto_floor", "proto_s1_wide"])
def test_module_forward_three_tuple_modes(golden, name):
    dev = _dev()
    g = golden(name)
    net = _proto_net(g, dev)
    np.testing.assert_array_equal(net.prototype_class_identity.cpu().numpy(), g["class_identity"])
    conv = torch.from_numpy(g["conv"]).to(dev)

    out = net.forward_from_conv_features(conv)                                  # default: (logits, distances)
    assert isinstance(out, tuple) and len(out) == 2
    _close_fwd(out[0], g["logits"], "logits")
    _close_fwd(out[1], g["default_1"], "distances")

    out = net.forward_from_conv_features(conv, return_activations=True)         # (logits, activations)
    assert len(out) == 2
    _close_fwd(out[0], g["logits"], "logits")
    _close_fwd(out[1], g["act_1"], "activations")

    out = net.forward_from_conv_features(conv, return_activations=True, return_distances=True)
    assert len(out) == 3
    _close_fwd(out[0], g["logits"], "logits")
    _close_fwd(out[1], g["distances"], "distances")
    _close_fwd(out[2], g["activations"], "activations")

    out = net.forward_from_conv_features(conv, return_distances=True)           # falls to the else branch (:382)
    assert len(out) == 2
    _close_fwd(out[1], g["distances"], "distances")

    # forward(x) = conv_features -> forward_from_conv_features (:332-338); the stand-in backbone is the identity and
    # the deeplab_simple add-on a Sigmoid (:206-208), so the values differ from the fixture here: shapes only
    out = net(conv, return_activations=True, return_distances=True)
    assert len(out) == 3 and out[1].shape == g["distances"].shape

    # stand-alone methods of the surface
    d = net._scale_l2_convolution(conv)
    _close_fwd(d, g["distances"], "distances")
    sim = net.distance_2_similarity(d)
    ref_act = torch.from_numpy(g["activations"])
    got = sim.permute(0, 2, 3, 1).reshape(ref_act.shape)
    _close_fwd(got, ref_act, "activations")
    _close_fwd(net.run_last_layer(torch.from_numpy(g["activations"]).to(dev)).reshape(g["logits"].shape), g["logits"], "logits")


def test_module_msc_list_recursion(golden):
    """MSC training inputs: conv_features is a LIST; the recursion drops the flags (model_multiscale.py:358-359)."""
    dev = _dev()
    g = golden("proto_ms_small")
    net = _proto_net(g, dev)
    conv = torch.from_numpy(g["conv"]).to(dev)
    half = conv[..., ::2, ::2].contiguous()
    outs = net.forward_from_conv_features([conv, half], return_activations=True, return_distances=True)
    assert isinstance(outs, list) and len(outs) == 2
    assert all(isinstance(o, tuple) and len(o) == 2 for o in outs)        # flags dropped: (logits, distances)
    _close_fwd(outs[0][0], g["logits"], "logits")
    _close_fwd(outs[0][1], g["distances"], "distances")
    S = int(g["num_scales"])
    ranges = {s: tuple(int(v) for v in g["scale_ranges"][s]) for s in range(S)}
    rl, rd, _ = O.forward_from_conv_features(half.cpu(), torch.from_numpy(g["prototype_vectors"]), ranges, S,
                                             torch.from_numpy(g["last_layer_weight"]))
    _close_fwd(outs[1][0], rl, "logits")
    _close_fwd(outs[1][1], rd, "distances")

    # forward() over a list-valued backbone (:335-336) keeps the flags
    net.features.mode = "list"
    net.add_on_layers = nn.Sequential()             # features are fed as they are
    outs = net(conv, return_activations=True)
    assert isinstance(outs, list) and len(outs) == 2
    _close_fwd(outs[0][1], g["act_1"], "activations")
    pf = net.push_forward(conv)                     # :390-398, list form
    assert isinstance(pf, list) and torch.equal(pf[0][0], conv)
    _close_fwd(pf[0][1], g["distances"], "distances")
    net.features.mode = "identity"
    c2, d2 = net.push_forward(conv)
    assert torch.equal(c2, conv)
    _close_fwd(d2, g["distances"], "distances")
    _close_fwd(net.prototype_distances(conv), g["distances"], "distances")


@pytest.mark.parametrize("name", ["proto_ms_city", "proto_ms_small", "proto_s1_wide"])
def test_module_backward_matches_reference(golden, name):
    """loss = sum(logits r1) + sum(distances r2) + sum(activations r3) through the MODULE: parameter .grad of
    prototype_vectors and last_layer.weight and the feature gradient against the reference's autograd."""
    dev = _dev()
    g = golden(name)
    net = _proto_net(g, dev)
    conv = torch.from_numpy(g["conv"]).to(dev).requires_grad_(True)
    logits, dist, act = net.forward_from_conv_features(conv, return_activations=True, return_distances=True)
    loss = (
        (logits * torch.from_numpy(g["g_logits"]).to(dev)).sum()
        + (dist * torch.from_numpy(g["g_dist"]).to(dev)).sum()
        + (act * torch.from_numpy(g["g_act"]).to(dev)).sum()
    )
    loss.backward()
    _grad_close(conv.grad, g["d_conv"], "dX")
    _grad_close(net.prototype_vectors.grad, g["d_prototypes"], "dPrototypes")
    _grad_close(net.last_layer.weight.grad, g["d_last_layer"], "dLastLayer")
    assert net.ones.grad is None


# ------------------------------------------------------------------------------------------------------------------
# (ii) PPNetMultiScale of the group phase (model_multiscale_group.py:283-308, :404-452)
# ------------------------------------------------------------------------------------------------------------------
def _group_net(g, dev):
    from scaleprotoseg_amd.model_multiscale_group import PPNetMultiScale as GroupNet   # the reference's class name

    P, K = g["class_identity"].shape
    Cs = g["prototype_vectors"].shape[1]
    S, G = int(g["num_scales"]), int(g["num_groups"])
    net = GroupNet(_Backbone(Cs * S), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                   patch_classification=True, num_scales=S, num_groups=G)
    with torch.no_grad():
        net.prototype_vectors.copy_(torch.from_numpy(g["prototype_vectors"]))
        for i, gp in enumerate(net.group_projection):
            gp.weight.copy_(torch.from_numpy(g[f"group_w_{i}"]))
        net.last_layer_group.weight.copy_(torch.from_numpy(g["last_layer_group_weight"]))
    return net.to(dev)


def test_group_train_step_call_pattern_matches_reference(golden):
    """The group-phase trainer's step (module_multiscale_group_train.py:222-262): forward(return_activations, return_distances)
    -> compute_group(activations) -> cross entropy on the logits + KLDLossGroup on the list (+ a term on the distances),
    against the reference's own run of the same step (oracle/gen_golden.py::case_group_train_step): loss values and EVERY
    gradient.  compute_group hands out views of the group activations the fused kernel wrote; the KLD gradient on them
    reaches the kernels' backward as part of dUnits."""
    import scaleprotoseg_amd as spx

    dev = _dev()
    g = golden("group_train_step")
    net = _group_net(g, dev)
    K, G = g["class_identity"].shape[1], int(g["num_groups"])
    conv = torch.from_numpy(g["conv"]).to(dev).requires_grad_(True)
    target = torch.from_numpy(g["target"]).to(dev)
    logits, dist, act = net.forward_from_conv_features(conv, return_activations=True, return_distances=True)
    groups = net.compute_group(act)
    assert isinstance(groups, list) and len(groups) == len(net.group_projection) and all(t.shape == (act.shape[0], G) for t in groups)
    ref_cat = torch.from_numpy(g["group_cat"])
    assert ((torch.cat(groups, dim=-1).detach().cpu() - ref_cat).abs() <= 2e-4 * (1 + ref_cat.abs())).all()
    ce = spx.PixelWiseCrossEntropyLoss(ignore_index=-1)(predicted_logits=logits, target_labels=target)
    kld = spx.KLDLossGroup(net.prototype_class_identity, net.group_class_identity, G)(list_group_activation=groups, target_labels=target)
    assert abs(ce.item() - float(g["ce"])) <= 1e-5 * max(1.0, abs(float(g["ce"])))
    assert abs(kld.item() - float(g["kld"])) <= 2e-4 * max(1.0, abs(float(g["kld"])))
    loss = ce + float(g["w_kld"]) * kld + (dist * torch.from_numpy(g["g_dist"]).to(dev)).sum()
    loss.backward()
    _grad_close(conv.grad, g["d_conv"], "dX")
    _grad_close(net.prototype_vectors.grad, g["d_prototypes"], "dPrototypes")
    _grad_close(net.last_layer_group.weight.grad, g["d_last_layer_group"], "dLastLayerGroup")
    scale = max(np.abs(g[f"d_group_w_{i}"]).max() for i in range(len(net.group_projection)))
    for i, gp in enumerate(net.group_projection):
        err = (gp.weight.grad.cpu() - torch.from_numpy(g[f"d_group_w_{i}"])).abs().max().item()
        assert err <= GRAD_TOL * scale, f"d group_projection[{i}]: {err:.3e} vs {scale:.3e}"
    # the same list from activations that did NOT come out of the forward (no attachment): the product kernels + exp kernel
    groups2 = net.compute_group(act.detach().clone())
    assert ((torch.cat(groups2, dim=-1).cpu() - ref_cat).abs() <= 2e-4 * (1 + ref_cat.abs())).all()


def test_group_module_forward_backward_matches_reference(golden):
    dev = _dev()
    g = golden("group_ms_small")
    net = _group_net(g, dev)
    np.testing.assert_array_equal(net.group_class_identity.cpu().numpy(), g["group_class_identity"])
    conv = torch.from_numpy(g["conv"]).to(dev).requires_grad_(True)

    out = net.forward_from_conv_features(conv)
    assert len(out) == 2
    _close_fwd(out[0], g["logits"], "logits")
    _close_fwd(out[1], g["distances"], "distances")
    out = net.forward_from_conv_features(conv, return_activations=True)
    assert len(out) == 2
    _close_fwd(out[1], g["activations"], "activations")

    logits, dist, act = net.forward_from_conv_features(conv, return_activations=True, return_distances=True)
    _close_fwd(logits, g["logits"], "logits")
    _close_fwd(dist, g["distances"], "distances")
    _close_fwd(act, g["activations"], "activations")
    # compute_group returns a LIST of per-class [M, G] tensors (consumed by KLDLossGroup); run_last_layer on top
    groups = net.compute_group(act.detach())
    assert isinstance(groups, list) and len(groups) == len(net.group_projection)
    cat = torch.cat(groups, dim=-1)
    ref_cat = torch.from_numpy(g["group_cat"])
    assert ((cat.cpu() - ref_cat).abs() <= 2e-4 * (1 + ref_cat.abs())).all()
    _close_fwd(net.run_last_layer(act.detach()).reshape(g["logits"].shape), g["logits"], "logits")

    loss = (
        (logits * torch.from_numpy(g["g_logits"]).to(dev)).sum()
        + (dist * torch.from_numpy(g["g_dist"]).to(dev)).sum()
        + (act * torch.from_numpy(g["g_act"]).to(dev)).sum()
    )
    loss.backward()
    _grad_close(conv.grad, g["d_conv"], "dX")
    _grad_close(net.prototype_vectors.grad, g["d_prototypes"], "dPrototypes")
    _grad_close(net.last_layer_group.weight.grad, g["d_last_layer_group"], "dLastLayerGroup")
    scale = max(np.abs(g[f"d_group_w_{i}"]).max() for i in range(len(net.group_projection)))
    for i, gp in enumerate(net.group_projection):
        assert gp.weight.grad is not None and gp.weight.grad.shape == gp.weight.shape
        err = (gp.weight.grad.cpu() - torch.from_numpy(g[f"d_group_w_{i}"])).abs().max().item()
        assert err <= GRAD_TOL * scale, f"d group_projection[{i}]: {err:.3e} vs {scale:.3e}"


# ------------------------------------------------------------------------------------------------------------------
# (iii) single-scale PPNet (segmentation/model/model.py)
# ------------------------------------------------------------------------------------------------------------------
def test_ppnet_single_scale_matches_reference(golden):
    import scaleprotoseg_amd as spx

    dev = _dev()
    g = golden("ppnet_single")
    net = _proto_net(g, dev, cls=spx.PPNet)
    conv = torch.from_numpy(g["conv"]).to(dev)
    logits, dist = net.forward_from_conv_features(conv)
    _close_fwd(logits, g["logits"], "logits")
    _close_fwd(dist, g["distances"], "distances")
    logits, act = net.forward_from_conv_features(conv, return_activations=True, return_distances=True)   # model.py:357-360
    _close_fwd(logits, g["logits"], "logits")
    _close_fwd(act, g["activations"], "activations")
    _close_fwd(net._l2_convolution(conv), g["distances"], "distances")
    net.add_on_layers = nn.Sequential()              # the fixture's conv is the add-on OUTPUT: feed it as it is
    lg, ds, cv = net.forward_with_features(conv)     # model.py:317-326
    assert torch.equal(cv, conv)
    _close_fwd(ds, g["distances"], "distances")
    _close_fwd(lg, g["logits"], "logits")


# ------------------------------------------------------------------------------------------------------------------
# (v) prototype_activation_function = "linear" (model_multiscale.py:327-328), forward + backward
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 4, 16, 40, 5, 9, 11), (1, 1, 64, 210, 21, 8, 16), (1, 4, 64, 228, 57, 9, 13)])
@pytest.mark.parametrize("x_dtype", [torch.float32, torch.bfloat16])
def test_linear_activation_forward_backward(shape, x_dtype):
    from scaleprotoseg_amd.functional import BankLayout, proto_head_forward

    dev = _dev()
    B, S, Cs, P, K, H, W = shape
    gen = torch.Generator().manual_seed(20220227 + 11)
    conv = O.bf16_representable(torch.sigmoid(torch.randn(B, S * Cs, H, W, generator=gen)))
    bank = O.bf16_representable(torch.rand(P, Cs, 1, 1, generator=gen))
    Wl = 0.1 * torch.randn(K, P, generator=gen)
    ranges = O.default_scale_ranges(P, S)
    g_logits = torch.randn(B, H, W, K, generator=gen) * 1e-3
    g_dist = torch.randn(B, P, H, W, generator=gen) * 1e-3
    g_act = torch.randn(B * H * W, P, generator=gen) * 1e-3

    c = conv.clone().requires_grad_(True)
    pv = bank.clone().requires_grad_(True)
    w = Wl.clone().requires_grad_(True)
    rl, rd, ra = O.forward_from_conv_features(c, pv, ranges, S, w, activation="linear")
    ((rl * g_logits).sum() + (rd * g_dist).sum() + (ra * g_act).sum()).backward()

    x = conv.to(dev, x_dtype).requires_grad_(True)
    pvg = bank.to(dev).requires_grad_(True)
    wg = Wl.to(dev).requires_grad_(True)
    lay = BankLayout(P, K, S, Cs, tuple(ranges[s] for s in range(S)))
    logits, dist, act = proto_head_forward(x, pvg, wg, lay, want_distances=True, want_activations=True, activation="linear")
    _close_fwd(dist, rd.detach(), "distances")
    _close_fwd(act, ra.detach(), "activations")
    _close_fwd(logits.reshape(rl.shape), rl.detach(), "logits")
    ((logits * g_logits.reshape(-1, K).to(dev)).sum() + (dist * g_dist.to(dev)).sum() + (act * g_act.to(dev)).sum()).backward()
    _grad_close(x.grad, c.grad, "dX", tol=GRAD_TOL if x_dtype == torch.float32 else BF16_DX_TOL)
    _grad_close(pvg.grad, pv.grad, "dPrototypes")
    _grad_close(wg.grad, w.grad, "dLastLayer")


def test_linear_activation_through_the_module(golden):
    dev = _dev()
    g = golden("proto_ms_small")
    net = _proto_net(g, dev, prototype_activation_function="linear")
    conv = torch.from_numpy(g["conv"]).to(dev)
    logits, dist, act = net.forward_from_conv_features(conv, return_activations=True, return_distances=True)
    _close_fwd(dist, g["distances"], "distances")
    ref_act = -torch.from_numpy(g["distances"]).permute(0, 2, 3, 1).reshape(act.shape)
    _close_fwd(act, ref_act, "activations")
    _close_fwd(logits.reshape(-1, logits.shape[-1]), ref_act @ torch.from_numpy(g["last_layer_weight"]).t(), "logits")
    _close_fwd(net.distance_2_similarity(dist), -torch.from_numpy(g["distances"]), "activations")


# ------------------------------------------------------------------------------------------------------------------
# (vi) BASELINE.json configs without a reference fixture: EM (P=10, K=2, S=1, C=64, 512x512, batch 1) and the ADE
#      literal bank (P=1500, S=1) backward
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,x_dtype", [
    ((1, 1, 64, 10, 2, 512, 512), torch.float32),      # EM literal (BASELINE.json configs[0])
    ((1, 1, 64, 10, 2, 512, 512), torch.bfloat16),
    ((1, 4, 64, 24, 2, 64, 64), torch.float32),        # scaleproto_em.gin: 2 x 3 x 4 prototypes
    ((1, 1, 64, 20, 2, 33, 47), torch.float32),        # baseline_em.gin: 2 x 10
    ((1, 1, 64, 1500, 150, 6, 8), torch.float32),      # ADE literal 150 x 10, backward
    ((2, 1, 64, 1500, 150, 5, 7), torch.bfloat16),
])
def test_em_and_ade_literal_configs(shape, x_dtype):
    from scaleprotoseg_amd.functional import BankLayout, proto_head_forward

    dev = _dev()
    B, S, Cs, P, K, H, W = shape
    gen = torch.Generator().manual_seed(20220227 + 17)
    conv = O.bf16_representable(torch.sigmoid(torch.randn(B, S * Cs, H, W, generator=gen)))
    bank = O.bf16_representable(torch.rand(P, Cs, 1, 1, generator=gen))
    ident = O.default_class_identity(P, K, S)
    Wl = O.last_layer_init(ident) + 0.05 * torch.randn(K, P, generator=gen)
    ranges = O.default_scale_ranges(P, S)
    g_logits = torch.randn(B, H, W, K, generator=gen) * 1e-3
    g_dist = torch.randn(B, P, H, W, generator=gen) * 1e-3
    rl, rd, ra, dx_ref, dp_ref, dw_ref = O.fwd_bwd_reference(conv, bank, ranges, S, Wl, g_logits, g_dist)
    x = conv.to(dev, x_dtype).requires_grad_(True)
    pv = bank.to(dev).requires_grad_(True)
    w = Wl.to(dev).requires_grad_(True)
    lay = BankLayout(P, K, S, Cs, tuple(ranges[s] for s in range(S)))
    logits, dist, _ = proto_head_forward(x, pv, w, lay, want_distances=True)
    _close_fwd(dist, rd, "distances")
    _close_fwd(logits.reshape(rl.shape), rl, "logits")
    torch.autograd.backward([logits, dist], [g_logits.reshape(-1, K).to(dev), g_dist.to(dev)])
    _grad_close(x.grad, dx_ref, "dX", tol=GRAD_TOL if x_dtype == torch.float32 else BF16_DX_TOL)
    _grad_close(pv.grad, dp_ref, "dPrototypes")
    _grad_close(w.grad, dw_ref, "dLastLayer")


# ------------------------------------------------------------------------------------------------------------------
# (iv) the push, end to end through the module (push_multiscale_optimization.py:34-190, :323-335)
# ------------------------------------------------------------------------------------------------------------------
class _PushDataset:
    """len / [i] -> (image [3,h,w] float, target [h,w] int with 0 = void, 1..K); see scaleprotoseg_amd/push.py."""

    convert_targets = None

    def __init__(self, n, h, w, K, absent, seed):
        gen = torch.Generator().manual_seed(seed)
        self.items = []
        for i in range(n):
            img = torch.randn(3, h, w, generator=gen)
            t = torch.randint(0, K + 1, (h // 8, w // 8), generator=gen).repeat_interleave(8, 0).repeat_interleave(8, 1)
            t[t == absent + 1] = 0                          # class `absent` never appears anywhere
            self.items.append((img, t.numpy().astype(np.int64)))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def _push_problem(dev, S=4, Cs=16, K=5, per=2, n_img=6, seed=3):
    import scaleprotoseg_amd as spx

    P = S * K * per
    torch.manual_seed(seed)
    net = spx.PPNetMultiScale(_Backbone(S * Cs, mode="pool", stride=4), 64, (P, Cs, 1, 1), [], K,
                              add_on_layers_type="deeplab_simple", patch_classification=True, num_scales=S)
    net.add_on_layers = nn.Sequential(nn.Sigmoid(), _RoundBf16())
    with torch.no_grad():
        net.prototype_vectors.copy_(O.bf16_representable(net.prototype_vectors.data))
        # a forced duplicate: prototypes 0 and 1 (same class, same scale) start equal -> same winning pixel
        net.prototype_vectors[1].copy_(net.prototype_vectors[0])
    net = net.to(dev)
    data = _PushDataset(n_img, 40, 56, K, absent=3, seed=seed + 1)
    return net, data, P


def _oracle_push(net, data, P, S, K):
    """The reference pipeline restated with the oracle's pieces on the module's own conv features."""
    ident = net.prototype_class_identity.cpu()
    bank = net.prototype_vectors.detach().cpu()
    ranges = {s: tuple(net.scale_num_prototypes[s]) for s in range(S)}
    convs, idxs, vals = [], [], []
    dev = net.prototype_vectors.device
    with torch.no_grad():
        for i in range(len(data)):
            img, tgt = data[i]
            conv = net.conv_features(img.unsqueeze(0).to(dev)).cpu()
            d = O.scale_l2_convolution(conv, bank, ranges, S)
            lab = O.resize_label(tgt, (d.shape[3], d.shape[2])).unsqueeze(0)
            idx, val = O.push_masked_argmin(d, lab, ident, K, void_class=0)
            convs.append(conv); idxs.append(idx); vals.append(val)
    best = O.min_across_images(vals)
    new_bank = O.gather_push_patches(convs, best, idxs, S, P)
    dups = O.duplicate_prototypes(new_bank)
    keep, new_ranges = O.prune_state(dups, ranges, S, P)
    _, uniq = np.unique(new_bank, axis=0, return_index=True)
    return dict(best=best, idxs=idxs, vals=vals, bank=new_bank, dups=dups, keep=keep, ranges=new_ranges,
                unique=sorted(int(i) for i in uniq), convs=convs)


def test_push_stages_match_oracle():
    """compute_distances / min_across_dataset / global_min, one by one."""
    from scaleprotoseg_amd import push as push_mod

    dev = _dev()
    S, K = 4, 5
    net, data, P = _push_problem(dev, S=S, K=K)
    ref = _oracle_push(net, data, P, S, K)
    img, tgt = data[2]
    idx, val = push_mod.compute_distances(net, data, img, tgt, K, void_class=0)
    assert idx.dtype == torch.int64 and tuple(idx.shape) == (1, P)
    assert torch.equal(idx.cpu(), ref["idxs"][2])                      # indices bit-exact
    rv = ref["vals"][2]                                                 # values: the distance itself (1e10 exactly when absent)
    assert ((val.cpu() - rv).abs() <= 1e-4 * (1 + rv)).all() and torch.equal(val.cpu() == 1e10, rv == 1e10)
    best, list_idx = push_mod.min_across_dataset(data, net, K, void_class=0)
    assert torch.equal(best.cpu(), ref["best"])
    assert all(torch.equal(a.cpu(), b) for a, b in zip(list_idx, ref["idxs"]))
    # the absent class: value 1e10 in every image, flat index 0, image 0
    absent_rows = torch.nonzero(net.prototype_class_identity[:, 3]).flatten().tolist()
    for p in absent_rows:
        assert int(best[p]) == 0 and int(list_idx[0][0, p]) == 0
    patches = push_mod.global_min(best, list_idx, data, net)
    assert len(patches) == P and patches[0].shape == (net.prototype_shape[1], 1, 1)
    np.testing.assert_array_equal(np.reshape(patches, ref["bank"].shape), ref["bank"])


def test_compute_distances_takes_the_fused_minimum_and_falls_back_when_it_must():
    """``push.compute_distances`` takes the minimum inside the distance kernel (``push_min_distances``: no [1, P, H, W] map)
    whenever the class identity is one-hot, and reduces the written map otherwise - with the same result either way."""
    from scaleprotoseg_amd import push as push_mod

    dev = _dev()
    S, K = 4, 5
    net, data, P = _push_problem(dev, S=S, K=K)
    img, tgt = data[1]
    calls = []
    orig = net.push_min_distances

    def spy(*a, **kw):
        out = orig(*a, **kw)
        calls.append(out is not None)
        return out

    net.push_min_distances = spy
    idx_f, val_f = push_mod.compute_distances(net, data, img, tgt, K, void_class=0)
    assert calls == [True]                                              # the fused kernel ran
    # a fractional identity row: the mask is no longer a class comparison -> the module declines, the map path answers
    ident = net.prototype_class_identity.clone()
    net.prototype_class_identity = ident
    idx_m, val_m = push_mod.compute_distances(net, data, img, tgt, K, void_class=0)        # (still one-hot: fused again)
    assert torch.equal(idx_m, idx_f) and torch.equal(val_m, val_f)
    frac = ident.clone()
    frac[0] = frac[0] * 0.5
    net.prototype_class_identity = frac
    idx_h, val_h = push_mod.compute_distances(net, data, img, tgt, K, void_class=0)
    assert calls[-1] is False                                           # declined
    ref_conv = net.conv_features(img.unsqueeze(0).to(dev)).cpu()
    ranges = {s: tuple(net.scale_num_prototypes[s]) for s in range(S)}
    d = O.scale_l2_convolution(ref_conv, net.prototype_vectors.detach().cpu(), ranges, S)
    lab = O.resize_label(tgt, (d.shape[3], d.shape[2])).unsqueeze(0)
    ridx, rval = O.push_masked_argmin(d, lab, frac.cpu(), K, void_class=0)
    assert torch.equal(idx_h.cpu()[:, 1:], ridx[:, 1:])                 # rows with a binary mask: bit-exact as ever
    net.prototype_class_identity = ident
    assert torch.equal(idx_h[:, 1:], idx_f[:, 1:])


def test_push_prototypes_multiscale_end_to_end(tmp_path):
    from scaleprotoseg_amd.push import push_prototypes_multiscale

    dev = _dev()
    S, K = 4, 5
    net, data, P = _push_problem(dev, S=S, K=K)
    ref = _oracle_push(net, data, P, S, K)
    assert 1 in ref["dups"], "the forced duplicate must be dropped"
    w_before = net.last_layer.weight.detach().clone()
    ident_before = net.prototype_class_identity.clone()
    logs = []
    push_prototypes_multiscale(data, net, root_dir_for_saving_prototypes=str(tmp_path), log=logs.append)
    np.testing.assert_array_equal(net.prototype_vectors.detach().cpu().numpy(), ref["bank"][ref["keep"]])   # bit-exact
    assert {s: tuple(net.scale_num_prototypes[s]) for s in range(S)} == ref["ranges"]
    assert net.num_prototypes == len(ref["keep"]) and tuple(net.ones.shape) == tuple(net.prototype_vectors.shape)
    assert torch.equal(net.last_layer.weight.detach(), w_before[:, ref["keep"]])
    assert torch.equal(net.prototype_class_identity.cpu(), ident_before.cpu()[ref["keep"]])
    with open(os.path.join(tmp_path, "unique_prototypes.json")) as fp:
        assert json.load(fp) == ref["unique"]
    assert any("duplicate" in str(l) for l in logs)
    # the pruned module still runs (unequal per-scale ranges are data for the kernels) and matches the oracle
    conv = ref["convs"][0].to(dev)
    logits, dist = net.forward_from_conv_features(conv)
    rl, rd, _ = O.forward_from_conv_features(ref["convs"][0], net.prototype_vectors.detach().cpu(), ref["ranges"], S,
                                             net.last_layer.weight.detach().cpu())
    _close_fwd(dist, rd, "distances")
    _close_fwd(logits, rl, "logits")


# ------------------------------------------------------------------------------------------------------------------
# heads wider than the fused kernels carry (group_scaleproto_ade.gin: 150 classes x 3 groups = 450 units over a
# 1800-prototype bank in 4 scales; scaleproto_coco.gin: 182 classes): distances / activations from the kernel, the head
# through the fp32 MFMA product kernels on the activations (functional.wide_linear / wide_group_tail, csrc/spx_gemm.hip)
# ------------------------------------------------------------------------------------------------------------------
def test_ade_group_phase_wide_head_forward_backward():
    from scaleprotoseg_amd.model_multiscale_group import PPNetMultiScale as GroupNet

    dev = _dev()
    S, Cs, K, G, P, B, H, W = 4, 64, 150, 3, 1800, 1, 9, 8
    gen = torch.Generator().manual_seed(20220227 + 23)
    conv = O.bf16_representable(torch.sigmoid(torch.randn(B, S * Cs, H, W, generator=gen)))
    bank = O.bf16_representable(torch.rand(P, Cs, 1, 1, generator=gen))
    torch.manual_seed(5)
    net = GroupNet(_Backbone(S * Cs), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                   patch_classification=True, num_scales=S, num_groups=G)
    assert net.last_layer_group.weight.shape == (K, G * K) and len(net.group_projection) == K
    with torch.no_grad():
        net.prototype_vectors.copy_(bank)
        net.last_layer_group.weight.add_(0.02 * torch.randn(K, G * K, generator=gen))
    net = net.to(dev)
    ident = net.prototype_class_identity.cpu()
    gw = [gp.weight.detach().cpu().clone().requires_grad_(True) for gp in net.group_projection]
    wg = net.last_layer_group.weight.detach().cpu().clone().requires_grad_(True)
    c0 = conv.clone().requires_grad_(True)
    p0 = bank.clone().requires_grad_(True)
    ranges = {s: tuple(net.scale_num_prototypes[s]) for s in range(S)}
    rl, rd, ra = O.forward_from_conv_features(c0, p0, ranges, S, None, class_identity=ident, group_weights=gw,
                                              last_layer_group_weight=wg)
    g_logits = torch.randn(rl.shape, generator=gen) * 1e-3
    g_dist = torch.randn(rd.shape, generator=gen) * 1e-3
    ((rl * g_logits).sum() + (rd * g_dist).sum()).backward()

    x = conv.to(dev).requires_grad_(True)
    logits, dist = net.forward_from_conv_features(x)
    _close_fwd(dist, rd.detach(), "distances")
    _close_fwd(logits, rl.detach(), "logits")
    ((logits * g_logits.to(dev)).sum() + (dist * g_dist.to(dev)).sum()).backward()
    _grad_close(x.grad, c0.grad, "dX")
    _grad_close(net.prototype_vectors.grad, p0.grad, "dPrototypes")
    _grad_close(net.last_layer_group.weight.grad, wg.grad, "dLastLayerGroup")
    scale = max(w.grad.abs().max().item() for w in gw)
    for i, gp in enumerate(net.group_projection):
        err = (gp.weight.grad.cpu() - gw[i].grad).abs().max().item()
        assert err <= GRAD_TOL * scale, f"d group_projection[{i}]: {err:.3e} vs {scale:.3e}"
    out = net.forward_from_conv_features(x.detach(), return_activations=True, return_distances=True)
    assert len(out) == 3
    _close_fwd(out[2], ra.detach(), "activations")


@pytest.mark.parametrize("K,G", [(182, None), (50, 3)])
def test_wide_class_heads(K, G):
    """182 classes (scaleproto_coco.gin) on the prototype-phase module; 50 classes x 3 groups = 150 units with a 50-class
    tail (unit product in the kernel, tail in the fp32 MFMA product kernel) on the group-phase module."""
    import scaleprotoseg_amd as spx
    from scaleprotoseg_amd.model_multiscale_group import PPNetMultiScale as GroupNet

    dev = _dev()
    S, Cs, B, H, W = 4, 16, 2, 7, 9
    P = S * K * 1
    gen = torch.Generator().manual_seed(20220227 + 29)
    conv = O.bf16_representable(torch.sigmoid(torch.randn(B, S * Cs, H, W, generator=gen)))
    bank = O.bf16_representable(torch.rand(P, Cs, 1, 1, generator=gen))
    torch.manual_seed(6)
    if G is None:
        net = spx.PPNetMultiScale(_Backbone(S * Cs), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                                  patch_classification=True, num_scales=S)
    else:
        net = GroupNet(_Backbone(S * Cs), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                       patch_classification=True, num_scales=S, num_groups=G)
    with torch.no_grad():
        net.prototype_vectors.copy_(bank)
    net = net.to(dev)
    ranges = {s: tuple(net.scale_num_prototypes[s]) for s in range(S)}
    c0 = conv.clone().requires_grad_(True)
    p0 = bank.clone().requires_grad_(True)
    if G is None:
        w0 = net.last_layer.weight.detach().cpu().clone().requires_grad_(True)
        rl, rd, _ = O.forward_from_conv_features(c0, p0, ranges, S, w0)
    else:
        gw = [gp.weight.detach().cpu().clone() for gp in net.group_projection]
        w0 = net.last_layer_group.weight.detach().cpu().clone().requires_grad_(True)
        rl, rd, _ = O.forward_from_conv_features(c0, p0, ranges, S, None, class_identity=net.prototype_class_identity.cpu(),
                                                 group_weights=gw, last_layer_group_weight=w0)
    g_logits = torch.randn(rl.shape, generator=gen) * 1e-3
    (rl * g_logits).sum().backward()
    x = conv.to(dev).requires_grad_(True)
    logits, dist = net.forward_from_conv_features(x)
    _close_fwd(dist, rd.detach(), "distances")
    _close_fwd(logits, rl.detach(), "logits")
    (logits * g_logits.to(dev)).sum().backward()
    _grad_close(x.grad, c0.grad, "dX")
    _grad_close(net.prototype_vectors.grad, p0.grad, "dPrototypes")
    head = net.last_layer.weight if G is None else net.last_layer_group.weight
    _grad_close(head.grad, w0.grad, "dHead")


def test_user_supplied_similarity_function(golden):
    """A callable ``prototype_activation_function`` (segmentation/model/model_multiscale.py:329-330): the reference calls it
    on the [B, P, H, W] distance map and feeds the result to ``last_layer``.  Here: the distance kernel, the user's torch
    code, the fp32 MFMA product kernel - forward and backward against the same op sequence on the CPU."""
    dev = _dev()
    g = golden("proto_ms_small")

    def similarity(d):
        return 1.0 / (1.0 + d)

    net = _proto_net(g, dev, prototype_activation_function=similarity)
    conv = torch.from_numpy(g["conv"])
    S = int(g["num_scales"])
    P, K = g["class_identity"].shape
    ranges = {s: tuple(net.scale_num_prototypes[s]) for s in range(S)}
    c0 = conv.clone().requires_grad_(True)
    p0 = torch.from_numpy(g["prototype_vectors"]).clone().requires_grad_(True)
    w0 = torch.from_numpy(g["last_layer_weight"]).clone().requires_grad_(True)
    rd = O.scale_l2_convolution(c0, p0, ranges, S)
    ra = similarity(rd).permute(0, 2, 3, 1).reshape(-1, P)
    rl = torch.nn.functional.linear(ra, w0).reshape(rd.shape[0], rd.shape[2], rd.shape[3], K)
    gen = torch.Generator().manual_seed(77)
    g_logits = torch.randn(rl.shape, generator=gen) * 1e-3
    g_dist = torch.randn(rd.shape, generator=gen) * 1e-3
    ((rl * g_logits).sum() + (rd * g_dist).sum()).backward()

    x = conv.to(dev).requires_grad_(True)
    logits, dist, act = net.forward_from_conv_features(x, return_activations=True, return_distances=True)
    _close_fwd(dist, rd.detach(), "distances")
    _close_fwd(act, ra.detach(), "activations")
    _close_fwd(logits, rl.detach(), "logits")
    ((logits * g_logits.to(dev)).sum() + (dist * g_dist.to(dev)).sum()).backward()
    _grad_close(x.grad, c0.grad, "dX")
    _grad_close(net.prototype_vectors.grad, p0.grad, "dPrototypes")
    _grad_close(net.last_layer.weight.grad, w0.grad, "dLastLayer")
    assert len(net.forward_from_conv_features(x.detach())) == 2


# ------------------------------------------------------------------------------------------------------------------
# cross entropy (segmentation/model/loss.py:9-48): stand-alone HIP kernels pinned to the reference's golden, and the form
# fused into the logits epilogue / the backward's d_logits prologue
# ------------------------------------------------------------------------------------------------------------------
def test_cross_entropy_kernels_match_reference_golden(golden):
    from scaleprotoseg_amd.loss import PixelWiseCrossEntropyLoss

    dev = _dev()
    g = golden("kld_loss")
    lg = torch.from_numpy(g["ce_logits"]).to(dev).requires_grad_(True)
    tgt = torch.from_numpy(g["ce_target"]).to(dev)
    ce, correct = PixelWiseCrossEntropyLoss(ignore_index=-1, return_correct=True)(lg, tgt)
    ce.backward()
    assert abs(ce.item() - float(g["ce_loss"])) <= 1e-5 * max(1.0, abs(float(g["ce_loss"])))
    scale = np.abs(g["ce_grad"]).max()
    assert np.abs(lg.grad.cpu().numpy() - g["ce_grad"]).max() <= 1e-5 * scale
    np.testing.assert_array_equal(correct.cpu().numpy().astype(np.int64), g["ce_correct"])
    # exact ties: argmax takes the lowest class index (torch.argmax's contract)
    tie = torch.zeros(4, 7, device=dev)
    tie[1, 3] = tie[1, 5] = 2.0
    from scaleprotoseg_amd.functional import cross_entropy_from_logits
    out = cross_entropy_from_logits(tie, torch.tensor([0, 3, -1, 9], device=dev))
    assert out.pred.tolist() == [0, 3, 0, 0]
    ref = torch.nn.functional.cross_entropy(tie.cpu(), torch.tensor([0, 3, -1, -1]), ignore_index=-1)
    assert abs(out.loss.item() - ref.item()) <= 1e-6


@pytest.mark.parametrize("shape,x_dtype,gather", [
    ((2, 4, 64, 228, 19, 17, 19), torch.float32, False),
    ((1, 1, 256, 190, 19, 16, 64), torch.bfloat16, False),
    ((2, 4, 16, 40, 5, 9, 11), torch.float32, True),
    ((1, 1, 64, 210, 64, 8, 16), torch.float32, False),       # 64 classes: two class blocks
    ((1, 4, 64, 600, 150, 5, 8), torch.float32, False),       # 150 classes: five class blocks (strided logits path)
])
def test_fused_cross_entropy_through_the_module(shape, x_dtype, gather):
    import scaleprotoseg_amd as spx
    from scaleprotoseg_amd.loss import ClassDistances, KLDLoss, PixelWiseCrossEntropyLoss

    dev = _dev()
    B, S, Cs, P, K, H, W = shape
    gen = torch.Generator().manual_seed(20220227 + 41)
    conv = O.bf16_representable(torch.sigmoid(torch.randn(B, S * Cs, H, W, generator=gen)))
    bank = O.bf16_representable(torch.rand(P, Cs, 1, 1, generator=gen))
    net = spx.PPNetMultiScale(_Backbone(S * Cs), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                              patch_classification=True, num_scales=S)
    with torch.no_grad():
        net.prototype_vectors.copy_(bank)
        net.last_layer.weight.add_(0.05 * torch.randn(K, P, generator=gen))
    net = net.to(dev)
    target = torch.randint(0, K + 1, (B, H, W), generator=gen)         # 0 = void
    target[0, 0, :3] = 0
    ranges = {s: tuple(net.scale_num_prototypes[s]) for s in range(S)}
    g_dist = torch.randn(B, P, H, W, generator=gen) * 1e-3

    c0 = conv.clone().requires_grad_(True)
    p0 = bank.clone().requires_grad_(True)
    w0 = net.last_layer.weight.detach().cpu().clone().requires_grad_(True)
    rl, rd, _ = O.forward_from_conv_features(c0, p0, ranges, S, w0)
    ref_ce = torch.nn.functional.cross_entropy(rl.reshape(-1, K), target.reshape(-1) - 1, ignore_index=-1)
    ref_correct = (rl.reshape(-1, K).argmax(-1) == target.reshape(-1) - 1)[(target.reshape(-1) - 1) != -1]
    ident = net.prototype_class_identity.cpu()
    if gather:
        ref_loss = 0.7 * ref_ce + O.kld_loss(rd, target, ident, S, ranges)
    else:
        ref_loss = 0.7 * ref_ce + (rd * g_dist).sum()
    ref_loss.backward()

    x = conv.to(dev, x_dtype).requires_grad_(True)
    tgt = target.to(dev)
    if gather:
        logits, cd = net.forward_from_conv_features(x, target_labels=tgt, ce_target=tgt)
        assert isinstance(cd, ClassDistances)
    else:
        logits, dist = net.forward_from_conv_features(x, ce_target=tgt)
    assert hasattr(logits, "spx_ce")
    _close_fwd(logits, rl.detach(), "logits")
    ce, correct = PixelWiseCrossEntropyLoss(ignore_index=-1, return_correct=True)(logits, tgt)
    assert ce is logits.spx_ce.loss                                    # the epilogue's value, no second pass
    assert abs(ce.item() - ref_ce.item()) <= 1e-4 * max(1.0, abs(ref_ce.item()))
    agree = (correct.cpu() == ref_correct).float().mean().item()
    assert agree >= 0.995                                               # argmax may flip where two logits agree to 1e-4
    if gather:
        loss = 0.7 * ce + KLDLoss(net.prototype_class_identity, S, net.scale_num_prototypes)(cd, tgt)
    else:
        loss = 0.7 * ce + (dist * g_dist.to(dev)).sum()
    loss.backward()
    _grad_close(x.grad, c0.grad, "dX", tol=GRAD_TOL if x_dtype == torch.float32 else BF16_DX_TOL)
    _grad_close(net.prototype_vectors.grad, p0.grad, "dPrototypes")
    _grad_close(net.last_layer.weight.grad, w0.grad, "dLastLayer")


def test_fused_cross_entropy_with_a_second_use_of_the_logits():
    """The logits feed the fused cross entropy AND another term: both gradients reach d_logits."""
    import scaleprotoseg_amd as spx

    dev = _dev()
    B, S, Cs, P, K, H, W = 1, 4, 16, 40, 5, 9, 11
    gen = torch.Generator().manual_seed(3)
    conv = O.bf16_representable(torch.sigmoid(torch.randn(B, S * Cs, H, W, generator=gen)))
    net = spx.PPNetMultiScale(_Backbone(S * Cs), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                              patch_classification=True, num_scales=S)
    with torch.no_grad():
        net.prototype_vectors.copy_(O.bf16_representable(net.prototype_vectors.data))
    bank = net.prototype_vectors.detach().clone()
    net = net.to(dev)
    target = torch.randint(0, K + 1, (B, H, W), generator=gen)
    r = torch.randn(B, H, W, K, generator=gen) * 1e-2
    ranges = {s: tuple(net.scale_num_prototypes[s]) for s in range(S)}
    c0 = conv.clone().requires_grad_(True)
    p0 = bank.clone().requires_grad_(True)
    w0 = net.last_layer.weight.detach().cpu().clone().requires_grad_(True)
    rl, _, _ = O.forward_from_conv_features(c0, p0, ranges, S, w0)
    (torch.nn.functional.cross_entropy(rl.reshape(-1, K), target.reshape(-1) - 1, ignore_index=-1) + (rl * r).sum()).backward()
    x = conv.to(dev).requires_grad_(True)
    logits, _ = net.forward_from_conv_features(x, ce_target=target.to(dev))
    (logits.spx_ce.loss + (logits * r.to(dev)).sum()).backward()
    _grad_close(x.grad, c0.grad, "dX")
    _grad_close(net.prototype_vectors.grad, p0.grad, "dPrototypes")
    _grad_close(net.last_layer.weight.grad, w0.grad, "dLastLayer")


def test_fused_cross_entropy_is_dropped_when_the_labels_change():
    """The attachment is only valid for the label tensor it was computed from, AS IT WAS: another tensor or an in-place
    edit (version counter) sends the loss through the stand-alone kernels on the current labels."""
    import scaleprotoseg_amd as spx
    from scaleprotoseg_amd.loss import PixelWiseCrossEntropyLoss

    dev = _dev()
    B, S, Cs, P, K, H, W = 1, 2, 16, 20, 5, 6, 7
    gen = torch.Generator().manual_seed(9)
    conv = O.bf16_representable(torch.sigmoid(torch.randn(B, S * Cs, H, W, generator=gen))).to(dev)
    net = spx.PPNetMultiScale(_Backbone(S * Cs), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                              patch_classification=True, num_scales=S).to(dev)
    tgt = torch.randint(1, K + 1, (B, H, W), generator=gen).to(dev)
    crit = PixelWiseCrossEntropyLoss(ignore_index=-1)
    with torch.no_grad():
        logits, _ = net.forward_from_conv_features(conv, ce_target=tgt)
        assert crit(logits, tgt) is logits.spx_ce.loss
        ref = lambda t: torch.nn.functional.cross_entropy(logits.reshape(-1, K).cpu(), t.reshape(-1).cpu() - 1, ignore_index=-1).item()
        tgt2 = tgt.clone()
        tgt2[0, 0, :] = (tgt2[0, 0, :] % K) + 1                               # another tensor
        got = crit(logits, tgt2)
        assert got is not logits.spx_ce.loss and abs(got.item() - ref(tgt2)) <= 1e-5 * max(1.0, abs(ref(tgt2)))
        tgt[0, 1, :] = (tgt[0, 1, :] % K) + 1                                 # the same tensor, edited in place
        got = crit(logits, tgt)
        assert got is not logits.spx_ce.loss and abs(got.item() - ref(tgt)) <= 1e-5 * max(1.0, abs(ref(tgt)))


@pytest.mark.parametrize("B,H,W", [(2, 9, 11), (1, 40, 65)])
def test_group_module_fused_cross_entropy(golden, B, H, W):
    """Group phase: ce_target through the module (tail kernel: summed units -> exp -> W_g -> logits -> CE statistics),
    forward value, `correct`, and the gradients of CE + a distance term against the oracle's autograd."""
    from scaleprotoseg_amd.loss import PixelWiseCrossEntropyLoss
    from scaleprotoseg_amd.model_multiscale_group import PPNetMultiScale as GroupNet

    dev = _dev()
    S, Cs, K, G, P = 4, 16, 5, 3, 40
    gen = torch.Generator().manual_seed(20220227 + 53)
    conv = O.bf16_representable(torch.sigmoid(torch.randn(B, S * Cs, H, W, generator=gen)))
    torch.manual_seed(9)
    net = GroupNet(_Backbone(S * Cs), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                   patch_classification=True, num_scales=S, num_groups=G)
    with torch.no_grad():
        net.prototype_vectors.copy_(O.bf16_representable(net.prototype_vectors.data))
        net.last_layer_group.weight.add_(0.05 * torch.randn(K, G * K, generator=gen))
    bank = net.prototype_vectors.detach().clone()
    net = net.to(dev)
    target = torch.randint(0, K + 1, (B, H, W), generator=gen)
    ranges = {s: tuple(net.scale_num_prototypes[s]) for s in range(S)}
    g_dist = torch.randn(B, P, H, W, generator=gen) * 1e-3
    c0 = conv.clone().requires_grad_(True)
    p0 = bank.clone().requires_grad_(True)
    gw = [gp.weight.detach().cpu().clone().requires_grad_(True) for gp in net.group_projection]
    wg = net.last_layer_group.weight.detach().cpu().clone().requires_grad_(True)
    rl, rd, _ = O.forward_from_conv_features(c0, p0, ranges, S, None, class_identity=net.prototype_class_identity.cpu(),
                                             group_weights=gw, last_layer_group_weight=wg)
    ref_ce = torch.nn.functional.cross_entropy(rl.reshape(-1, K), target.reshape(-1) - 1, ignore_index=-1)
    (ref_ce + (rd * g_dist).sum()).backward()

    x = conv.to(dev).requires_grad_(True)
    tgt = target.to(dev)
    logits, dist = net.forward_from_conv_features(x, ce_target=tgt)
    _close_fwd(logits, rl.detach(), "logits")
    _close_fwd(dist, rd.detach(), "distances")
    ce, correct = PixelWiseCrossEntropyLoss(ignore_index=-1, return_correct=True)(logits, tgt)
    assert ce is logits.spx_ce.loss
    assert abs(ce.item() - ref_ce.item()) <= 1e-4 * max(1.0, abs(ref_ce.item()))
    (ce + (dist * g_dist.to(dev)).sum()).backward()
    _grad_close(x.grad, c0.grad, "dX")
    _grad_close(net.prototype_vectors.grad, p0.grad, "dPrototypes")
    _grad_close(net.last_layer_group.weight.grad, wg.grad, "dLastLayerGroup")
    scale = max(w.grad.abs().max().item() for w in gw)
    for i, gp in enumerate(net.group_projection):
        err = (gp.weight.grad.cpu() - gw[i].grad).abs().max().item()
        assert err <= GRAD_TOL * scale, f"d group_projection[{i}]: {err:.3e} vs {scale:.3e}"


def test_fused_cross_entropy_leaves_no_graph_alive():
    """The step's autograd graph must die with the step (an operator that kept an output on its ctx would form a
    node -> ctx -> tensor -> node cycle: a memory leak, and a stale graph that capture_step rightly refuses)."""
    import gc
    import scaleprotoseg_amd as spx
    from scaleprotoseg_amd.graphs import capture_step
    from scaleprotoseg_amd.loss import PixelWiseCrossEntropyLoss

    dev = _dev()
    gc.collect()
    gc.disable()                                        # the claim is about reference counts, not about the collector
    try:
        B, S, Cs, P, K, H, W = 2, 4, 16, 40, 5, 9, 11
        net = spx.PPNetMultiScale(_Backbone(S * Cs), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                                  patch_classification=True, num_scales=S).to(dev)
        net.add_on_layers = nn.Sequential()
        x = torch.rand(B, S * Cs, H, W, device=dev).requires_grad_(True)
        tgt = torch.randint(0, K + 1, (B, H, W), device=dev)
        lossf = PixelWiseCrossEntropyLoss(ignore_index=-1)

        def step():
            x.grad = None
            for p in net.parameters():
                p.grad = None
            logits, _ = net.forward_from_conv_features(x, ce_target=tgt)
            lossf(logits, tgt).backward()

        step()
        step()
        graph, _ = capture_step(step, warmup=1)          # raises SpxError if an eager graph is still alive
        graph.replay()
        torch.cuda.synchronize()
    finally:
        gc.enable()


def test_group_step_leaves_no_graph_alive():
    """The same for the group phase, whose kernel output exp(units) is differentiable (compute_group's list) and must reach the
    backward through save_for_backward, not as a ctx attribute (round 4: that cycle made capture_step refuse the step)."""
    import gc
    from scaleprotoseg_amd.graphs import capture_step
    from scaleprotoseg_amd.model_multiscale_group import PPNetMultiScale as GroupNet

    dev = _dev()
    gc.collect()
    gc.disable()
    try:
        B, S, Cs, P, K, H, W = 2, 4, 16, 40, 5, 9, 11
        net = GroupNet(_Backbone(S * Cs), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                       patch_classification=True, num_scales=S, num_groups=3).to(dev)
        net.add_on_layers = nn.Sequential()
        x = torch.rand(B, S * Cs, H, W, device=dev).requires_grad_(True)
        gl = torch.randn(B, H, W, K, device=dev) * 1e-3

        def step():
            x.grad = None
            for p in net.parameters():
                p.grad = None
            logits, dist, act = net.forward_from_conv_features(x, return_activations=True, return_distances=True)
            groups = net.compute_group(act)
            ((logits * gl).sum() + 1e-3 * torch.cat(groups, dim=1).sum()).backward()

        step()
        step()
        graph, _ = capture_step(step, warmup=1)          # raises SpxError if an eager graph is still alive
        graph.replay()
        torch.cuda.synchronize()
    finally:
        gc.enable()


@pytest.mark.gpu
@pytest.mark.parametrize("P,K,G,drop", [(228, 19, 3, ()), (60, 7, 2, (1, 4)), (1800, 150, 3, ())])
def test_group_dense_matrix_kernel_matches_index_put(P, K, G, drop):
    """spx_group_dense / spx_group_dense_bwd (the dense [U, P] form of the per-class group projections in one launch each,
    model_multiscale_group.py:249-269) against torch's zeros + cat + index_put and its autograd; classes in ``drop`` own no
    prototype (pruned away): their columns / rows do not exist."""
    from scaleprotoseg_amd.model_multiscale_group import PPNetMultiScale as GroupNet

    dev = torch.device("cuda:0")
    torch.manual_seed(P + K)
    net = GroupNet(_Backbone(64), 64, (P, 16, 1, 1), [], K, add_on_layers_type="deeplab_simple", patch_classification=True,
                   num_scales=4, num_groups=G).to(dev)
    if drop:
        keep = [p for p in range(P) if int(net.prototype_class_identity[p].argmax()) not in drop]
        net.prune_prototypes([p for p in range(P) if p not in keep])
    rows, cols, ng, tables = net._group_index(dev)
    assert tables is not None
    ws = [gp.weight for gp in net.group_projection]
    wd = net._dense_group_matrix()
    ref = torch.zeros(ng, net.num_prototypes, device=dev).index_put((rows, cols), torch.cat([w.reshape(-1) for w in ws]))
    assert torch.equal(wd, ref)
    g = torch.randn_like(wd)
    got = torch.autograd.grad(wd, ws, g)
    want = torch.autograd.grad(ref, ws, g)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
