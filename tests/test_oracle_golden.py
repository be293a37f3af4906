"""Pin the CPU oracle (oracle/ppnet_oracle.py) to golden vectors produced by the
reference's own classes (oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import ppnet_oracle as O

PROTO_CASES = ["proto_ms_small", "proto_ms_city", "proto_s3", "proto_floor", "proto_s1_wide"]


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _ranges(g):
    return {s: (int(lo), int(hi)) for s, (lo, hi) in enumerate(g["scale_ranges"])}


@pytest.mark.parametrize("name", PROTO_CASES)
def test_forward_matches_reference(golden, name):
    g = golden(name)
    S = int(g["num_scales"])
    logits, d, act = O.forward_from_conv_features(
        _t(g["conv"]), _t(g["prototype_vectors"]), _ranges(g), S, _t(g["last_layer_weight"])
    )
    # same op order as the reference -> agreement to fp32 rounding
    np.testing.assert_allclose(d.numpy(), g["distances"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(act.numpy(), g["activations"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=1e-5, atol=1e-5)
    # tuple quirk, model_multiscale.py:378-385: default -> distances, return_activations alone -> activations
    np.testing.assert_array_equal(g["default_1"], g["distances"])
    np.testing.assert_array_equal(g["act_1"], g["activations"])


@pytest.mark.parametrize("name", PROTO_CASES)
def test_backward_matches_reference(golden, name):
    g = golden(name)
    S = int(g["num_scales"])
    _, _, _, dx, dp, dw = O.fwd_bwd_reference(
        _t(g["conv"]), _t(g["prototype_vectors"]), _ranges(g), S, _t(g["last_layer_weight"]),
        _t(g["g_logits"]), _t(g["g_dist"]), _t(g["g_act"]),
    )
    for got, key in ((dx, "d_conv"), (dp, "d_prototypes"), (dw, "d_last_layer")):
        ref = g[key]
        scale = np.abs(ref).max() + 1e-30
        assert np.abs(got.numpy() - ref).max() <= 1e-5 * scale, key


@pytest.mark.parametrize("name", PROTO_CASES)
def test_default_layouts(golden, name):
    g = golden(name)
    S = int(g["num_scales"])
    P, K = g["class_identity"].shape
    np.testing.assert_array_equal(O.default_class_identity(P, K, S).numpy(), g["class_identity"])
    r = O.default_scale_ranges(P, S)
    np.testing.assert_array_equal(np.array([r[s] for s in range(S)]), g["scale_ranges"])
    assert list(g["state_keys"]) == ["prototype_vectors", "ones", "last_layer.weight"]


def test_group_head(golden):
    g = golden("group_ms_small")
    S = int(g["num_scales"])
    G = int(g["num_groups"])
    ident = _t(g["class_identity"])
    n_cls = len(O.class_prototype_index(ident))
    gw = [_t(g[f"group_w_{i}"]) for i in range(n_cls)]
    logits, d, act = O.forward_from_conv_features(
        _t(g["conv"]), _t(g["prototype_vectors"]), _ranges(g), S, None,
        class_identity=ident, group_weights=gw, last_layer_group_weight=_t(g["last_layer_group_weight"]),
    )
    np.testing.assert_allclose(d.numpy(), g["distances"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=2e-5, atol=2e-5)
    cat = torch.cat(O.compute_group(act, ident, gw), dim=-1)
    np.testing.assert_allclose(cat.numpy(), g["group_cat"], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(O.group_class_identity(ident, G).numpy(), g["group_class_identity"])
    keys = list(g["state_keys"])
    assert "last_layer.weight" not in keys and "last_layer_group.weight" in keys
    assert keys[:2] == ["prototype_vectors", "ones"]
    # phase-1 -> phase-2 load_state_dict(strict=False): finetune_wandb_group.py:76
    assert list(g["unexpected_keys"]) == ["last_layer.weight"]
    # rows of every group projection live on the simplex after init
    for w in gw:
        np.testing.assert_allclose(w.sum(1).numpy(), 1.0, atol=1e-5)
        assert (w >= 0).all()


def test_single_scale_is_s1(golden):
    g = golden("ppnet_single")
    P = g["prototype_vectors"].shape[0]
    logits, d, act = O.forward_from_conv_features(
        _t(g["conv"]), _t(g["prototype_vectors"]), {0: (0, P)}, 1, _t(g["last_layer_weight"])
    )
    np.testing.assert_allclose(d.numpy(), g["distances"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(act.numpy(), g["activations"], rtol=1e-6, atol=1e-6)


def test_push_argmin(golden):
    g = golden("push_argmin")
    idx, val = O.push_masked_argmin(
        _t(g["distances"]), _t(g["target"])[None], _t(g["class_identity"]), int(g["num_classes"]), void_class=0
    )
    np.testing.assert_array_equal(idx.numpy(), g["indices"])
    np.testing.assert_array_equal(val.numpy(), g["values"])
    # the fixture holds an absent class (-> index 0, value 1e10) and an exact tie (-> lowest flat index)
    assert (g["values"] == np.float32(1e10)).any()
    assert (g["indices"][g["values"] == np.float32(1e10)] == 0).all()
    W = g["distances"].shape[-1]
    assert g["indices"][0, 4] == 2 * W + 3 and g["values"][0, 4] == np.float32(0.125)


def test_resize_label(golden):
    g = golden("push_argmin")
    for w, h in g["resize_sizes"]:
        out = O.resize_label(g["label_full"], (int(w), int(h)))
        np.testing.assert_array_equal(out.numpy(), g[f"resized_{w}x{h}"])


def test_simplex_and_prune(golden):
    g = golden("misc")
    np.testing.assert_allclose(O.projection_simplex_sort(_t(g["simplex_in"])).numpy(), g["simplex_out"], atol=1e-6)
    P, S = int(g["prune_P"]), int(g["prune_S"])
    keep, ranges = O.prune_state(list(g["prune_drop"]), O.default_scale_ranges(P, S), S, P)
    np.testing.assert_array_equal(np.array([ranges[s] for s in range(S)]), g["prune_after_ranges"])
    np.testing.assert_array_equal(g["prune_before_protos"][keep], g["prune_after_protos"])
    np.testing.assert_array_equal(g["prune_before_last"][:, keep], g["prune_after_last"])
    assert tuple(g["prune_after_ones_shape"]) == g["prune_after_protos"].shape


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_kld_loss_matches_reference(golden, tag):
    """oracle.kld_loss vs the reference's KLDLoss (segmentation/model/loss.py:51-146): value and gradient."""
    g = golden("kld_loss")
    d = torch.from_numpy(g[f"{tag}_dist"]).requires_grad_(True)
    t = torch.from_numpy(g[f"{tag}_target"])
    ident = torch.from_numpy(g[f"{tag}_ident"])
    S = int(g[f"{tag}_S"])
    ranges = {s: tuple(int(v) for v in g[f"{tag}_ranges"][s]) for s in range(S)}
    loss = O.kld_loss(d, t, ident, S, ranges)
    loss.backward()
    assert abs(loss.item() - float(g[f"{tag}_loss"])) <= 1e-7
    assert np.abs(d.grad.numpy() - g[f"{tag}_grad"]).max() <= 1e-8
    assert float(g["empty_loss"]) == 0.0
    assert O.kld_loss(torch.rand(1, 8, 3, 3), torch.zeros(1, 3, 3, dtype=torch.long), O.default_class_identity(8, 4, 1), 1, {0: (0, 8)}).item() == 0.0


def test_kld_loss_group_matches_reference(golden):
    """oracle.kld_loss_group vs the reference's KLDLossGroup (segmentation/model/loss.py:461-545): value and gradient;
    one class of the fixture owns no prototypes (its pixels must contribute nothing)."""
    g = golden("kld_loss")
    n = int(g["grp_n"])
    acts = [torch.from_numpy(g[f"grp_act{i}"]).requires_grad_(True) for i in range(n)]
    loss = O.kld_loss_group(acts, torch.from_numpy(g["grp_target"]), torch.from_numpy(g["grp_ident"]), torch.from_numpy(g["grp_gci"]), int(g["grp_G"]))
    loss.backward()
    assert abs(loss.item() - float(g["grp_loss"])) <= 1e-7
    for i, a in enumerate(acts):
        assert np.abs(a.grad.numpy() - g[f"grp_grad{i}"]).max() <= 1e-8


@pytest.mark.parametrize("case", [(1, 3, 129, 257, 1024, 2048), (2, 19, 65, 65, 513, 513), (1, 7, 17, 33, 129, 257), (2, 5, 9, 11, 70, 90),
                                  (1, 4, 33, 65, 256, 512)])
def test_bilinear_restatement_is_the_host_kernel_bit_for_bit(case):
    """The operation-by-operation restatement of torch's CPU upsample_bilinear2d that the eval kernel is held to
    (segmentation/eval_valid_multiscale.py:229-234 calls F.interpolate) equals F.interpolate itself bit for bit - at the
    reference's evaluation shapes (latent 129 x 257 -> 1024 x 2048, 65 x 65 -> 513 x 513) and the other maps large enough
    for the host kernel's main loop instantiation (tiny maps take another one, contracted differently by its compiler:
    there the GPU test falls back to a tolerance)."""
    from oracle import ppnet_oracle as O

    N, C, h, w, H, W = case
    src = torch.rand(N, C, h, w, generator=torch.Generator().manual_seed(5)) * 10
    up = torch.nn.functional.interpolate(src, size=(H, W), mode="bilinear", align_corners=False)
    assert torch.equal(up, O.upsample_bilinear_restated(src, (H, W)))
