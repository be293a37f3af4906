"""CPU checks of the drop-in modules' host logic against the reference-generated fixtures: constructor
surface, state_dict keys / shapes, class tables, pruning, phase-1 -> phase-2 hand-off, label resize."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import scaleprotoseg_amd as spx


class _Backbone(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.base = nn.Sequential(nn.Conv2d(3, ch, 1), nn.Conv2d(ch, ch, 1))

    def __repr__(self):
        return "MSC(standin)"

    def forward(self, x):
        return x


def _proto(P, Cs, S, K, **kw):
    return spx.PPNetMultiScale(_Backbone(Cs * S), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                               patch_classification=True, num_scales=S, **kw)


@pytest.mark.parametrize("name", ["proto_ms_small", "proto_ms_city", "proto_s3", "proto_floor", "proto_s1_wide"])
def test_state_dict_and_tables_match_reference(golden, name):
    g = golden(name)
    S = int(g["num_scales"])
    P, K = g["class_identity"].shape
    Cs = g["prototype_vectors"].shape[1]
    net = _proto(P, Cs, S, K)
    sd = {k: v for k, v in net.state_dict().items() if not k.startswith("features.")}
    assert list(sd.keys()) == list(g["state_keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["state_shapes"])
    np.testing.assert_array_equal(net.prototype_class_identity.numpy(), g["class_identity"])
    np.testing.assert_array_equal(np.array([net.scale_num_prototypes[s] for s in range(S)]), g["scale_ranges"])
    assert net.num_prototypes == P and net.num_classes == K and tuple(net.prototype_shape) == (P, Cs, 1, 1)
    assert not net.ones.requires_grad and net.prototype_vectors.requires_grad
    # +1 / -0.5 last layer (model_multiscale.py:449-464)
    w = net.last_layer.weight.detach()
    ident = net.prototype_class_identity.t()
    assert torch.equal(w, ident - 0.5 * (1 - ident))
    # loading the reference's parameters works key for key
    net.load_state_dict({"prototype_vectors": torch.from_numpy(g["prototype_vectors"]),
                         "ones": torch.ones(P, Cs, 1, 1),
                         "last_layer.weight": torch.from_numpy(g["last_layer_weight"])}, strict=False)


def test_prune_matches_reference(golden):
    g = golden("misc")
    P, S, K = int(g["prune_P"]), int(g["prune_S"]), int(g["prune_K"])
    net = spx.PPNetMultiScale(_Backbone(8 * S), 64, (P, 8, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                              patch_classification=True, num_scales=S)
    with torch.no_grad():
        net.prototype_vectors.copy_(torch.from_numpy(g["prune_before_protos"]))
        net.last_layer.weight.copy_(torch.from_numpy(g["prune_before_last"]))
    net.prune_prototypes([int(i) for i in g["prune_drop"]])
    np.testing.assert_array_equal(net.prototype_vectors.detach().numpy(), g["prune_after_protos"])
    np.testing.assert_array_equal(net.last_layer.weight.detach().numpy(), g["prune_after_last"])
    np.testing.assert_array_equal(net.prototype_class_identity.numpy(), g["prune_after_identity"])
    np.testing.assert_array_equal(np.array([net.scale_num_prototypes[s] for s in range(S)]), g["prune_after_ranges"])
    assert tuple(net.ones.shape) == tuple(g["prune_after_ones_shape"])
    assert net.last_layer.in_features == net.num_prototypes
    # the kernel plan follows the re-packed scale table
    lay = net._layout(K)
    assert lay.scale_ranges == tuple(tuple(int(v) for v in r) for r in g["prune_after_ranges"])
    from scaleprotoseg_amd import _lib

    plan = _lib.make_plan(net.num_prototypes, K, S, 16, [r[0] for r in lay.scale_ranges], [r[1] for r in lay.scale_ranges])
    assert [plan.panel_np[q] for q in range(plan.npanels)] == [hi - lo for lo, hi in lay.scale_ranges]


def test_group_phase_surface(golden):
    g = golden("group_ms_small")
    S, G = int(g["num_scales"]), int(g["num_groups"])
    P, K = g["class_identity"].shape
    Cs = g["prototype_vectors"].shape[1]
    old = _proto(P, Cs, S, K)
    net = spx.PPNetMultiScaleGroup(_Backbone(Cs * S), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                                   patch_classification=True, num_scales=S, num_groups=G)
    sd = {k: v for k, v in net.state_dict().items() if not k.startswith("features.")}
    assert list(sd.keys()) == list(g["state_keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["state_shapes"])
    # phase-1 -> phase-2 hand-off (finetune_wandb_group.py:76): strict=False, last_layer is the only stranger
    res = net.load_state_dict(old.state_dict(), strict=False)
    assert list(res.unexpected_keys) == list(g["unexpected_keys"])
    assert sorted(res.missing_keys) == sorted(g["missing_keys"])
    np.testing.assert_array_equal(net.group_class_identity.numpy(), g["group_class_identity"])
    for gp in net.group_projection:   # simplex rows after init (model_multiscale_group.py:516-517)
        w = gp.weight.detach()
        assert (w >= 0).all() and torch.allclose(w.sum(1), torch.ones(G), atol=1e-5)
    gi = net.group_class_identity.t()
    assert torch.equal(net.last_layer_group.weight.detach(), gi + net.incorrect_strength * (1 - gi))
    # dense form of the grouping head == the reference's per-class gather + linear (compute_group, restated by the oracle)
    from oracle import ppnet_oracle as O

    act = torch.rand(7, P)
    dense = torch.exp(act @ net._dense_group_matrix().t())
    ref = torch.cat(O.compute_group(act, net.prototype_class_identity, [gp.weight.detach() for gp in net.group_projection]), dim=-1)
    np.testing.assert_allclose(dense.detach().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)
    with pytest.raises(Exception):        # the method itself runs on the HIP kernels only: no CPU fallback
        net.compute_group(act)


def test_single_scale_ppnet(golden):
    g = golden("ppnet_single")
    P, K = g["class_identity"].shape
    net = spx.PPNet(_Backbone(32), 64, (P, 32, 1, 1), [], K, add_on_layers_type="deeplab_simple", patch_classification=True)
    np.testing.assert_array_equal(net.prototype_class_identity.numpy(), g["class_identity"])
    assert net.num_prototypes_per_class == P // K and net.num_scales == 1


def test_reference_error_behaviour():
    net = _proto(40, 16, 4, 5)
    net.patch_classification = False
    with pytest.raises(Exception, match="Original Prototype Network Implementation"):
        net.forward_from_conv_features(torch.zeros(1, 64, 4, 4))     # model_multiscale.py:387-388
    with pytest.raises(Exception, match="base_architecture NOT implemented"):
        spx.PPNetMultiScale(nn.Linear(2, 2), 64, (8, 16, 1, 1), [], 2)   # :171
    net = _proto(40, 16, 4, 5)
    with pytest.raises(spx.SpxError, match="no CPU fallback"):
        net.forward_from_conv_features(torch.zeros(1, 64, 4, 4))


def test_resize_label_and_simplex(golden):
    g = golden("push_argmin")
    for w, h in g["resize_sizes"]:
        np.testing.assert_array_equal(spx.resize_label(g["label_full"], (int(w), int(h))).numpy(), g[f"resized_{w}x{h}"])
    m = golden("misc")
    np.testing.assert_allclose(spx.projection_simplex_sort(torch.from_numpy(m["simplex_in"])).numpy(), m["simplex_out"], atol=1e-6)


def test_resize_label_matches_pil_sampling():
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    for _ in range(60):
        hi, wi, ho, wo = (int(v) for v in rng.integers(1, 300, 4))
        lab = rng.integers(0, 20, (hi, wi))
        ref = np.asarray(PIL.fromarray(lab.astype(float)).resize((wo, ho), resample=PIL.NEAREST)).astype(np.int64)
        np.testing.assert_array_equal(spx.resize_label(lab, (wo, ho)).numpy(), ref)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_kld_loss_module_matches_reference(golden, tag):
    """scaleprotoseg_amd.loss.KLDLoss's host logic (slot table, pair mask, class gather) under the oracle's torch restatement of
    the algebra (oracle/loss_oracle.py) against the reference's golden value and gradient, on the full map and on the
    class-gathered form.  (The HIP kernels are held against the same oracle in fp64 by tests/test_gpu_parity.py.)"""
    import numpy as np
    from oracle import loss_oracle as LO
    from oracle import ppnet_oracle as O
    from scaleprotoseg_amd.loss import ClassDistances, KLDLoss, class_slot_table, gather_class_distances

    g = golden("kld_loss")
    t = torch.from_numpy(g[f"{tag}_target"])
    ident = torch.from_numpy(g[f"{tag}_ident"])
    S = int(g[f"{tag}_S"])
    ranges = {s: tuple(int(v) for v in g[f"{tag}_ranges"][s]) for s in range(S)}
    d = torch.from_numpy(g[f"{tag}_dist"]).requires_grad_(True)
    mod = KLDLoss(ident, S, ranges)              # the product's host logic (slot table, pair mask); the algebra: the oracle's
    loss = LO.kld_loss(mod, d, t)
    loss.backward()
    assert abs(loss.item() - float(g[f"{tag}_loss"])) <= 1e-6
    scale = np.abs(g[f"{tag}_grad"]).max()
    assert np.abs(d.grad.numpy() - g[f"{tag}_grad"]).max() <= 1e-5 * scale
    table = class_slot_table(ident)
    assert torch.equal(table, O.class_slot_table(ident))
    lab0 = t.reshape(t.shape[0], -1) - 1
    cv = gather_class_distances(torch.from_numpy(g[f"{tag}_dist"]), lab0, table)
    assert torch.equal(cv, O.gather_class_distances(torch.from_numpy(g[f"{tag}_dist"]), lab0, ident))
    lg = LO.kld_loss(mod, ClassDistances(cv.permute(0, 2, 1).contiguous(), lab0, table, tuple(t.shape[1:])), t)
    assert abs(lg.item() - float(g[f"{tag}_loss"])) <= 1e-6


def test_kld_loss_no_terms():
    from scaleprotoseg_amd.loss import KLDLoss
    from oracle import ppnet_oracle as O

    ident = O.default_class_identity(8, 4, 1)
    from oracle import loss_oracle as LO

    assert LO.kld_loss(KLDLoss(ident, 1, {0: (0, 8)}), torch.rand(1, 8, 3, 3), torch.zeros(1, 3, 3, dtype=torch.long)).item() == 0.0


def test_reference_state_dict_after_push_dedup(tmp_path, golden):
    """a state_dict written after the push's de-duplication loads into a fresh module once the kept
    indices (unique_prototypes.json) are given; class table and scale table equal the pruned module's (pinned to the
    reference's prune_prototypes by the golden)."""
    import json
    from scaleprotoseg_amd.checkpoint import export_state, import_state, load_reference_state_dict

    g = golden("misc")
    P, S, K = int(g["prune_P"]), int(g["prune_S"]), int(g["prune_K"])
    drop = [int(i) for i in g["prune_drop"]]
    src = _proto(P, 8, S, K)
    with torch.no_grad():
        src.prototype_vectors.copy_(torch.from_numpy(g["prune_before_protos"]))
        src.last_layer.weight.copy_(torch.from_numpy(g["prune_before_last"]))
    src.prune_prototypes(drop)
    keep = sorted(set(range(P)) - set(drop))
    path = tmp_path / "unique_prototypes.json"
    path.write_text(json.dumps(keep))

    dst = _proto(P, 8, S, K)
    with pytest.raises(ValueError, match="unique_prototypes"):
        load_reference_state_dict(_proto(P, 8, S, K), src.state_dict())
    res = load_reference_state_dict(dst, src.state_dict(), unique_prototypes=str(path))
    assert not res.missing_keys and not res.unexpected_keys
    np.testing.assert_array_equal(dst.prototype_class_identity.numpy(), g["prune_after_identity"])
    assert [tuple(dst.scale_num_prototypes[s]) for s in range(S)] == [tuple(r) for r in g["prune_after_ranges"]]
    np.testing.assert_array_equal(dst.prototype_vectors.detach().numpy(), g["prune_after_protos"])
    np.testing.assert_array_equal(dst.last_layer.weight.detach().numpy(), g["prune_after_last"])

    dst2 = _proto(P, 8, S, K)
    import_state(dst2, export_state(src))
    np.testing.assert_array_equal(dst2.prototype_class_identity.numpy(), g["prune_after_identity"])
    np.testing.assert_array_equal(dst2.prototype_vectors.detach().numpy(), g["prune_after_protos"])
    assert dst2.scale_num_prototypes == {s: tuple(int(v) for v in g["prune_after_ranges"][s]) for s in range(S)}


def test_pruned_prototype_checkpoint_loads_into_the_group_module(tmp_path, golden):
    """finetune_wandb_group.py:74-80 hand-over after a push: a de-duplicated prototype-phase state_dict +
    unique_prototypes.json loads into the group-phase module; projection shapes and group_class_identity follow the
    pruned class table."""
    import json
    from scaleprotoseg_amd.checkpoint import export_state, import_state, load_reference_state_dict
    from scaleprotoseg_amd.model_multiscale_group import PPNetMultiScale as GroupNet

    g = golden("misc")
    P, S, K = int(g["prune_P"]), int(g["prune_S"]), int(g["prune_K"])
    drop = [int(i) for i in g["prune_drop"]]
    src = _proto(P, 8, S, K)
    with torch.no_grad():
        src.prototype_vectors.copy_(torch.from_numpy(g["prune_before_protos"]))
    src.prune_prototypes(drop)
    keep = sorted(set(range(P)) - set(drop))
    path = tmp_path / "unique_prototypes.json"
    path.write_text(json.dumps(keep))

    def fresh():
        return GroupNet(_Backbone(8 * S), 64, (P, 8, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                        patch_classification=True, num_scales=S, num_groups=3)

    dst = fresh()
    res = load_reference_state_dict(dst, src.state_dict(), unique_prototypes=str(path))
    assert list(res.unexpected_keys) == ["last_layer.weight"]
    np.testing.assert_array_equal(dst.prototype_vectors.detach().numpy(), g["prune_after_protos"])
    np.testing.assert_array_equal(dst.prototype_class_identity.numpy(), g["prune_after_identity"])
    assert [tuple(dst.scale_num_prototypes[s]) for s in range(S)] == [tuple(r) for r in g["prune_after_ranges"]]
    ident = torch.from_numpy(g["prune_after_identity"])
    present = [k for k in range(K) if ident[:, k].sum() > 0]
    assert [tuple(gp.weight.shape) for gp in dst.group_projection] == [(3, int(ident[:, k].sum())) for k in present]
    assert tuple(dst.group_class_identity.shape) == (3 * len(present), K)
    assert tuple(dst.last_layer_group.weight.shape) == (K, 3 * len(present))
    allowed = torch.cat([ident[:, k].bool().unsqueeze(0).expand(3, -1) for k in present])
    assert not ((dst._dense_group_matrix() != 0) & ~allowed).any()     # dense head: weights only on own-class columns
    dst2 = fresh()
    import_state(dst2, export_state(src))
    np.testing.assert_array_equal(dst2.prototype_class_identity.numpy(), g["prune_after_identity"])
    assert [tuple(gp.weight.shape) for gp in dst2.group_projection] == [tuple(gp.weight.shape) for gp in dst.group_projection]


def test_table_caches_follow_in_place_edits_and_reassignment():
    """The (class, slot) gather table and the dense grouping index are keyed on the identity OBJECT, its in-place
    version and the module's table counter: an in-place edit or a re-assignment invalidates them."""
    from scaleprotoseg_amd.model_multiscale_group import PPNetMultiScale as GroupNet

    net = GroupNet(_Backbone(32), 64, (12, 8, 1, 1), [], 3, add_on_layers_type="deeplab_simple",
                   patch_classification=True, num_scales=4, num_groups=2)
    r0, c0, n0, _ = net._group_index("cpu")
    assert net._group_index("cpu")[0] is r0                      # cached
    v = net._tables_version
    net.prototype_class_identity = net.prototype_class_identity.clone()
    assert net._tables_version == v + 1
    assert net._group_index("cpu")[0] is not r0                  # re-assignment: rebuilt
    r1 = net._group_index("cpu")[0]
    net.prototype_class_identity[0, 0] = 0                       # in-place edit: tensor version moves
    net.prototype_class_identity[0, 1] = 1
    r2, c2, _, _ = net._group_index("cpu")
    assert r2 is not r1 and not torch.equal(c2, c0)


def test_cross_entropy_matches_reference(golden):
    from scaleprotoseg_amd.loss import PixelWiseCrossEntropyLoss

    g = golden("kld_loss")
    lg = torch.from_numpy(g["ce_logits"]).requires_grad_(True)
    from oracle import loss_oracle as LO

    ce, correct = LO.pixelwise_cross_entropy(lg, torch.from_numpy(g["ce_target"]), ignore_index=-1, return_correct=True)
    with pytest.raises(spx.SpxError):            # the product class has no CPU backend
        PixelWiseCrossEntropyLoss(ignore_index=-1)(lg, torch.from_numpy(g["ce_target"]))
    ce.backward()
    assert abs(ce.item() - float(g["ce_loss"])) <= 1e-6
    np.testing.assert_allclose(lg.grad.numpy(), g["ce_grad"], atol=1e-7)
    np.testing.assert_array_equal(correct.numpy().astype(np.int64), g["ce_correct"])


def test_kld_loss_group_module_matches_reference(golden):
    """scaleprotoseg_amd.loss.KLDLossGroup (host logic; algebra: oracle/loss_oracle.py) against the reference's golden
    value and gradients; list input (the reference's) and the concatenated [M, U] tensor give the same loss."""
    import numpy as np
    from scaleprotoseg_amd.loss import KLDLossGroup

    g = golden("kld_loss")
    n = int(g["grp_n"])
    acts = [torch.from_numpy(g[f"grp_act{i}"]).requires_grad_(True) for i in range(n)]
    t = torch.from_numpy(g["grp_target"])
    from oracle import loss_oracle as LO

    m = KLDLossGroup(torch.from_numpy(g["grp_ident"]), torch.from_numpy(g["grp_gci"]), int(g["grp_G"]))
    loss = LO.kld_group_loss(m, acts, t)
    loss.backward()
    assert abs(loss.item() - float(g["grp_loss"])) <= 1e-6
    for i, a in enumerate(acts):
        ref = g[f"grp_grad{i}"]
        assert np.abs(a.grad.numpy() - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-12)
    cat = torch.cat([a.detach() for a in acts], dim=1)
    assert abs(LO.kld_group_loss(m, cat, t).item() - loss.item()) <= 1e-7
    assert LO.kld_group_loss(m, acts, torch.zeros_like(t)).item() == 0.0   # void only: no term (loss.py:541-542)


def test_kld_loss_refuses_inputs_outside_the_kernels():
    """Without the explicit opt-in the loss modules do not leave the GPU path: a CPU tensor is an error, not a fallback."""
    from oracle import ppnet_oracle as O
    from scaleprotoseg_amd import SpxError
    from scaleprotoseg_amd.loss import KLDLoss, KLDLossGroup

    ident = O.default_class_identity(8, 4, 1)
    with pytest.raises(SpxError):
        KLDLoss(ident, 1, {0: (0, 8)})(torch.rand(1, 8, 3, 3), torch.ones(1, 3, 3, dtype=torch.long))
    gci = torch.zeros(8, 4)
    for k in range(4):
        gci[2 * k:2 * k + 2, k] = 1
    with pytest.raises(SpxError):
        KLDLossGroup(ident, gci, 2)([torch.rand(9, 2) for _ in range(4)], torch.ones(1, 3, 3, dtype=torch.long))
