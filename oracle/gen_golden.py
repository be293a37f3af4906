"""Generate golden input/output vectors from the REAL reference classes.

Run in the build container only (``python oracle/gen_golden.py``): it imports
``/root/reference`` with stub modules for the third-party packages the image
lacks (gin, dotenv, pytorch_lightning, torchvision, cv2, the absent
``deeplab_pytorch`` submodule) and for ``settings`` (which opens a log file at
import).  Output: small ``.npz`` fixtures under ``tests/golden/`` — data only;
nothing of the reference itself is written anywhere.  The reference never
travels to the GPU box; tests there read the committed fixtures.
"""
from __future__ import annotations

import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("SPX_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
SEED = 20220227  # segmentation/configs/scaleproto_cityscapes.gin:15


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def configurable(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    mod("gin", configurable=configurable, REQUIRED=object(), external_configurable=lambda f, *a, **k: f)
    mod("dotenv", load_dotenv=lambda *a, **k: None)
    mod("pytorch_lightning", LightningModule=nn.Module)
    mod("settings", log=lambda *a, **k: None, data_path={})

    class _Dummy(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    mod("deeplab_pytorch")
    mod("deeplab_pytorch.libs")
    mod("deeplab_pytorch.libs.models")
    mod("deeplab_pytorch.libs.models.deeplabv2", DeepLabV2=_Dummy, DeepLabV2_VGG=_Dummy)
    mod("deeplab_pytorch.libs.models.deeplabv2_multiscale", DeepLabV2=_Dummy, DeepLabV2_VGG=_Dummy)
    mod("deeplab_pytorch.libs.models.deeplabv2_multiscaleplus", DeepLabV2=_Dummy)
    mod("deeplab_pytorch.libs.models.deeplabv3_multiscale", DeepLabV3=_Dummy)
    mod("deeplab_pytorch.libs.models.unet", UNet=_Dummy, UNetASPP=_Dummy, UNetASPPBN=_Dummy)

    # push module extras
    mod("cv2")
    tv = mod("torchvision")
    tr = mod("torchvision.transforms", ToTensor=lambda: (lambda x: x), Compose=lambda x: x, Normalize=lambda *a, **k: None)
    ds = mod("torchvision.datasets", VisionDataset=object)
    tv.transforms = tr
    tv.datasets = ds
    mod("matplotlib")
    mod("matplotlib.pyplot")
    mod("tqdm", tqdm=lambda it, **k: it)
    mod("find_nearest", to_normalized_tensor=lambda img: img)
    mod("helpers", find_continuous_high_activation_crop=None, makedir=None)


class _Backbone(nn.Module):
    """Stand-in features module: str() must start with 'MSC' (model_multiscale.py:166-169)."""

    def __init__(self, channels):
        super().__init__()
        self.base = nn.Sequential(nn.Conv2d(3, channels, 1), nn.Conv2d(channels, channels, 1))

    def __repr__(self):
        return "MSC(standin)"

    def forward(self, x):
        return x


def _bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _np(t):
    return t.detach().cpu().numpy()


def _make_proto_phase(ms, *, P, Cs, S, K):
    net = ms.PPNetMultiScale(
        features=_Backbone(Cs * S),
        img_size=64,
        prototype_shape=(P, Cs, 1, 1),
        proto_layer_rf_info=[],
        num_classes=K,
        init_weights=True,
        add_on_layers_type="deeplab_simple",
        patch_classification=True,
        num_scales=S,
    )
    with torch.no_grad():
        net.prototype_vectors.copy_(_bf16r(net.prototype_vectors))
        # perturb the +1/-0.5 init so dW is exercised on a generic matrix
        net.last_layer.weight.add_(0.05 * torch.randn_like(net.last_layer.weight))
    return net


def case_proto_phase(ms, name, *, B, S, Cs, K, P, H, W):
    torch.manual_seed(SEED)
    net = _make_proto_phase(ms, P=P, Cs=Cs, S=S, K=K)
    conv = _bf16r(torch.sigmoid(torch.randn(B, S * Cs, H, W)))
    g_logits = torch.randn(B, H, W, K) * 1e-3
    g_dist = torch.randn(B, P, H, W) * 1e-3
    g_act = torch.randn(B * H * W, P) * 1e-3

    # three tuple modes (model_multiscale.py:378-385)
    with torch.no_grad():
        o_default = net.forward_from_conv_features(conv)
        o_act = net.forward_from_conv_features(conv, return_activations=True)
        o_both = net.forward_from_conv_features(conv, return_activations=True, return_distances=True)
    assert len(o_default) == 2 and len(o_act) == 2 and len(o_both) == 3

    x = conv.clone().requires_grad_(True)
    logits, dist, act = net.forward_from_conv_features(x, return_activations=True, return_distances=True)
    loss = (logits * g_logits).sum() + (dist * g_dist).sum() + (act * g_act).sum()
    net.zero_grad()
    loss.backward()

    sd = net.state_dict()
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        conv=_np(conv),
        prototype_vectors=_np(net.prototype_vectors),
        last_layer_weight=_np(net.last_layer.weight),
        class_identity=_np(net.prototype_class_identity),
        scale_ranges=np.array([net.scale_num_prototypes[s] for s in range(S)], dtype=np.int64),
        num_scales=np.int64(S),
        logits=_np(o_both[0]),
        distances=_np(o_both[1]),
        activations=_np(o_both[2]),
        default_1=_np(o_default[1]),
        act_1=_np(o_act[1]),
        g_logits=_np(g_logits),
        g_dist=_np(g_dist),
        g_act=_np(g_act),
        d_conv=_np(x.grad),
        d_prototypes=_np(net.prototype_vectors.grad),
        d_last_layer=_np(net.last_layer.weight.grad),
        state_keys=np.array([k for k in sd.keys() if not k.startswith("features.")]),
        state_shapes=np.array([str(tuple(v.shape)) for k, v in sd.items() if not k.startswith("features.")]),
    )
    return net


def case_group_phase(msg, ms, name, *, B, S, Cs, K, P, H, W, G):
    torch.manual_seed(SEED + 1)
    old = _make_proto_phase(ms, P=P, Cs=Cs, S=S, K=K)
    net = msg.PPNetMultiScale(
        features=_Backbone(Cs * S),
        img_size=64,
        prototype_shape=(P, Cs, 1, 1),
        proto_layer_rf_info=[],
        num_classes=K,
        init_weights=True,
        add_on_layers_type="deeplab_simple",
        patch_classification=True,
        num_scales=S,
        num_groups=G,
    )
    # phase-1 -> phase-2 hand-off, finetune_wandb_group.py:74-83
    missing = net.load_state_dict(old.state_dict(), strict=False)
    with torch.no_grad():
        net.last_layer_group.weight.add_(0.05 * torch.randn_like(net.last_layer_group.weight))
    conv = _bf16r(torch.sigmoid(torch.randn(B, S * Cs, H, W)))
    g_logits = torch.randn(B, H, W, K) * 1e-3
    g_dist = torch.randn(B, P, H, W) * 1e-3
    g_act = torch.randn(B * H * W, P) * 1e-3

    x = conv.clone().requires_grad_(True)
    logits, dist, act = net.forward_from_conv_features(x, return_activations=True, return_distances=True)
    groups = net.compute_group(act)
    loss = (logits * g_logits).sum() + (dist * g_dist).sum() + (act * g_act).sum()
    net.zero_grad()
    loss.backward()
    sd = net.state_dict()
    out = dict(
        conv=_np(conv),
        prototype_vectors=_np(net.prototype_vectors),
        class_identity=_np(net.prototype_class_identity),
        group_class_identity=_np(net.group_class_identity),
        scale_ranges=np.array([net.scale_num_prototypes[s] for s in range(S)], dtype=np.int64),
        num_scales=np.int64(S),
        num_groups=np.int64(G),
        last_layer_group_weight=_np(net.last_layer_group.weight),
        logits=_np(logits),
        distances=_np(dist),
        activations=_np(act),
        group_cat=_np(torch.cat(groups, dim=-1)),
        g_logits=_np(g_logits),
        g_dist=_np(g_dist),
        g_act=_np(g_act),
        d_conv=_np(x.grad),
        d_prototypes=_np(net.prototype_vectors.grad),
        d_last_layer_group=_np(net.last_layer_group.weight.grad),
        state_keys=np.array([k for k in sd.keys() if not k.startswith("features.")]),
        state_shapes=np.array([str(tuple(v.shape)) for k, v in sd.items() if not k.startswith("features.")]),
        missing_keys=np.array(list(missing.missing_keys)),
        unexpected_keys=np.array(list(missing.unexpected_keys)),
    )
    for i, gp in enumerate(net.group_projection):
        out[f"group_w_{i}"] = _np(gp.weight)
        out[f"d_group_w_{i}"] = _np(gp.weight.grad)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def case_group_train_step(msg, ms, lossmod, name, *, B, S, Cs, K, P, H, W, G):
    """The group-phase training step's call pattern (module_multiscale_group_train.py:222-262): forward with activations and
    distances -> compute_group(activations) -> cross entropy on the logits + KLDLossGroup on the group activations
    (+ a linear term on the distances, standing in for the other distance-side losses)."""
    torch.manual_seed(SEED + 5)
    old = _make_proto_phase(ms, P=P, Cs=Cs, S=S, K=K)
    net = msg.PPNetMultiScale(
        features=_Backbone(Cs * S), img_size=64, prototype_shape=(P, Cs, 1, 1), proto_layer_rf_info=[], num_classes=K,
        init_weights=True, add_on_layers_type="deeplab_simple", patch_classification=True, num_scales=S, num_groups=G,
    )
    net.load_state_dict(old.state_dict(), strict=False)
    with torch.no_grad():
        net.last_layer_group.weight.add_(0.05 * torch.randn_like(net.last_layer_group.weight))
    conv = _bf16r(torch.sigmoid(torch.randn(B, S * Cs, H, W)))
    target = torch.randint(0, K + 1, (B, H, W))
    g_dist = torch.randn(B, P, H, W) * 1e-3
    w_kld = 0.25

    x = conv.clone().requires_grad_(True)
    logits, dist, act = net.forward_from_conv_features(x, return_activations=True, return_distances=True)
    groups = net.compute_group(act)
    ce = lossmod.PixelWiseCrossEntropyLoss(ignore_index=-1)(predicted_logits=logits, target_labels=target)
    kld = lossmod.KLDLossGroup(net.prototype_class_identity, net.group_class_identity, G)(
        list_group_activation=groups, target_labels=target)
    loss = ce + w_kld * kld + (dist * g_dist).sum()
    net.zero_grad()
    loss.backward()
    out = dict(
        conv=_np(conv), prototype_vectors=_np(net.prototype_vectors), class_identity=_np(net.prototype_class_identity),
        group_class_identity=_np(net.group_class_identity),
        scale_ranges=np.array([net.scale_num_prototypes[s] for s in range(S)], dtype=np.int64),
        num_scales=np.int64(S), num_groups=np.int64(G), last_layer_group_weight=_np(net.last_layer_group.weight),
        target=target.numpy().astype(np.int64), g_dist=_np(g_dist), w_kld=np.float32(w_kld),
        logits=_np(logits), group_cat=_np(torch.cat(groups, dim=-1)), ce=_np(ce), kld=_np(kld), loss=_np(loss),
        d_conv=_np(x.grad), d_prototypes=_np(net.prototype_vectors.grad),
        d_last_layer_group=_np(net.last_layer_group.weight.grad),
    )
    for i, gp in enumerate(net.group_projection):
        out[f"group_w_{i}"] = _np(gp.weight)
        out[f"d_group_w_{i}"] = _np(gp.weight.grad)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def case_single_scale(m1, name, *, B, Cs, K, P, H, W):
    torch.manual_seed(SEED + 2)
    net = m1.PPNet(
        features=_Backbone(Cs),
        img_size=64,
        prototype_shape=(P, Cs, 1, 1),
        proto_layer_rf_info=[],
        num_classes=K,
        init_weights=True,
        add_on_layers_type="deeplab_simple",
        patch_classification=True,
    )
    with torch.no_grad():
        net.prototype_vectors.copy_(_bf16r(net.prototype_vectors))
    conv = _bf16r(torch.sigmoid(torch.randn(B, Cs, H, W)))
    with torch.no_grad():
        logits, dist = net.forward_from_conv_features(conv)
        _, act = net.forward_from_conv_features(conv, return_activations=True)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        conv=_np(conv),
        prototype_vectors=_np(net.prototype_vectors),
        last_layer_weight=_np(net.last_layer.weight),
        class_identity=_np(net.prototype_class_identity),
        logits=_np(logits),
        distances=_np(dist),
        activations=_np(act),
    )


def case_push(push, dsmod, ms, name):
    """compute_distances (push_multiscale_optimization.py:34-91) on prepared distances."""
    torch.manual_seed(SEED + 3)
    K, S, r, H, W = 5, 2, 2, 6, 7
    P = K * S * r
    net = _make_proto_phase(ms, P=P, Cs=8, S=S, K=K)
    dist = torch.rand(1, P, H, W) * 10
    # exact tie inside class 2 for prototype 4 (class 2 of scale 0): two equal minima
    target = torch.randint(0, K + 1, (H, W))
    target[target == 4] = 1  # class index 3 (label 4) absent from the image
    target[0, 0] = 3
    target[2, 3] = 3
    target[4, 5] = 3
    p_tie = 2 * r  # first prototype of class 2 at scale 0
    dist[0, p_tie] = 5.0 + torch.rand(H, W)
    dist[0, p_tie, 2, 3] = 0.125
    dist[0, p_tie, 4, 5] = 0.125
    # a huge distance that is still below the 1e10 mask
    dist[0, 1, :, :] += 3e4

    class _FakeNet:
        prototype_class_identity = net.prototype_class_identity

        def to(self, d):
            return self

        def eval(self):
            return self

        def __call__(self, img, return_activations=False):
            return None, dist

    class _FakeDs:
        convert_targets = None

    class _Img:
        def unsqueeze(self, i):
            return self

        def to(self, d):
            return self

    push.to_normalized_tensor = lambda img: _Img()
    # label passes through resize_label with size == its own size (identity resample)
    idx, val = push.compute_distances(_FakeNet(), _FakeDs(), None, target.numpy(), num_classes=K, void_class=0)

    # resize_label goldens (dataset.py:22-30): PIL NEAREST to (W,H)
    lab = torch.randint(0, K + 1, (37, 53)).numpy()
    sizes = [(7, 5), (13, 9), (53, 37), (26, 18)]
    resized = {f"resized_{w}x{h}": _np(dsmod.resize_label(lab, (w, h))) for (w, h) in sizes}

    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        distances=_np(dist),
        target=target.numpy().astype(np.int64),
        class_identity=_np(net.prototype_class_identity),
        num_classes=np.int64(K),
        indices=_np(idx),
        values=_np(val),
        label_full=lab.astype(np.int64),
        resize_sizes=np.array(sizes, dtype=np.int64),
        **resized,
    )


def case_misc(ms, utils, name):
    torch.manual_seed(SEED + 4)
    v = torch.randn(6, 12)
    v[1] = torch.rand(12) * 0.01
    v[2] = 5 * torch.randn(12)
    w = utils.projection_simplex_sort(v)

    # prune_prototypes (model_multiscale.py:400-432)
    K, S, r = 3, 4, 2
    P = K * S * r
    net = _make_proto_phase(ms, P=P, Cs=8, S=S, K=K)
    drop = [0, 1, 7, 13, 23]
    before_w = net.last_layer.weight.detach().clone()
    before_p = net.prototype_vectors.detach().clone()
    net.prune_prototypes(drop)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        simplex_in=_np(v),
        simplex_out=_np(w),
        prune_P=np.int64(P),
        prune_S=np.int64(S),
        prune_K=np.int64(K),
        prune_drop=np.array(drop, dtype=np.int64),
        prune_before_last=_np(before_w),
        prune_before_protos=_np(before_p),
        prune_after_last=_np(net.last_layer.weight),
        prune_after_protos=_np(net.prototype_vectors),
        prune_after_ones_shape=np.array(net.ones.shape, dtype=np.int64),
        prune_after_identity=_np(net.prototype_class_identity),
        prune_after_ranges=np.array([net.scale_num_prototypes[s] for s in range(S)], dtype=np.int64),
    )


def case_kld(lossmod, ms, name):
    """KLDLoss of the reference (segmentation/model/loss.py:51-146) on small distance maps: value + gradient."""
    out = {}
    cases = {
        # tag: (B, S, K, P, H, W)
        "a": (2, 4, 5, 40, 9, 11),      # 2 prototypes per (class, scale): one pair each
        "b": (1, 1, 4, 16, 8, 9),       # 4 prototypes per class: six pairs
        "c": (1, 2, 3, 16, 4, 5),       # floor semantics: 2 prototypes per scale without a class
    }
    for tag, (B, S, K, P, H, W) in cases.items():
        torch.manual_seed(SEED + 40 + ord(tag))
        net = _make_proto_phase(ms, P=P, Cs=16, S=S, K=K)
        ident = net.prototype_class_identity.clone()
        ranges = {s: tuple(net.scale_num_prototypes[s]) for s in range(S)}
        d = (torch.rand(B, P, H, W) * 6.0).requires_grad_(True)
        target = torch.randint(0, K + 1, (B, H, W))          # 0 = void
        target[target == K] = 0                              # class K-1 absent
        target[0, 0, 0] = K                                  # ... except ONE pixel (len < 2: its pairs are skipped)
        loss = lossmod.KLDLoss(ident, S, ranges)(prototype_distances=d, target_labels=target)
        loss.backward()
        out[f"{tag}_dist"] = _np(d)
        out[f"{tag}_target"] = target.numpy().astype(np.int64)
        out[f"{tag}_ident"] = _np(ident)
        out[f"{tag}_ranges"] = np.array([ranges[s] for s in range(S)], dtype=np.int64)
        out[f"{tag}_S"] = np.int64(S)
        out[f"{tag}_loss"] = _np(loss)
        out[f"{tag}_grad"] = _np(d.grad)
    # KLDLossGroup (loss.py:461-545) on group activations of the group-phase module
    torch.manual_seed(SEED + 70)
    Kg, Sg, Gg, Pg = 5, 4, 3, 40
    netp = _make_proto_phase(ms, P=Pg, Cs=16, S=Sg, K=Kg)
    ident_g = netp.prototype_class_identity.clone()
    ident_g[ident_g[:, 4] > 0] = 0                      # class 4 owns no prototype: its projection does not exist
    present = [k for k in range(Kg) if int(ident_g[:, k].sum()) > 0]
    gci = torch.zeros(len(present) * Gg, Kg)
    for j, k in enumerate(present):
        gci[j * Gg:(j + 1) * Gg, k] = 1
    Bg, Hg, Wg = 2, 7, 9
    acts = [torch.exp(torch.randn(Bg * Hg * Wg, Gg)).requires_grad_(True) for _ in present]
    tgt = torch.randint(0, Kg + 1, (Bg, Hg, Wg))
    lg_loss = lossmod.KLDLossGroup(ident_g, gci, Gg)(list_group_activation=acts, target_labels=tgt)
    lg_loss.backward()
    out["grp_ident"] = _np(ident_g)
    out["grp_gci"] = _np(gci)
    out["grp_G"] = np.int64(Gg)
    out["grp_target"] = tgt.numpy().astype(np.int64)
    for i, t_ in enumerate(acts):
        out[f"grp_act{i}"] = _np(t_)
        out[f"grp_grad{i}"] = _np(t_.grad)
    out["grp_n"] = np.int64(len(acts))
    out["grp_loss"] = _np(lg_loss)
    # PixelWiseCrossEntropyLoss (loss.py:9-48): the training modules construct it with ignore_index=-1 so that void
    # (label 0 -> -1 after the shift) is skipped
    torch.manual_seed(SEED + 60)
    lg = torch.randn(2, 6, 7, 5, requires_grad=True)
    tg = torch.randint(0, 6, (2, 6, 7))
    ce, correct = lossmod.PixelWiseCrossEntropyLoss(ignore_index=-1, return_correct=True)(predicted_logits=lg, target_labels=tg)
    ce.backward()
    out["ce_logits"] = _np(lg)
    out["ce_target"] = tg.numpy().astype(np.int64)
    out["ce_loss"] = _np(ce)
    out["ce_grad"] = _np(lg.grad)
    out["ce_correct"] = correct.numpy().astype(np.int64)
    # no valid term at all -> 0.0 (loss.py:143-144)
    net = _make_proto_phase(ms, P=8, Cs=16, S=1, K=4)
    t0 = torch.zeros(1, 3, 3, dtype=torch.long)
    l0 = lossmod.KLDLoss(net.prototype_class_identity, 1, {0: (0, 8)})(prototype_distances=torch.rand(1, 8, 3, 3), target_labels=t0)
    out["empty_loss"] = _np(l0)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    os.makedirs(OUT, exist_ok=True)
    import segmentation.model.model_multiscale as ms
    import segmentation.model.model_multiscale_group as msg
    import segmentation.model.model as m1
    import segmentation.utils as utils
    import segmentation.data.dataset as dsmod
    import segmentation.push_multiscale_optimization as push
    import segmentation.model.loss as lossmod

    torch.set_num_threads(1)
    case_proto_phase(ms, "proto_ms_small", B=2, S=4, Cs=16, K=5, P=40, H=9, W=11)
    case_proto_phase(ms, "proto_ms_city", B=1, S=4, Cs=64, K=19, P=228, H=17, W=17)
    case_proto_phase(ms, "proto_s3", B=1, S=3, Cs=16, K=19, P=171, H=5, W=6)
    # P % (K*S) != 0: floor semantics leave 2 prototypes per scale without a class (model_multiscale.py:132-141)
    case_proto_phase(ms, "proto_floor", B=1, S=2, Cs=16, K=3, P=16, H=4, W=5)
    case_proto_phase(ms, "proto_s1_wide", B=1, S=1, Cs=256, K=19, P=190, H=6, W=7)
    case_group_phase(msg, ms, "group_ms_small", B=2, S=4, Cs=16, K=5, P=40, H=9, W=11, G=3)
    case_group_train_step(msg, ms, lossmod, "group_train_step", B=2, S=4, Cs=16, K=5, P=40, H=9, W=11, G=3)
    case_single_scale(m1, "ppnet_single", B=2, Cs=32, K=5, P=20, H=7, W=9)
    case_push(push, dsmod, ms, "push_argmin")
    case_misc(ms, utils, "misc")
    case_kld(lossmod, ms, "kld_loss")
    print("golden fixtures written to", os.path.normpath(OUT))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
