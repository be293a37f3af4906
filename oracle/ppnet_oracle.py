"""CPU oracle for the ScaleProtoSeg prototype-distance hot path.

TEST INFRASTRUCTURE ONLY.  This file is a CPU restatement (stock PyTorch CPU
ops, fp32) of the reference's algorithm for the hot path named in
BASELINE.json.  It is imported only by ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg, as the *checker*.  The product path
(``scaleprotoseg_amd``) never imports it and has no CPU fallback.

Parity pin: the functions below are checked against golden vectors generated
in the build container by importing the reference's real classes
(``oracle/gen_golden.py`` -> ``tests/golden/*.npz``; see tests/test_oracle_golden.py).

Every function cites the reference file:line it restates (paths relative to
the reference checkout).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

EPSILON = 1e-4  # segmentation/model/model_multiscale.py:106


# --------------------------------------------------------------------------- #
# distances
# --------------------------------------------------------------------------- #
def l2_convolution(x: torch.Tensor, prototypes: torch.Tensor) -> torch.Tensor:
    """relu(|x|^2 - 2 x.p + |p|^2) per pixel x prototype.

    Restates segmentation/model/model_multiscale.py:255-281 (same op order:
    an all-ones 1x1 conv for |x|^2, a 1x1 conv for x.p, ``-2*xp + p2`` then
    ``+ x2`` then relu), so fp32 results agree with the reference to rounding.

    x: [B, Cs, H, W] fp32; prototypes: [n, Cs, 1, 1] fp32 -> [B, n, H, W].
    """
    ones = torch.ones_like(prototypes)
    x_sq_sum = F.conv2d(x * x, ones)
    p_sq = (prototypes * prototypes).sum(dim=(1, 2, 3)).view(-1, 1, 1)
    cross = F.conv2d(x, prototypes)
    return F.relu(x_sq_sum + (-2 * cross + p_sq))


def default_scale_ranges(num_prototypes: int, num_scales: int) -> Dict[int, Tuple[int, int]]:
    """scale -> (lo, hi) prototype rows; model_multiscale.py:132,146-149."""
    per = num_prototypes // num_scales
    return {s: (s * per, (s + 1) * per) for s in range(num_scales)}


def default_class_identity(num_prototypes: int, num_classes: int, num_scales: int) -> torch.Tensor:
    """[P, K] one-hot, scale-major / class-minor blocks; model_multiscale.py:129-141.

    Floor semantics are kept: prototypes beyond ``num_scales * per_scale`` or
    beyond ``num_classes * per_class_scale`` inside a scale get no class.
    """
    ident = torch.zeros(num_prototypes, num_classes)
    per_scale = num_prototypes // num_scales
    per_cs = num_prototypes // num_classes // num_scales
    for s in range(num_scales):
        for k in range(num_classes):
            ident[s * per_scale + k * per_cs : s * per_scale + (k + 1) * per_cs, k] = 1
    return ident


def scale_l2_convolution(
    x: torch.Tensor,
    prototype_vectors: torch.Tensor,
    scale_ranges: Dict[int, Tuple[int, int]],
    num_scales: int,
) -> torch.Tensor:
    """Per-scale distances concatenated in scale order 0..S-1.

    Restates model_multiscale.py:283-317 for ``scale_head is None`` (every
    shipped config): channel block ``s`` meets prototype rows ``scale_ranges[s]``.
    x: [B, S*Cs, H, W] -> [B, P, H, W].
    """
    B, C, H, W = x.shape
    cs = C // num_scales
    xs = x.view(B, num_scales, cs, H, W)
    protos = prototype_vectors.view(prototype_vectors.shape[0], cs, 1, 1)
    outs = []
    for s in range(num_scales):
        lo, hi = scale_ranges[s]
        outs.append(l2_convolution(xs[:, s], protos[lo:hi]))
    return torch.cat(outs, dim=1)


def distance_2_similarity(d: torch.Tensor, epsilon: float = EPSILON, fn: str = "log") -> torch.Tensor:
    """model_multiscale.py:324-330."""
    if fn == "log":
        return torch.log((d + 1) / (d + epsilon))
    if fn == "linear":
        return -d
    raise ValueError(fn)


# --------------------------------------------------------------------------- #
# heads
# --------------------------------------------------------------------------- #
def last_layer_init(class_identity: torch.Tensor, incorrect_strength: float = -0.5) -> torch.Tensor:
    """[K, P] = +1 on own class, ``incorrect_strength`` elsewhere; model_multiscale.py:449-464."""
    pos = class_identity.t()
    return 1.0 * pos + incorrect_strength * (1 - pos)


def class_prototype_index(class_identity: torch.Tensor) -> List[torch.Tensor]:
    """Index sets idx_k for classes owning >=1 prototype; model_multiscale_group.py:293-298."""
    out = []
    for k in range(class_identity.shape[1]):
        if int(class_identity[:, k].sum().item()) > 0:
            out.append(torch.nonzero(class_identity[:, k]).flatten())
    return out


def compute_group(
    activations: torch.Tensor, class_identity: torch.Tensor, group_weights: Sequence[torch.Tensor]
) -> List[torch.Tensor]:
    """g_k = exp(act[:, idx_k] @ W_k^T); model_multiscale_group.py:283-303."""
    outs = []
    for idx, w in zip(class_prototype_index(class_identity), group_weights):
        outs.append(torch.exp(F.linear(activations[:, idx], w)))
    return outs


def group_class_identity(class_identity: torch.Tensor, num_groups: int) -> torch.Tensor:
    """[G*K', K]; model_multiscale_group.py:262-267."""
    present = [k for k in range(class_identity.shape[1]) if int(class_identity[:, k].sum().item()) > 0]
    gi = torch.zeros(len(present) * num_groups, class_identity.shape[1])
    for j, k in enumerate(present):
        gi[j * num_groups : (j + 1) * num_groups, k] = 1
    return gi


def projection_simplex_sort(v: torch.Tensor, z: float = 1.0) -> torch.Tensor:
    """Row-wise Euclidean projection onto the simplex; segmentation/utils.py:113-124."""
    n = v.size(1)
    u, _ = torch.sort(v, descending=True)
    css = torch.cumsum(u, 1) - z
    k = torch.arange(n).type_as(v) + 1
    cond = (u - css / k) > 0
    rho, rho_idx = (k * cond).max(1)
    theta = torch.gather(css, 1, rho_idx[:, None]) / rho[:, None]
    return torch.clamp(v - theta, min=0)


# --------------------------------------------------------------------------- #
# forward
# --------------------------------------------------------------------------- #
def forward_from_conv_features(
    conv: torch.Tensor,
    prototype_vectors: torch.Tensor,
    scale_ranges: Dict[int, Tuple[int, int]],
    num_scales: int,
    last_layer_weight: Optional[torch.Tensor] = None,
    *,
    class_identity: Optional[torch.Tensor] = None,
    group_weights: Optional[Sequence[torch.Tensor]] = None,
    last_layer_group_weight: Optional[torch.Tensor] = None,
    epsilon: float = EPSILON,
    activation: str = "log",
):
    """(logits [B,H,W,K], distances [B,P,H,W], activations [M,P]).

    Restates model_multiscale.py:340-385 (linear head, ``last_layer_weight``
    given) and model_multiscale_group.py:404-449 + :305-308 (grouping head).
    """
    d = scale_l2_convolution(conv, prototype_vectors, scale_ranges, num_scales)
    B, P, H, W = d.shape
    act = distance_2_similarity(d.permute(0, 2, 3, 1).contiguous().reshape(-1, P), epsilon, activation)
    if last_layer_weight is not None:
        logits = F.linear(act, last_layer_weight)
    else:
        groups = compute_group(act, class_identity, group_weights)
        logits = F.linear(torch.cat(groups, dim=-1), last_layer_group_weight)
    return logits.reshape(B, H, W, -1), d, act


# --------------------------------------------------------------------------- #
# prototype push
# --------------------------------------------------------------------------- #
def resize_label(label: np.ndarray, size: Tuple[int, int]) -> torch.Tensor:
    """PIL NEAREST resize of a float image to (W, H); segmentation/data/dataset.py:22-30."""
    from PIL import Image

    img = Image.fromarray(label.astype(float)).resize(size, resample=Image.NEAREST)
    return torch.LongTensor(np.array(img))


def push_masked_argmin(
    distances: torch.Tensor,
    target: torch.Tensor,
    class_identity: torch.Tensor,
    num_classes: int,
    max_dist: float = 1e10,
    void_class: Optional[int] = 0,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Class-masked per-prototype min over H*W -> (indices int64 [B,P], values f32 [B,P]).

    Restates segmentation/push_multiscale_optimization.py:68-91 on prepared
    ``distances [B,P,H,W]`` and an already resized ``target [B,H,W]`` (labels
    0..K with ``void_class`` dropped from the one-hot).  fp32 ``d + 1e10``
    absorbs d, so masked pixels tie at 1e10 and ``torch.min`` returns the
    first flat index; a prototype whose class is absent yields (0, 1e10).
    """
    one_hot = F.one_hot(target, num_classes=num_classes if void_class is None else num_classes + 1).to(torch.float32)
    if void_class is not None:
        one_hot = torch.cat([one_hot[..., :void_class], one_hot[..., void_class + 1 :]], dim=-1)
    mask = torch.matmul(one_hot, class_identity.t().clone())  # [B,H,W,P]
    penal = max_dist * (1 - mask.permute(0, 3, 1, 2))
    masked = (distances + penal).flatten(-2, -1)
    res = masked.min(dim=-1)
    return res.indices, res.values


def min_across_images(values: Sequence[torch.Tensor]) -> torch.Tensor:
    """argmin over images per prototype (ties -> lowest image); push_multiscale_optimization.py:135-137."""
    return torch.cat(list(values), dim=0).argmin(dim=0)


def gather_push_patches(
    conv_per_image: Sequence[torch.Tensor],
    best_img: torch.Tensor,
    idx_per_image: Sequence[torch.Tensor],
    num_scales: int,
    num_prototypes: int,
) -> np.ndarray:
    """New prototype bank [P, Cs, 1, 1] from the winning latent pixels.

    Restates push_multiscale_optimization.py:162-188: scale of prototype p is
    ``p // (P // S)``; flat index -> (i, j) = (idx // W, idx % W).
    """
    per_scale = num_prototypes // num_scales
    out = []
    for p in range(num_prototypes):
        s = p // per_scale
        conv = conv_per_image[int(best_img[p])]
        _, C, H, W = conv.shape
        cv = conv.view(num_scales, C // num_scales, H, W)
        flat = int(idx_per_image[int(best_img[p])][:, p].item())
        i, j = flat // W, flat % W
        out.append(cv[s, :, i : i + 1, j : j + 1].detach().cpu().numpy())
    return np.reshape(out, (num_prototypes, -1, 1, 1))


def duplicate_prototypes(bank: np.ndarray) -> List[int]:
    """Indices dropped by the post-push de-dup; push_multiscale_optimization.py:327-329."""
    _, uniq = np.unique(bank, axis=0, return_index=True)
    keep = set(int(i) for i in uniq)
    return [i for i in range(bank.shape[0]) if i not in keep]


def prune_state(
    prototypes_to_prune: Sequence[int],
    scale_ranges: Dict[int, Tuple[int, int]],
    num_scales: int,
    num_prototypes: int,
):
    """(keep list, new scale ranges) after dropping rows; model_multiscale.py:400-423."""
    drop = set(prototypes_to_prune)
    keep = sorted(set(range(num_prototypes)) - drop)
    new_ranges: Dict[int, Tuple[int, int]] = {}
    for s in range(num_scales):
        lo, hi = scale_ranges[s]
        n = len(set(range(lo, hi)) - drop)
        start = 0 if s == 0 else new_ranges[s - 1][1]
        new_ranges[s] = (start, start + n)
    return keep, new_ranges


# --------------------------------------------------------------------------- #
# KLD loss over the distance map and its class-gathered form
# --------------------------------------------------------------------------- #
def kld_loss(
    prototype_distances: torch.Tensor,
    target_labels: torch.Tensor,
    class_identity: torch.Tensor,
    num_scales: int,
    scale_ranges: Dict[int, Tuple[int, int]],
) -> torch.Tensor:
    """segmentation/model/loss.py:57-146 (KLDLoss.forward), loop for loop.

    prototype_distances [B,P,H,W]; target_labels [B,H,W] with 0 = void, 1..K = class (the loss subtracts 1,
    loss.py:73).  Per (image, class present, scale): log_softmax over the class's pixels of every prototype of
    that class and scale, symmetric KL between every pair, exp(-kld), mean over all terms (0.0 when none)."""
    tl = target_labels.reshape(target_labels.shape[0], -1) - 1
    d = prototype_distances.permute(0, 2, 3, 1)
    d = d.reshape(d.shape[0], -1, d.shape[-1])
    terms = []
    for b in range(tl.shape[0]):
        for c in torch.unique(tl[b]).tolist():
            if c < 0 or c >= class_identity.shape[1]:
                continue
            protos = torch.nonzero(class_identity[:, c]).flatten().tolist()
            if len(protos) == 0:
                continue
            mask = tl[b] == c
            for s in range(num_scales):
                ps = [p for p in protos if scale_ranges[s][0] <= p < scale_ranges[s][1]]
                logp = [F.log_softmax(torch.masked_select(d[b, :, p], mask), dim=0) for p in ps]
                if len(ps) < 2:
                    continue
                for j in range(len(ps)):
                    if len(logp[j]) < 2:
                        continue
                    for k in range(j + 1, len(ps)):
                        if len(logp[k]) < 2:
                            continue
                        k1 = F.kl_div(logp[j], logp[k], log_target=True, reduction="sum")
                        k2 = F.kl_div(logp[k], logp[j], log_target=True, reduction="sum")
                        terms.append((k1 + k2) / 2.0)
    if not terms:
        return torch.tensor(0.0)
    return torch.exp(-torch.stack(terms)).mean()


def kld_loss_group(
    list_group_activation: Sequence[torch.Tensor],
    target_labels: torch.Tensor,
    class_identity: torch.Tensor,
    group_class_identity_: torch.Tensor,
    num_groups: int,
) -> torch.Tensor:
    """segmentation/model/loss.py:478-545 (KLDLossGroup.forward), loop for loop: per (image, class present with
    prototypes): log_softmax over the class's pixels of each of the class's group activations, symmetric KL of every
    group pair, exp(-kld), mean."""
    tl = target_labels.reshape(target_labels.shape[0], -1) - 1
    terms = []
    for b in range(tl.shape[0]):
        for c in torch.unique(tl[b]).tolist():
            if c < 0 or c >= class_identity.shape[1]:
                continue
            if class_identity[:, c].sum() == 0:
                continue
            id_proj = int(group_class_identity_[:, c].argmax().item()) // num_groups
            ga = list_group_activation[id_proj].reshape(tl.shape[0], -1, num_groups)
            mask = tl[b] == c
            logp = [F.log_softmax(torch.masked_select(ga[b, :, i], mask), dim=0) for i in range(num_groups)]
            for j in range(num_groups):
                if len(logp[j]) < 2:
                    continue
                for k in range(j + 1, num_groups):
                    if len(logp[k]) < 2:
                        continue
                    k1 = F.kl_div(logp[j], logp[k], log_target=True, reduction="sum")
                    k2 = F.kl_div(logp[k], logp[j], log_target=True, reduction="sum")
                    terms.append((k1 + k2) / 2.0)
    if not terms:
        return torch.tensor(0.0)
    return torch.exp(-torch.stack(terms)).mean()


def class_slot_table(class_identity: torch.Tensor) -> torch.Tensor:
    """table [K, J]: prototype index of (class, slot), slot = rank among the class's prototypes (ascending), -1 = none."""
    P, K = class_identity.shape
    per = [torch.nonzero(class_identity[:, c]).flatten().tolist() for c in range(K)]
    J = max(1, max(len(x) for x in per))
    table = torch.full((K, J), -1, dtype=torch.long)
    for c in range(K):
        for j, p in enumerate(per[c]):
            table[c, j] = p
    return table


def gather_class_distances(prototype_distances: torch.Tensor, labels0: torch.Tensor, class_identity: torch.Tensor) -> torch.Tensor:
    """[B, H*W, J]: entry (px, j) = distance of px to prototype j of class labels0[px] (0..K-1; other = no class -> 0).
    Exactly the entries loss.py:89-107 selects with masked_select for that pixel."""
    B, P = prototype_distances.shape[:2]
    table = class_slot_table(class_identity)
    K, J = table.shape
    d = prototype_distances.reshape(B, P, -1).permute(0, 2, 1)          # [B, HW, P]
    lab = labels0.reshape(B, -1)
    ok = (lab >= 0) & (lab < K)
    idx = table[lab.clamp(0, K - 1)]                                     # [B, HW, J]
    valid = ok.unsqueeze(-1) & (idx >= 0)
    out = torch.gather(d, 2, idx.clamp(min=0))
    return torch.where(valid, out, torch.zeros_like(out))


# --------------------------------------------------------------------------- #
# evaluation maps
# --------------------------------------------------------------------------- #
def upsample_argext(src: torch.Tensor, size: Tuple[int, int], largest: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """segmentation/eval_valid_multiscale.py:229-234: F.interpolate(bilinear, align_corners=False) then argmin /
    argmax over the channels.  Returns (indices, extremum, upsampled tensor)."""
    up = F.interpolate(src, size=size, mode="bilinear", align_corners=False)
    val, idx = (up.max(dim=1) if largest else up.min(dim=1))
    return idx, val, up


def upsample_bilinear_restated(src: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """The arithmetic of torch's CPU ``upsample_bilinear2d`` (align_corners=False, fp32, contiguous NCHW:
    aten/src/ATen/native/cpu/UpSampleKernel.cpp, ``Interpolate<2>`` + ``HelperInterpLinear``) written out operation by
    operation, with the fused multiply-adds the shipped x86 builds contract it to - pinned here against
    ``F.interpolate`` bit for bit (tests/test_oracle_golden.py), so the HIP kernel can be held to it bit for bit:
        s  = fma(scale, dst + 0.5, -0.5) clamped at 0;  i0 = floor(s);  l1 = s - i0;  l0 = 1 - l1
        t0 = fma(v00, lx0, v01 * lx1);  t1 = fma(v10, lx0, v11 * lx1);  out = fma(t0, ly0, t1 * ly1)
    (an fma is emulated as a float64 product + sum rounded once to float32: the product of two float32 is exact in
    float64 and the 29 spare bits make the double rounding unobservable for these magnitudes)."""
    import numpy as np

    f32 = np.float32
    x = src.detach().cpu().numpy().astype(f32)
    N, C, h, w = x.shape
    H, W = int(size[0]), int(size[1])

    def fma(a, b, c):
        return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)

    def index(n_out, n_in):
        scale = f32(n_in) / f32(n_out)
        d = (np.arange(n_out).astype(f32) + f32(0.5)).astype(f32)
        s = np.maximum(fma(np.full_like(d, scale), d, np.full_like(d, -0.5)), f32(0)).astype(f32)
        i0 = np.minimum(s.astype(np.int64), n_in - 1)
        i1 = i0 + (i0 < n_in - 1)
        l1 = (s - i0.astype(f32)).astype(f32)
        return i0, i1, (f32(1) - l1).astype(f32), l1

    y0, y1, ly0, ly1 = index(H, h)
    x0, x1, lx0, lx1 = index(W, w)
    v00, v01 = x[:, :, y0][:, :, :, x0], x[:, :, y0][:, :, :, x1]
    v10, v11 = x[:, :, y1][:, :, :, x0], x[:, :, y1][:, :, :, x1]
    shp = v00.shape
    LX0, LX1 = np.broadcast_to(lx0[None, None, None, :], shp), np.broadcast_to(lx1[None, None, None, :], shp)
    LY0, LY1 = np.broadcast_to(ly0[None, None, :, None], shp), np.broadcast_to(ly1[None, None, :, None], shp)
    t0 = fma(v00, LX0, (v01 * LX1).astype(f32))
    t1 = fma(v10, LX0, (v11 * LX1).astype(f32))
    return torch.from_numpy(fma(t0, LY0, (t1 * LY1).astype(f32)))


# --------------------------------------------------------------------------- #
# helpers used by tests / bench
# --------------------------------------------------------------------------- #
def bf16_representable(t: torch.Tensor) -> torch.Tensor:
    """Round fp32 values to the nearest bf16-representable fp32 (SURVEY 8d 'identical inputs')."""
    return t.to(torch.bfloat16).to(torch.float32)


def fwd_bwd_reference(conv, prototype_vectors, scale_ranges, num_scales, last_layer_weight, g_logits, g_dist, g_act=None):
    """Autograd gradients of sum(logits*g_logits) + sum(dist*g_dist) [+ sum(act*g_act)]."""
    conv = conv.clone().requires_grad_(True)
    pv = prototype_vectors.clone().requires_grad_(True)
    w = last_layer_weight.clone().requires_grad_(True)
    logits, d, act = forward_from_conv_features(conv, pv, scale_ranges, num_scales, w)
    loss = (logits * g_logits).sum() + (d * g_dist).sum()
    if g_act is not None:
        loss = loss + (act * g_act).sum()
    loss.backward()
    return logits.detach(), d.detach(), act.detach(), conv.grad, pv.grad, w.grad
