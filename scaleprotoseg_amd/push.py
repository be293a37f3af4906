"""Prototype push on the MI355X kernels.

Mirrors the numerical part of segmentation/push_multiscale_optimization.py (lines 34-190 and 323-335):
``compute_distances`` -> ``min_across_dataset`` -> ``global_min`` -> commit + de-dup.  The plotting half
(``update_prototypes_on_image``, :341-685) is visualisation and is not part of this package.

Dataset protocol (image decoding / normalisation is the data layer, out of scope): ``len(dataset)`` and
``dataset[i] -> (image, target)`` with ``image`` a normalised float tensor [3, h, w] and ``target`` an
integer label map [h, w] (0 = void, 1..K), optionally ``dataset.convert_targets``.
"""
from __future__ import annotations

import json
import os
import time
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .functional import argmin_over_images, push_masked_argmin
from .utils import resize_label


@torch.no_grad()
def compute_distances(
    ppnet,
    dataset,
    img: torch.Tensor,
    target: np.ndarray,
    num_classes: int,
    max_dist: float = 1e10,
    device: Optional[str] = None,
    void_class: Optional[int] = None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-prototype class-masked minimum over one image -> (indices int64 [1,P], values f32 [1,P]).

    Same contract as push_multiscale_optimization.py:34-91; the one_hot / matmul / masked add / two min()
    passes are one HIP reduction (spx_push_argmin)."""
    device = device or str(ppnet.prototype_vectors.device)
    ppnet.eval()
    x = img.unsqueeze(0).to(device) if img.dim() == 3 else img.to(device)
    if dataset is not None and getattr(dataset, "convert_targets", None) is not None:
        target = dataset.convert_targets(target)
    fused = getattr(ppnet, "push_min_distances", None)
    if fused is not None:
        # the minimum is taken inside the distance kernel: the [1, P, H, W] map is never written (spx_dist_push_min)
        out = fused(x, lambda hw: resize_label(np.asarray(target), (hw[1], hw[0])).unsqueeze(0), void_class=void_class,
                    max_dist=max_dist)
        if out is not None:
            return out
    _, distances = ppnet(x, return_activations=False)
    lab = resize_label(np.asarray(target), (distances.shape[3], distances.shape[2])).unsqueeze(0)
    return push_masked_argmin(
        distances, lab, ppnet.prototype_class_identity, void_class=void_class, max_dist=max_dist
    )


def min_across_dataset(
    dataset,
    ppnet,
    num_classes: int,
    void_class: Optional[int] = None,
    device: Optional[str] = None,
    image_range: Optional[range] = None,
    return_values: bool = False,
):
    """(best image per prototype int64 [P], per-image flat indices) — push_multiscale_optimization.py:94-137.

    ``image_range`` restricts the scan to a shard of the image list (data-parallel push, see dp.py); the
    returned image ids are then positions inside the shard.  ``return_values=True`` appends the per-image minima
    ([n_images, P] fp32) the sharded reduction needs."""
    rng = image_range if image_range is not None else range(len(dataset))
    list_idx, list_val = [], []
    for i in rng:
        img, target = dataset[i]
        idx, val = compute_distances(ppnet, dataset, img, target, num_classes, void_class=void_class, device=device)
        list_idx.append(idx)
        list_val.append(val)
    tot = torch.cat(list_val, dim=0)
    best = argmin_over_images(tot)
    return (best, list_idx, tot) if return_values else (best, list_idx)


def _winning_patches(best: Sequence[int], list_min_patch, dataset, ppnet, device, image_offset: int = 0,
                     only: Optional[Sequence[bool]] = None) -> torch.Tensor:
    """[P, Cs] feature vectors of the winning latent pixels (rows outside ``only`` stay zero).  Each winning image is
    encoded once, the reference re-runs the backbone once per prototype."""
    P, S = ppnet.num_prototypes, ppnet.num_scales
    per_scale = P // S
    conv_cache: Dict[int, torch.Tensor] = {}
    out = None
    for p in range(P):
        if only is not None and not bool(only[p]):
            continue
        s = p // per_scale
        i = int(best[p])
        if i not in conv_cache:
            img, _ = dataset[image_offset + i]
            x = img.unsqueeze(0).to(device) if img.dim() == 3 else img.to(device)
            conv_cache[i] = ppnet.conv_features(x)
        conv = conv_cache[i]
        _, C, H, W = conv.shape
        cv = conv.view(S, C // S, H, W)
        flat = int(list_min_patch[i][:, p].item())
        r, c = flat // W, flat % W
        if out is None:
            out = torch.zeros((P, C // S), dtype=torch.float32, device=conv.device)
        out[p] = cv[s, :, r, c].detach().float()
    if out is None:
        cs = int(ppnet.prototype_shape[1])
        out = torch.zeros((P, cs), dtype=torch.float32, device=device)
    return out


@torch.no_grad()
def global_min(
    proto_min_dist: torch.Tensor,
    list_min_patch: Sequence[torch.Tensor],
    dataset,
    ppnet,
    device: Optional[str] = None,
    image_offset: int = 0,
) -> List[np.ndarray]:
    """Feature vector [Cs,1,1] of every prototype's winning latent pixel (push_multiscale_optimization.py:140-190)."""
    device = device or str(ppnet.prototype_vectors.device)
    rows = _winning_patches(proto_min_dist.tolist(), list_min_patch, dataset, ppnet, device, image_offset).cpu().numpy()
    return [rows[p].reshape(-1, 1, 1) for p in range(rows.shape[0])]


def commit_push(ppnet, patches: Sequence[np.ndarray], root_dir: Optional[os.PathLike] = None, log: Callable = print):
    """Overwrite the bank with the pushed patches and drop exact duplicates
    (push_multiscale_optimization.py:323-335; prune semantics model_multiscale.py:400-432)."""
    shape = tuple(ppnet.prototype_shape)
    update = np.reshape(patches, shape)
    ppnet.prototype_vectors.data.copy_(torch.tensor(update, dtype=torch.float32).to(ppnet.prototype_vectors.device))
    from .functional import invalidate_pack_cache

    invalidate_pack_cache()                  # (an in-place write through .data is invisible to the parameter's version counter)
    _, unique_index = np.unique(update, axis=0, return_index=True)
    keep = set(int(i) for i in unique_index)
    dup = [i for i in range(ppnet.num_prototypes) if i not in keep]
    log(f"Removing {len(dup)} duplicate prototypes.")
    ppnet.prune_prototypes(dup)
    if root_dir is not None:
        os.makedirs(root_dir, exist_ok=True)
        with open(os.path.join(root_dir, "unique_prototypes.json"), "w") as fp:
            json.dump([int(i) for i in sorted(unique_index)], fp)
    return dup


def _dp_world(group) -> Tuple[int, int]:
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def push_prototypes_multiscale(
    dataset,
    prototype_network_parallel,
    root_dir_for_saving_prototypes: Optional[os.PathLike] = None,
    log: Callable = print,
    device: Optional[str] = None,
    group=None,
    **_ignored,
):
    """Numerical part of push_multiscale_optimization.py:193-338 (plot/file-dump arguments are accepted and ignored).

    Data-parallel form (SURVEY.md 8e "Push"; new capability, the reference is single-process): when
    ``torch.distributed`` is initialised with more than one rank, every rank scans a contiguous shard of the image list
    (``dp.shard_range``), the per-prototype winners are combined with one all-gather and a lexicographic minimum on
    (value, global image index) - the reference's lowest-image tie-break (:137) - the winning feature vectors are
    assembled with one sum all-reduce (each row has exactly one contributor), and EVERY rank commits the same bank,
    de-dup and pruning; rank 0 alone writes ``unique_prototypes.json``.  Returns (best image per prototype [P] as GLOBAL
    image indices, the local shard's per-image flat indices, dropped duplicates)."""
    from . import dp

    net = prototype_network_parallel
    if hasattr(net, "module"):
        net = net.module
    net.eval()
    log("\tpush")
    start = time.time()
    num_classes = net.num_classes
    device = device or str(net.prototype_vectors.device)
    rank, world = _dp_world(group)
    if world == 1:
        best, tot_idx = min_across_dataset(dataset, net, num_classes, void_class=0, device=device)
        patches = global_min(best, tot_idx, dataset, net, device=device)
        dup = commit_push(net, patches, root_dir_for_saving_prototypes, log=log)
        log("\tpush time: \t{0}".format(time.time() - start))
        return best, tot_idx, dup

    P = net.num_prototypes
    rng = dp.shard_range(len(dataset), rank, world)
    dev = torch.device(device)
    ar = torch.arange(P, device=dev)
    if len(rng) > 0:
        best_local, tot_idx, tot_val = min_across_dataset(dataset, net, num_classes, void_class=0, device=device, image_range=rng,
                                                          return_values=True)                    # tot_val: [n_local, P]
        local_val = tot_val[best_local, ar]
        local_flat = torch.cat(list(tot_idx), dim=0)[best_local, ar]
    else:                                                              # more ranks than images: this rank never wins
        best_local = torch.zeros(P, dtype=torch.int64, device=dev)
        tot_idx = []
        local_val = torch.full((P,), float("inf"), dtype=torch.float32, device=dev)
        local_flat = torch.zeros(P, dtype=torch.int64, device=dev)
    gimg, _, gflat = dp.reduce_push_candidates(best_local, local_val, local_flat, rng.start, group=group)
    owner = (gimg >= rng.start) & (gimg < rng.stop)
    with torch.no_grad():
        local = _winning_patches((gimg - rng.start).tolist(), tot_idx, dataset, net, device, image_offset=rng.start,
                                 only=owner.tolist())
    full = dp.gather_push_patches(local, owner, group=group)           # [P, Cs], identical on every rank
    patches = full.cpu().numpy().reshape(tuple(net.prototype_shape))
    dup = commit_push(net, patches, root_dir_for_saving_prototypes if rank == 0 else None, log=log)
    log("\tpush time: \t{0}".format(time.time() - start))
    return gimg, tot_idx, dup
