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
    _, distances = ppnet(x, return_activations=False)
    if dataset is not None and getattr(dataset, "convert_targets", None) is not None:
        target = dataset.convert_targets(target)
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
) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """(best image per prototype int64 [P], per-image flat indices) — push_multiscale_optimization.py:94-137.

    ``image_range`` restricts the scan to a shard of the image list (data-parallel push, see dp.py); the
    returned image ids are then positions inside the shard."""
    rng = image_range if image_range is not None else range(len(dataset))
    list_idx, list_val = [], []
    for i in rng:
        img, target = dataset[i]
        idx, val = compute_distances(ppnet, dataset, img, target, num_classes, void_class=void_class, device=device)
        list_idx.append(idx)
        list_val.append(val)
    tot = torch.cat(list_val, dim=0)
    min_across_dataset.last_values = tot  # kept for the sharded reduction
    return argmin_over_images(tot), list_idx


@torch.no_grad()
def global_min(
    proto_min_dist: torch.Tensor,
    list_min_patch: Sequence[torch.Tensor],
    dataset,
    ppnet,
    device: Optional[str] = None,
    image_offset: int = 0,
) -> List[np.ndarray]:
    """Feature vector [Cs,1,1] of every prototype's winning latent pixel (push_multiscale_optimization.py:140-190).

    The reference re-runs the backbone once per prototype; here each winning image is encoded once
     and the P vectors are gathered on the GPU."""
    device = device or str(ppnet.prototype_vectors.device)
    P, S = ppnet.num_prototypes, ppnet.num_scales
    per_scale = P // S
    best = proto_min_dist.tolist()
    conv_cache: Dict[int, torch.Tensor] = {}
    out = []
    for p in range(P):
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
        out.append(cv[s, :, r : r + 1, c : c + 1].detach().float().cpu().numpy())
    return out


def commit_push(ppnet, patches: Sequence[np.ndarray], root_dir: Optional[os.PathLike] = None, log: Callable = print):
    """Overwrite the bank with the pushed patches and drop exact duplicates
    (push_multiscale_optimization.py:323-335; prune semantics model_multiscale.py:400-432)."""
    shape = tuple(ppnet.prototype_shape)
    update = np.reshape(patches, shape)
    ppnet.prototype_vectors.data.copy_(torch.tensor(update, dtype=torch.float32).to(ppnet.prototype_vectors.device))
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


def push_prototypes_multiscale(
    dataset,
    prototype_network_parallel,
    root_dir_for_saving_prototypes: Optional[os.PathLike] = None,
    log: Callable = print,
    device: Optional[str] = None,
    **_ignored,
):
    """Numerical part of push_multiscale_optimization.py:193-338 (plot/file-dump arguments are accepted and ignored)."""
    net = prototype_network_parallel
    if hasattr(net, "module"):
        net = net.module
    net.eval()
    log("\tpush")
    start = time.time()
    num_classes = net.num_classes
    best, tot_idx = min_across_dataset(dataset, net, num_classes, void_class=0, device=device)
    patches = global_min(best, tot_idx, dataset, net, device=device)
    dup = commit_push(net, patches, root_dir_for_saving_prototypes, log=log)
    log("\tpush time: \t{0}".format(time.time() - start))
    return best, tot_idx, dup
