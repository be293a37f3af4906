"""Loss of the hot path's distance output: KLDLoss (segmentation/model/loss.py:51-146), vectorised, and its
class-gathered form.

The reference's KLDLoss reads, for a pixel of class c, only the distance columns of class c's prototypes
(loss.py:89-107).  ``ClassDistances`` carries exactly those entries ([B, J, H*W] slot planes, produced by the fused kernels
with ``forward_from_conv_features(..., target_labels=...)``), so the fp32 [B, P, H, W] map and its gradient never
cross HBM.  ``KLDLoss`` accepts either form and returns the same value; it is plain torch (device-agnostic host
logic around the kernels' output), differentiable through ``ClassDistances.values``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple, Union

import torch
from torch import nn


@dataclass
class ClassDistances:
    """values [B, J, H*W] (slot planes: entry (j, px) = distance to prototype j of the pixel's class), labels [B, H*W] int
    (class 0..K-1, anything else = no class), table [K, J] prototype index of (class, slot) or -1."""

    values: torch.Tensor
    labels: torch.Tensor
    table: torch.Tensor
    grid: Tuple[int, int]


def class_slot_table(prototype_class_identity: torch.Tensor) -> torch.Tensor:
    """[K, J] prototype index of (class, slot); slot = rank among the class's prototypes (ascending index)."""
    ident = prototype_class_identity.detach().cpu()
    P, K = ident.shape
    per = [torch.nonzero(ident[:, c]).flatten().tolist() for c in range(K)]
    J = max(1, max(len(x) for x in per))
    table = torch.full((K, J), -1, dtype=torch.long)
    for c in range(K):
        for j, p in enumerate(per[c]):
            table[c, j] = p
    return table


def gather_class_distances(prototype_distances: torch.Tensor, labels0: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """torch gather of the P-wide map into the class-gathered form (used when KLDLoss is given the full map)."""
    B, P = prototype_distances.shape[:2]
    K, J = table.shape
    d = prototype_distances.reshape(B, P, -1).permute(0, 2, 1)
    lab = labels0.reshape(B, -1).long()
    ok = (lab >= 0) & (lab < K)
    idx = table.to(d.device)[lab.clamp(0, K - 1)]
    valid = ok.unsqueeze(-1) & (idx >= 0)
    out = torch.gather(d, 2, idx.clamp(min=0))
    return torch.where(valid, out, torch.zeros_like(out))


class KLDLoss(nn.Module):
    """Drop-in for segmentation/model/loss.py:51-146: same constructor, same ``forward(prototype_distances,
    target_labels)`` (labels 0 = void, 1..K = class); ``prototype_distances`` may be the [B, P, H, W] map or a
    ``ClassDistances``.  One pass of segment reductions instead of the reference's (image, class, scale, pair)
    Python loops with host syncs."""

    def __init__(self, prototype_class_identity: torch.Tensor, num_scales: int, scale_num_prototypes: Dict[int, Tuple[int, int]]) -> None:
        super().__init__()
        self.prototype_class_identity = prototype_class_identity
        self.num_scales = num_scales
        self.scale_num_prototypes = scale_num_prototypes

    def _pair_mask(self, table: torch.Tensor) -> torch.Tensor:
        """[K, J, J] bool: slots j < k of class c are prototypes of the same scale (loss.py:99-104, :118-121)."""
        K, J = table.shape
        scale = torch.full((K, J), -1, dtype=torch.long)
        t = table.cpu()
        for s in range(self.num_scales):
            lo, hi = self.scale_num_prototypes[s]
            scale[(t >= lo) & (t < hi)] = s
        same = (scale.unsqueeze(2) == scale.unsqueeze(1)) & (scale.unsqueeze(2) >= 0)
        upper = torch.triu(torch.ones(J, J, dtype=torch.bool), diagonal=1)
        return same & upper

    def forward(self, prototype_distances: Union[torch.Tensor, ClassDistances], target_labels: torch.Tensor) -> torch.Tensor:
        labels0 = target_labels.reshape(target_labels.shape[0], -1).long() - 1          # loss.py:73
        if isinstance(prototype_distances, ClassDistances):
            table = prototype_distances.table
            vals = prototype_distances.values.permute(0, 2, 1)          # [B, H*W, J] view
        else:
            table = class_slot_table(self.prototype_class_identity)
            vals = gather_class_distances(prototype_distances, labels0, table)
        dev = vals.device
        table = table.to(dev)
        K, J = table.shape
        B = vals.shape[0]
        lab = labels0.to(dev)
        ok = ((lab >= 0) & (lab < K)).reshape(-1)
        seg_all = (torch.arange(B, device=dev).unsqueeze(1) * K + lab.clamp(0, K - 1)).reshape(-1)
        sel = torch.nonzero(ok).flatten()
        if sel.numel() == 0:
            return torch.tensor(0.0)
        seg = seg_all[sel]                                               # (image, class) segment of each pixel
        d = vals.reshape(-1, J)[sel]                                     # [N, J]
        nseg = B * K
        count = torch.zeros(nseg, device=dev).index_add_(0, seg, torch.ones_like(seg, dtype=torch.float32))
        # log_softmax over the segment's pixels, per slot (loss.py:110)
        m = torch.full((nseg, J), float("-inf"), device=dev, dtype=d.dtype)
        m = m.scatter_reduce(0, seg.unsqueeze(1).expand(-1, J), d.detach(), reduce="amax", include_self=True)
        m = torch.where(torch.isfinite(m), m, torch.zeros_like(m))
        ssum = torch.zeros((nseg, J), device=dev, dtype=d.dtype).index_add_(0, seg, torch.exp(d - m[seg]))
        lse = m + torch.log(ssum.clamp_min(1e-38))
        logp = d - lse[seg]
        p = torch.exp(logp)
        # symmetric KL of every slot pair: 0.5 * sum_px (p_j - p_k)(logp_j - logp_k)   (loss.py:129-136)
        kld = torch.zeros((nseg, J, J), device=dev, dtype=d.dtype)
        step = max(1, (1 << 22) // (J * J))
        for i in range(0, d.shape[0], step):
            lp, pp, sg = logp[i:i + step], p[i:i + step], seg[i:i + step]
            term = 0.5 * (pp.unsqueeze(2) - pp.unsqueeze(1)) * (lp.unsqueeze(2) - lp.unsqueeze(1))
            kld = kld.index_add(0, sg, term)
        pair_ok = self._pair_mask(table).to(dev)                                    # [K, J, J]
        seg_cls = torch.arange(nseg, device=dev) % K
        valid = pair_ok[seg_cls] & (count >= 2).reshape(-1, 1, 1)                   # loss.py:113-127 (len < 2 skipped)
        terms = kld[valid]
        if terms.numel() == 0:
            return torch.tensor(0.0)
        return torch.exp(-terms).mean()                                              # loss.py:138-142
