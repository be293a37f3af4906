"""Loss of the hot path's distance output: KLDLoss (segmentation/model/loss.py:51-146), vectorised, and its
class-gathered form.

The reference's KLDLoss reads, for a pixel of class c, only the distance columns of class c's prototypes
(loss.py:89-107).  ``ClassDistances`` carries exactly those entries ([B, J, H*W] slot planes, produced by the fused kernels
with ``forward_from_conv_features(..., target_labels=...)``), so the fp32 [B, P, H, W] map and its gradient never
cross HBM.  ``KLDLoss`` accepts either form and returns the same value; on fp32 GPU tensors the pixel loops run in the
HIP kernels of csrc/spx_kld.hip (differentiable through ``ClassDistances.values``).  There is no other backend: inputs the
kernels do not take (CPU tensors, fp64, more than 16 slots per class) raise ``SpxError``.  (The torch restatement of the same
algebra that the tests hold the kernels against is test infrastructure and lives outside the package.)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple, Union

import torch
from torch import nn

from ._lib import SpxError


@dataclass
class ClassDistances:
    """values [B, J, H*W] (slot planes: entry (j, px) = distance to prototype j of the pixel's class), labels [B, H*W] int
    (class 0..K-1, anything else = no class), table [K, J] prototype index of (class, slot) or -1."""

    values: torch.Tensor
    labels: torch.Tensor
    table: torch.Tensor
    grid: Tuple[int, int]
    # the target map ``labels`` was derived from (labels = target - 1) and its version then: a loss that is handed this very
    # tensor takes ``labels`` as they are instead of shifting and converting the map a second time
    target: Optional[torch.Tensor] = None
    target_version: int = -1


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


class PixelWiseCrossEntropyLoss(nn.Module):
    """Drop-in for segmentation/model/loss.py:9-48: cross entropy over the [..., K] logits with labels shifted by one
    (1..K -> 0..K-1; the training modules pass ``ignore_index=-1`` so that void = 0 is skipped,
    module_multiscale.py:162-164), optionally with the per-pixel correctness of the non-ignored pixels.

    On GPU tensors the loss runs in HIP: if the logits come from ``forward_from_conv_features(..., ce_target=target)``
    the value was already computed in the logits epilogue (``logits.spx_ce``) and is returned as is;
    otherwise the stand-alone kernels of csrc/spx_ce.hip run.  Labels outside 0..K-1 other than ``ignore_index`` make
    torch raise; here they are ignored.  Logits that are not on the GPU raise ``SpxError``: there is no other backend."""

    def __init__(self, ignore_index: int = 255, return_correct: bool = False) -> None:
        super().__init__()
        self.return_correct = return_correct
        self.ignore_index = ignore_index

    def forward(self, predicted_logits: torch.Tensor, target_labels: torch.Tensor):
        fused = getattr(predicted_logits, "spx_ce", None)
        K = predicted_logits.size(-1)
        if predicted_logits.is_cuda:
            from .functional import cross_entropy_from_logits

            ignores_a_class = self.ignore_index is not None and 0 <= self.ignore_index < K
            stale = fused is not None and (fused.target is not target_labels or fused.target_version != target_labels._version)
            if fused is not None and not stale and not ignores_a_class and not self.return_correct:
                return fused.loss                       # the epilogue's value: not one more launch here
            labels0 = target_labels.reshape(-1).to(predicted_logits.device) - 1                  # loss.py:32
            if fused is None or stale or ignores_a_class:
                lab = labels0 if not ignores_a_class else torch.where(labels0 == self.ignore_index, torch.full_like(labels0, -1), labels0)
                fused = cross_entropy_from_logits(predicted_logits, lab)
            if not self.return_correct:
                return fused.loss
            correct = fused.pred.reshape(-1).to(labels0.dtype) == labels0
            mask = (labels0 != self.ignore_index).nonzero().squeeze()                             # loss.py:43-46
            return fused.loss, correct[mask]
        raise SpxError(f"cross entropy: logits on {predicted_logits.device}; the loss runs on the GPU only (no CPU fallback)")


def _kld_kernels_usable(vals: torch.Tensor, K: int, J: int) -> bool:
    # fp32 planes on the GPU, at most 16 slots per class; the [K, J] segment tables of the reduction passes must fit the LDS
    # (K*J*12 + K*4 bytes <= 60 KiB: 150 x 12 and 182 x 12 of the reference's ADE / COCO configs take 22 / 27 KiB); the
    # [K, J, J] tables of the pair and gradient passes are tiled over class blocks and set no limit
    return vals.is_cuda and vals.dtype == torch.float32 and J <= 16 and K * J * 12 + K * 4 + 8 <= 60 * 1024


def _kld_segment_passes(lib, v, lab, K, Wk, s):
    """The three reduction passes over the gathered planes (csrc/spx_kld.hip): (a_fx int64 [B,K,J,J], counts [B,K], lse [B,K,J],
    scale double [1]).  One zero-filled workspace holds the integer tables: [a_fx | ssum_fx | scale | keys | counts | range keys];
    the fixed-point scale of the pair sums is derived on the device (segment-lse kernel) from the value range pass 0 collects:
    |p_j (l_k - l_j)| is bounded by twice the range, and HW terms must stay inside int64 - no host sync, no torch glue."""
    from . import _lib

    B, J, HW = v.shape
    dev = v.device
    n_a, n_s = B * K * J * J, B * K * J
    ws = torch.zeros(n_a + n_s + 1 + (n_s + B * K + 2 + 1) // 2, dtype=torch.int64, device=dev)
    a_fx, ssum_fx = ws[:n_a].view(B, K, J, J), ws[n_a:n_a + n_s]
    scale = ws[n_a + n_s:n_a + n_s + 1].view(torch.float64)
    tail32 = ws[n_a + n_s + 1:].view(torch.int32)
    keys, counts, rng = tail32[:n_s], tail32[n_s:n_s + B * K].view(B, K), tail32[n_s + B * K:n_s + B * K + 2]
    _lib.check(lib.spx_kld_segment_max(_lib.ptr(v), _lib.ptr(lab), B, J, HW, Wk, K, _lib.ptr(keys), _lib.ptr(counts), _lib.ptr(rng), s))
    _lib.check(lib.spx_kld_segment_sumexp(_lib.ptr(v), _lib.ptr(lab), B, J, HW, Wk, K, _lib.ptr(keys), _lib.ptr(ssum_fx), s))
    lse = torch.empty((B, K, J), dtype=torch.float32, device=dev)
    _lib.check(lib.spx_kld_segment_lse(_lib.ptr(keys), _lib.ptr(ssum_fx), n_s, _lib.ptr(lse), _lib.ptr(rng), HW, _lib.ptr(scale), s))
    _lib.check(lib.spx_kld_pair_sums(_lib.ptr(v), _lib.ptr(lab), B, J, HW, Wk, K, _lib.ptr(lse), _lib.ptr(scale), _lib.ptr(a_fx), s))
    return a_fx, counts, lse, scale


class _KLDFusedLoss(torch.autograd.Function):
    """The whole loss of class-gathered planes on the GPU in SIX launches (workspace fill, segment max + value range, segment
    sum-exp, lse + fixed-point scale, pair sums, and spx_kld_gram_loss: symmetric KL of the slot pairs, exp(-kld), mean and its
    gradient with respect to the pair sums, loss.py:113-142); backward: one scalar multiply + the per-pixel gradient pass."""

    @staticmethod
    def forward(ctx, vals, labels, K, W, pair_ok):
        from . import _lib

        lib = _lib.load()
        B, J, HW = vals.shape
        v = vals.detach().contiguous()
        lab = labels.to(device=v.device, dtype=torch.int32).contiguous()
        s = _lib.stream_ptr()
        Wk = int(W) if W and HW % int(W) == 0 else 0
        a_fx, counts, lse, scale = _kld_segment_passes(lib, v, lab, K, Wk, s)
        n = B * K * J * J
        out = torch.empty((2 * n + 2,), dtype=torch.float32, device=v.device)
        A, cf, loss = out[:n], out[n:2 * n], out[2 * n:]                       # loss = (value, 1 / number of valid pairs)
        part = torch.empty((2 * B * K,), dtype=torch.float64, device=v.device)
        _lib.check(lib.spx_kld_gram_loss(_lib.ptr(a_fx), _lib.ptr(scale), _lib.ptr(counts), _lib.ptr(pair_ok), B * K, K, J,
                                         _lib.ptr(A), _lib.ptr(cf), _lib.ptr(part), _lib.ptr(loss), s))
        ctx.save_for_backward(v, lab, lse, A, cf, loss)
        ctx.K = K
        return loss[0].reshape(())

    @staticmethod
    def backward(ctx, g):
        from . import _lib

        lib = _lib.load()
        v, lab, lse, A, cf, loss = ctx.saved_tensors
        B, J, HW = v.shape
        grad = torch.empty_like(v)
        coef = (g.reshape(()).float() * loss[1]).reshape(1).contiguous()       # dLoss_total/dloss x 1/n, on the device
        _lib.check(lib.spx_kld_backward(_lib.ptr(v), _lib.ptr(lab), B, J, HW, ctx.K, _lib.ptr(lse), _lib.ptr(A), _lib.ptr(cf), _lib.ptr(coef),
                                        _lib.ptr(grad), _lib.stream_ptr()))
        return grad, None, None, None, None


def segment_pair_sums(planes: torch.Tensor, labels0: torch.Tensor, K: int, W: int = 0) -> torch.Tensor:
    """A [B, K, J, J] = sum over the segment's pixels of p_j (l_k - l_j) (= -KL(j || k); diagonal 0) of class-gathered planes
    [B, J, H*W] through the three reduction passes of csrc/spx_kld.hip - what ``KLDLoss`` builds its value from (no autograd;
    diagnostics and tests).  ``W``: row length of the pixel grid (0 = unknown: linear walk)."""
    from . import _lib

    lib = _lib.load()
    v = planes.detach().contiguous()
    B, J, HW = v.shape
    if not _kld_kernels_usable(v, K, J):
        raise SpxError(f"segment_pair_sums: input {tuple(v.shape)} {v.dtype} on {v.device} is outside the HIP kernels' domain")
    lab = labels0.to(device=v.device, dtype=torch.int32).contiguous()
    Wk = int(W) if W and HW % int(W) == 0 else 0
    a_fx, _, _, scale = _kld_segment_passes(lib, v, lab, K, Wk, _lib.stream_ptr())
    return (a_fx.to(torch.float64) / scale).float()


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

    def _slot_table(self) -> torch.Tensor:
        """class_slot_table of the loss's identity, built once per identity OBJECT and in-place version (a strong
        reference is kept, so neither an id nor a device address can be recycled under the cache)."""
        ident = self.prototype_class_identity
        c = getattr(self, "_slot_table_cache", None)
        if c is None or c[0] is not ident or c[1] != ident._version:
            c = (ident, ident._version, class_slot_table(ident))
            self._slot_table_cache = c
        return c[2]

    def _pair_mask(self, table: torch.Tensor) -> torch.Tensor:
        """[K, J, J] bool: slots j < k of class c are prototypes of the same scale (loss.py:99-104, :118-121).
        Cached per (table object + its in-place version, scale table): it is host-side work with a device read-back,
        not something to redo per step.  The cache holds the table itself, so the key cannot alias a recycled tensor."""
        scales = tuple(sorted((int(s), tuple(r)) for s, r in self.scale_num_prototypes.items()))
        cached = getattr(self, "_pair_mask_cache", None)
        if cached is not None and cached[0] is table and cached[1] == (table._version, scales):
            return cached[2]
        mask = self._pair_mask_build(table).to(table.device)
        self._pair_mask_cache = (table, (table._version, scales), mask)
        return mask

    def _table_on(self, table: torch.Tensor, dev) -> torch.Tensor:
        """``table`` on ``dev`` (the same object every step, so the pair-mask cache keyed on it holds)."""
        if table.device == dev:
            return table
        c = getattr(self, "_table_dev_cache", None)
        if c is None or c[0] is not table or c[1] != (table._version, str(dev)):
            c = (table, (table._version, str(dev)), table.to(dev))
            self._table_dev_cache = c
        return c[2]

    def _pair_mask_build(self, table: torch.Tensor) -> torch.Tensor:
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
        cd = prototype_distances if isinstance(prototype_distances, ClassDistances) else None
        if cd is not None and cd.target is target_labels and cd.target_version == target_labels._version:
            labels0 = cd.labels                                                           # already target - 1 (int32)
        else:
            labels0 = target_labels.reshape(target_labels.shape[0], -1).long() - 1      # loss.py:73
        if isinstance(prototype_distances, ClassDistances):
            table = prototype_distances.table
            vals = prototype_distances.values.permute(0, 2, 1)          # [B, H*W, J] view
        else:
            table = self._slot_table()
            vals = gather_class_distances(prototype_distances, labels0, table)
        planes = prototype_distances.values if isinstance(prototype_distances, ClassDistances) else None
        width = prototype_distances.grid[-1] if isinstance(prototype_distances, ClassDistances) else prototype_distances.shape[-1]
        return self._forward_gathered(vals, planes, labels0, table, width)

    def _forward_gathered(self, vals: torch.Tensor, planes, labels0: torch.Tensor, table: torch.Tensor, width: int = 0) -> torch.Tensor:
        """Loss from the class-gathered values ``vals`` [B, H*W, J] (``planes``: the same as [B, J, H*W], if the caller
        already holds that layout; ``width``: W of the pixel grid if known, a traversal hint for the kernels)."""
        dev = vals.device
        table = self._table_on(table, dev)
        K, J = table.shape
        B = vals.shape[0]
        lab = labels0.to(dev)
        nseg = B * K
        if planes is None and _kld_kernels_usable(vals, K, J):
            planes = vals.permute(0, 2, 1).contiguous()      # full map given: its gathered entries as [B, J, H*W] planes
        if planes is not None and _kld_kernels_usable(planes, K, J):
            # the gathered planes on the GPU: segment statistics and the gradient run in the HIP kernels; nothing on
            # this path reads a value back to the host (capturable in a HIP graph)
            return _KLDFusedLoss.apply(planes, lab, K, width, self._pair_mask_u8(table, dev))
        raise SpxError(
            f"KLD loss: input {tuple(vals.shape)} {vals.dtype} on {vals.device} (K={K}, J={J}) is outside the HIP kernels' "
            "domain (fp32 on the GPU, J <= 16, K*J*12 <= 60 KiB); there is no other backend"
        )

    def _pair_mask_u8(self, table: torch.Tensor, dev) -> torch.Tensor:
        """The [K, J, J] pair mask as uint8 on ``dev`` (cached with the mask it is made from)."""
        m = self._pair_mask(table)
        c = getattr(self, "_pair_u8_cache", None)
        if c is None or c[0] is not m or c[1].device != torch.device(dev):
            c = (m, m.to(device=dev, dtype=torch.uint8).contiguous())
            self._pair_u8_cache = c
        return c[1]


class KLDLossGroup(KLDLoss):
    """Drop-in for segmentation/model/loss.py:461-545: same constructor, same ``forward(list_group_activation,
    target_labels)``.  The groups of a pixel's class play the role KLDLoss gives the class's prototypes (every group
    pair of a class is compared, loss.py:527-536), so the segment kernels are shared; ``list_group_activation`` may
    also be the concatenated [M, n_projections * num_groups] tensor the grouping head produces."""

    def __init__(self, prototype_class_identity: torch.Tensor, group_class_identity: torch.Tensor, num_groups: int) -> None:
        nn.Module.__init__(self)
        self.prototype_class_identity = prototype_class_identity
        self.group_class_identity = group_class_identity
        self.num_groups = num_groups
        self._tables = None

    def _class_tables(self):
        """(projection of class c or -1 [K], stand-in slot table [K, G]: slot ids where the class has a projection)."""
        if self._tables is None:
            ident, gci, G = self.prototype_class_identity, self.group_class_identity, self.num_groups
            K = ident.shape[1]
            has = ident.sum(dim=0) > 0                                              # loss.py:504
            proj = torch.where(has, gci.argmax(dim=0) // G, torch.full((K,), -1, dtype=torch.long)).cpu()   # :507
            table = torch.where(has.cpu().unsqueeze(1), torch.arange(G).unsqueeze(0).expand(K, G), torch.full((K, G), -1, dtype=torch.long))
            self._tables = (proj, table.contiguous())
        return self._tables

    def _pair_mask_build(self, table: torch.Tensor) -> torch.Tensor:
        K, J = table.shape
        upper = torch.triu(torch.ones(J, J, dtype=torch.bool), diagonal=1)
        return upper.unsqueeze(0) & (table.cpu()[:, :1] >= 0).unsqueeze(2)

    def _pair_mask(self, table: torch.Tensor) -> torch.Tensor:
        cached = getattr(self, "_pair_mask_cache", None)
        if cached is None or cached[0] is not table or cached[1] != table._version:
            cached = (table, table._version, self._pair_mask_build(table).to(table.device))
            self._pair_mask_cache = cached
        return cached[2]

    def _gather_groups(self, list_group_activation, target_labels: torch.Tensor):
        """(vals [B, HW, G]: the group activations of the pixel's class, labels0 [B, HW], stand-in slot table [K, G])."""
        G = self.num_groups
        labels0 = target_labels.reshape(target_labels.shape[0], -1).long() - 1          # loss.py:493
        B, HW = labels0.shape
        proj, table = self._class_tables()
        K = table.shape[0]
        if isinstance(list_group_activation, torch.Tensor):
            ga = list_group_activation.reshape(B, HW, -1, G)
        else:
            ga = torch.stack([a.reshape(B, HW, G) for a in list_group_activation], dim=2)
        dev = ga.device
        lab = labels0.to(dev)
        pix_proj = proj.to(dev)[lab.clamp(0, K - 1)].clamp_min(0)                       # [B, HW]; unused where no class
        vals = torch.gather(ga, 2, pix_proj.view(B, HW, 1, 1).expand(B, HW, 1, G)).squeeze(2)      # [B, HW, G]
        return vals, labels0, table

    def forward(self, list_group_activation, target_labels: torch.Tensor) -> torch.Tensor:
        vals, labels0, table = self._gather_groups(list_group_activation, target_labels)
        return self._forward_gathered(vals, None, labels0, table, target_labels.shape[-1] if target_labels.dim() >= 3 else 0)
