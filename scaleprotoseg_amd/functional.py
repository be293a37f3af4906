"""Autograd operators over the libspx_hip.so C ABI.

`proto_head_forward` is the fused replacement of the reference's
``_scale_l2_convolution`` -> ``distance_2_similarity`` -> ``last_layer`` chain
(segmentation/model/model_multiscale.py:255-330, :243-244, :362-376) and of its
autograd backward.  PyTorch is used for device memory, streams and autograd
plumbing only; all arithmetic on the path runs in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import SpxError, SpxPlan

ACT_FN = {"log": 0, "linear": 1}

# Optional per-operator timing used by bench.py: when a list is installed here, every C-ABI operator call is
# bracketed by HIP events recorded on the stream the kernel is launched on (torch's current stream).
_PROFILE: Optional[list] = None


def set_profile(sink: Optional[list]) -> None:
    global _PROFILE
    _PROFILE = sink


class _timed:
    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        if _PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _PROFILE is not None:
            self.e1.record()
            _PROFILE.append((self.name, self.e0, self.e1))
        return False


_PLAN_CACHE: Dict[tuple, SpxPlan] = {}


@dataclass
class BankLayout:
    """Static description of the prototype bank that the kernels are planned for."""

    num_prototypes: int
    num_classes: int          # rows of the dense head matrix fed to the kernel
    num_scales: int
    channels_per_scale: int
    scale_ranges: Tuple[Tuple[int, int], ...]

    def plan(self) -> SpxPlan:
        """The kernels' panel plan (spx_make_plan); cached per layout value - it is pure host work."""
        key = (self.num_prototypes, self.num_classes, self.num_scales, self.channels_per_scale,
               tuple(tuple(int(v) for v in r) for r in self.scale_ranges))
        plan = _PLAN_CACHE.get(key)
        if plan is None:
            lo = [r[0] for r in self.scale_ranges]
            hi = [r[1] for r in self.scale_ranges]
            plan = _lib.make_plan(self.num_prototypes, self.num_classes, self.num_scales, self.channels_per_scale, lo, hi)
            if len(_PLAN_CACHE) > 64:
                _PLAN_CACHE.clear()
            _PLAN_CACHE[key] = plan
        return plan


@dataclass
class FusedCrossEntropy:
    """Cross entropy computed in the logits epilogue: ``loss`` (scalar, differentiable: mean over the non-ignored pixels,
    segmentation/model/loss.py:9-48), ``pred`` int32 [B*H*W] (argmax class), ``labels`` int32 [B, H*W] it was computed for
    (class 0..K-1; anything else was ignored)."""

    loss: torch.Tensor
    pred: torch.Tensor
    labels: torch.Tensor
    target: Optional[torch.Tensor] = None      # the caller's label tensor (0 = void, 1..K) the labels were derived from
    target_version: int = -1                   # ... and its version counter at that time


def cross_entropy_from_logits(logits: torch.Tensor, labels0: torch.Tensor) -> FusedCrossEntropy:
    """The same cross entropy as stand-alone HIP kernels over any [..., K] logits tensor on the GPU (heads the fused
    kernels do not carry: more than 160 classes, the grouping tail).  ``labels0``: class 0..K-1, anything else ignored."""
    loss, pred, lab = _CrossEntropyFn.apply(logits, labels0)
    return FusedCrossEntropy(loss, pred, lab)


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels0):
        lib = _lib.load()
        if not logits.is_cuda:
            raise SpxError("scaleprotoseg_amd runs on an AMD GPU only; there is no CPU fallback")
        K = int(logits.shape[-1])
        lg = logits.detach().reshape(-1, K).contiguous().float()
        M = int(lg.shape[0])
        lab = labels0.to(device=lg.device, dtype=torch.int32).reshape(-1).contiguous()
        if lab.numel() != M:
            raise SpxError(f"{lab.numel()} labels for {M} logit rows")
        lse = torch.empty((M,), dtype=torch.float32, device=lg.device)
        pred = torch.empty((M,), dtype=torch.int32, device=lg.device)
        partials = torch.empty((lib.spx_ce_partials_flat(M), 2), dtype=torch.float32, device=lg.device)
        _lib.check(lib.spx_ce_fwd(_lib.ptr(lg), _lib.ptr(lab), M, K, _lib.ptr(lse), _lib.ptr(pred), _lib.ptr(partials), _lib.stream_ptr()))
        loss, tot = _ce_finish(partials)                             # mean, (count, sum)
        ctx.save_for_backward(lg, lse, lab, tot)
        ctx.shape = tuple(logits.shape)
        ctx.mark_non_differentiable(pred, lab)
        return loss, pred, lab

    @staticmethod
    def backward(ctx, g_loss, _g_pred=None, _g_lab=None):
        lib = _lib.load()
        lg, lse, lab, tot = ctx.saved_tensors
        coef = (g_loss.float() / tot[0]).reshape(1).contiguous()
        dl = torch.empty_like(lg)
        _lib.check(lib.spx_ce_bwd(_lib.ptr(lg), _lib.ptr(lse), _lib.ptr(lab), _lib.ptr(coef), lg.shape[0], lg.shape[1], _lib.ptr(dl), _lib.stream_ptr()))
        return dl.reshape(ctx.shape), None


@dataclass
class ClassGather:
    """Class-gathered distance mode: per pixel, only the distances to its own class's prototypes.

    ``labels`` int32 [B, H*W] (class 0..K-1, anything else = no class), ``keys`` int32 [npanels * 32 npb] in the
    plan's padded row order ((class << 16) | slot, -1 = none), ``width`` = J = slots per pixel,
    ``table`` int64 [K, J]: prototype index of (class, slot) or -1.
    """

    labels: torch.Tensor
    keys: torch.Tensor
    width: int
    table: torch.Tensor


def class_gather_table(layout: BankLayout, prototype_class_identity: torch.Tensor, device) -> Tuple[torch.Tensor, int, torch.Tensor]:
    """(keys, J, table) for ClassGather from the module's prototype_class_identity [P, K]
    (segmentation/model/model_multiscale.py:132-141: one-hot rows, possibly all-zero rows)."""
    ident = prototype_class_identity.detach().cpu()
    P, K = ident.shape
    if P != layout.num_prototypes:
        raise SpxError(f"class identity has {P} rows, bank has {layout.num_prototypes}")
    has = ident.sum(dim=1) > 0
    cls = torch.where(has, ident.argmax(dim=1), torch.full((P,), -1, dtype=torch.long))
    slot = torch.zeros(P, dtype=torch.long)
    counts = [0] * K
    for p in range(P):
        c = int(cls[p])
        if c >= 0:
            slot[p] = counts[c]
            counts[c] += 1
    J = max(1, max(counts) if counts else 1)
    table = torch.full((K, J), -1, dtype=torch.long)
    for p in range(P):
        if int(cls[p]) >= 0:
            table[int(cls[p]), int(slot[p])] = p
    plan = layout.plan()
    rows = plan.npb * 32
    keys = torch.full((plan.npanels * rows,), -1, dtype=torch.int64)
    for q in range(plan.npanels):
        p0, n = plan.panel_p0[q], plan.panel_np[q]
        for r_ in range(n):
            p = p0 + r_
            if int(cls[p]) >= 0:
                keys[q * rows + r_] = (int(cls[p]) << 16) | int(slot[p])
    keys = torch.where(keys < 0, torch.full_like(keys, 0xFFFFFFFF), keys)
    keys32 = (keys & 0xFFFFFFFF).to(torch.int64)
    keys32 = torch.where(keys32 >= 2**31, keys32 - 2**32, keys32).to(torch.int32)   # same bits as uint32
    return keys32.to(device), J, table.to(device)


def _x_dtype_code(x: torch.Tensor) -> int:
    if x.dtype == torch.bfloat16:
        return 0
    if x.dtype == torch.float32:
        return 1
    raise SpxError(f"features must be bfloat16 or float32, got {x.dtype}")


def _check_x(x: torch.Tensor, layout: BankLayout) -> Tuple[int, int]:
    if x.dim() != 4:
        raise SpxError(f"features must be [B, C, H, W], got shape {tuple(x.shape)}")
    B, Cx, H, W = x.shape
    if B < 1 or H * W < 1:
        raise SpxError(f"empty input: features have shape {tuple(x.shape)}")
    if Cx != layout.num_scales * layout.channels_per_scale:
        raise SpxError(
            f"features have {Cx} channels, prototype bank expects "
            f"{layout.num_scales} x {layout.channels_per_scale}"
        )
    return B, H * W


def _rows_gemm(A: torch.Tensor, a_strides, B: torch.Tensor, b_strides, M: int, N: int, K: int, flags: int = 0,
               E: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C[i][j] = sum_k A(i, k) B(j, k) on row-major fp32 device tensors through the fp32 MFMA kernels of csrc/spx_gemm.hip
    (include/spx_hip.h: spx_rows_gemm); ``*_strides`` = (row stride, k stride) in elements."""
    if not (A.is_cuda and B.is_cuda):
        raise SpxError("scaleprotoseg_amd runs on an AMD GPU only; there is no CPU fallback")
    if A.dtype != torch.float32 or B.dtype != torch.float32 or not A.is_contiguous() or not B.is_contiguous():
        raise SpxError("spx_rows_gemm takes contiguous float32 operands")
    lib = _lib.load()
    out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    nws = lib.spx_rows_gemm_workspace_bytes(M, N, K, flags)
    ws = torch.empty(nws, dtype=torch.uint8, device=A.device) if nws else None
    _lib.check(lib.spx_rows_gemm(_lib.ptr(A), a_strides[0], a_strides[1], _lib.ptr(B), b_strides[0], b_strides[1], _lib.ptr(out), N,
                                 M, N, K, flags, _lib.ptr(E), (int(E.shape[1]) if E is not None else 0), _lib.ptr(ws),
                                 _lib.stream_ptr()))
    return out


def _pixel_outer(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a^T . b for tall-skinny [M, n] operands (M = pixels).  Up to 8192 output elements (d W_g of the grouping tail:
    19 x 57) in the fp32 FMA kernel spx_pixel_outer; larger ones (the heads wider than the fused kernels: 450 x 1800) in the
    MFMA product kernel with the pixels as the contraction index (split over workgroups, slabs summed in a fixed order)."""
    M, n1, n2 = a.shape[0], a.shape[1], b.shape[1]
    if not a.is_cuda:
        raise SpxError("scaleprotoseg_amd runs on an AMD GPU only; there is no CPU fallback")
    a, b = a.contiguous().float(), b.contiguous().float()
    if n1 * n2 <= 8192:
        lib = _lib.load()
        out = torch.empty((n1, n2), dtype=torch.float32, device=a.device)
        ws = torch.empty(lib.spx_pixel_outer_workspace_bytes(M, n1, n2), dtype=torch.uint8, device=a.device)
        _lib.check(lib.spx_pixel_outer(_lib.ptr(a), _lib.ptr(b), M, n1, n2, _lib.ptr(out), _lib.ptr(ws), _lib.stream_ptr()))
        return out
    return _rows_gemm(a, (1, n1), b, (1, n2), n1, n2, M)


# Limits of the head product fused into the distance kernels (class blocks of 32 rows: 1, 2 or 5 per instance)
MAX_FUSED_HEAD_ROWS = 160
MAX_FUSED_TAIL_CLASSES = 32


class _WideLinearFn(torch.autograd.Function):
    """y = a . w^T for heads wider than the fused kernels carry (more than 160 rows: scaleproto_coco.gin's 182 classes,
    the dense grouping heads of group_scaleproto_ade.gin / _coco.gin with 450 / 546 rows) and for the head behind a
    user-supplied similarity.  The [pixel][P] activations come out of the distance kernel once; the three products of the
    layer (y, d_a = g . w, d_w = g^T . a) are the hand-written fp32 MFMA kernels of csrc/spx_gemm.hip - fp32 operands, so
    these heads carry the reference's fp32 arithmetic (segmentation/model/model_multiscale.py:243-244)."""

    @staticmethod
    def forward(ctx, a, w):
        a, w = a.contiguous().float(), w.contiguous().float()
        ctx.save_for_backward(a, w)
        M, P = a.shape
        return _rows_gemm(a, (P, 1), w, (P, 1), M, int(w.shape[0]), P)

    @staticmethod
    def backward(ctx, g):
        a, w = ctx.saved_tensors
        g = g.contiguous().float()
        M, P = a.shape
        N = int(w.shape[0])
        da = _rows_gemm(g, (N, 1), w, (1, P), M, P, N) if ctx.needs_input_grad[0] else None
        dw = _pixel_outer(g, a) if ctx.needs_input_grad[1] else None
        return da, dw


class _WideGroupTailFn(torch.autograd.Function):
    """logits = exp(units) . W_g^T for a grouping tail over more than 32 classes (group_scaleproto_ade.gin: 150,
    _coco.gin: 182; segmentation/model/model_multiscale_group.py:303-308): the exponential is applied while the operand is
    staged, the backward's d_units = (g . W_g) * exp(units) in the product's epilogue, d_W_g = g^T . exp(units) with the
    pixels as the contraction index - no [pixel][units] temporary besides the saved units."""

    @staticmethod
    def forward(ctx, units, wg):
        units, wg = units.contiguous().float(), wg.contiguous().float()
        ctx.save_for_backward(units, wg)
        M, U = units.shape
        return _rows_gemm(units, (U, 1), wg, (U, 1), M, int(wg.shape[0]), U, flags=1)

    @staticmethod
    def backward(ctx, g):
        units, wg = ctx.saved_tensors
        g = g.contiguous().float()
        M, U = units.shape
        K2 = int(wg.shape[0])
        du = _rows_gemm(g, (K2, 1), wg, (1, U), M, U, K2, flags=4, E=units) if ctx.needs_input_grad[0] else None
        dwg = _rows_gemm(g, (1, K2), units, (1, U), K2, U, M, flags=2) if ctx.needs_input_grad[1] else None
        return du, dwg


class _ExpFn(torch.autograd.Function):
    """g = exp(units) as a HIP elementwise kernel pair (spx_exp / spx_exp_bwd): the group activations of compute_group
    (segmentation/model/model_multiscale_group.py:299-300) outside the fused forward."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = x.contiguous().float()
        y = torch.empty_like(x)
        _lib.check(lib.spx_exp(_lib.ptr(x), _lib.ptr(y), x.numel(), _lib.stream_ptr()))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (y,) = ctx.saved_tensors
        g = g.contiguous().float()
        dx = torch.empty_like(y)
        _lib.check(lib.spx_exp_bwd(_lib.ptr(g), _lib.ptr(y), _lib.ptr(dx), y.numel(), _lib.stream_ptr()))
        return dx


def _ce_finish(partials: torch.Tensor) -> torch.Tensor:
    """(mean loss [], (count, loss sum) [2]) of the kernels' (sum, count) partial pairs in one launch (spx_ce_finish); two
    allocations: the loss becomes an autograd OUTPUT, the count is kept by the node (never the same storage, see forward())."""
    lib = _lib.load()
    loss = torch.empty((), dtype=torch.float32, device=partials.device)
    aux = torch.empty(2, dtype=torch.float32, device=partials.device)
    _lib.check(lib.spx_ce_finish(_lib.ptr(partials), int(partials.shape[0]), _lib.ptr(loss), _lib.ptr(aux), _lib.stream_ptr()))
    return loss, aux


def shifted_labels_i32(labels: torch.Tensor, device) -> torch.Tensor:
    """labels - 1 as contiguous int32 on ``device`` (the kernels' class index: 0 = void becomes -1, loss.py:32) in ONE elementwise
    launch (spx_shift_labels) for int64 / int32 labels already on the device (``.to(int32)`` followed by ``- 1`` are two)."""
    if labels.is_cuda and labels.device == torch.device(device) and labels.dtype in (torch.int64, torch.int32):
        lib = _lib.load()
        labels = labels.contiguous()
        out = torch.empty(labels.shape, dtype=torch.int32, device=labels.device)
        if labels.numel():
            _lib.check(lib.spx_shift_labels(_lib.ptr(labels), int(labels.dtype == torch.int64), labels.numel(), _lib.ptr(out),
                                            _lib.stream_ptr()))
        return out
    return (labels.to(device=device, dtype=torch.int32) - 1).contiguous()


class GroupTables:
    """Device index tables of the dense [U, P] form of the per-class group projections (one per module and device; built by
    ``PPNetMultiScale._group_index``): which block (class present) a unit row / prototype column belongs to and where, and the
    (row, col) of every weight element in block order."""

    def __init__(self, row_block, row_local, col_block, col_local, flat_row, flat_col, block_cols, U, P):
        self.row_block, self.row_local, self.col_block, self.col_local = row_block, row_local, col_block, col_local
        self.flat_row, self.flat_col, self.block_cols, self.U, self.P = flat_row, flat_col, list(block_cols), int(U), int(P)
        self.c_cols = (C.c_int32 * len(self.block_cols))(*self.block_cols)


class _GroupDenseFn(torch.autograd.Function):
    """Wd [U, P] from the group_projection weights in one launch (spx_group_dense); backward: one gather (spx_group_dense_bwd),
    the weights' gradients are views of its flat output (segmentation/model/model_multiscale_group.py:283-303 in dense form)."""

    @staticmethod
    def forward(ctx, tables, *weights):
        lib = _lib.load()
        ws = [w.detach().contiguous().float() for w in weights]
        if len(ws) != len(tables.block_cols) or any(int(w.shape[1]) != n for w, n in zip(ws, tables.block_cols)):
            raise SpxError("group_dense: the weights do not match the index tables")
        ptrs = (C.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
        out = torch.empty((tables.U, tables.P), dtype=torch.float32, device=ws[0].device)
        _lib.check(lib.spx_group_dense(ptrs, tables.c_cols, len(ws), _lib.ptr(tables.row_block), _lib.ptr(tables.row_local),
                                       _lib.ptr(tables.col_block), _lib.ptr(tables.col_local), tables.U, tables.P, _lib.ptr(out),
                                       _lib.stream_ptr()))
        ctx.tables, ctx.shapes = tables, [tuple(w.shape) for w in weights]
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        t = ctx.tables
        g = g.contiguous().float()
        n = int(t.flat_row.numel())
        d_flat = torch.empty(n, dtype=torch.float32, device=g.device)
        _lib.check(lib.spx_group_dense_bwd(_lib.ptr(g), _lib.ptr(t.flat_row), _lib.ptr(t.flat_col), n, t.P, _lib.ptr(d_flat),
                                           _lib.stream_ptr()))
        parts = torch.split(d_flat, [a * b for a, b in ctx.shapes])
        return (None,) + tuple(v.view(sh) for v, sh in zip(parts, ctx.shapes))


def group_dense(tables: "GroupTables", weights) -> torch.Tensor:
    weights = list(weights)
    if not weights or not weights[0].is_cuda:
        raise SpxError("scaleprotoseg_amd runs on an AMD GPU only; there is no CPU fallback")
    return _GroupDenseFn.apply(tables, *weights)


def group_exp(units: torch.Tensor) -> torch.Tensor:
    if not units.is_cuda:
        raise SpxError("scaleprotoseg_amd runs on an AMD GPU only; there is no CPU fallback")
    return _ExpFn.apply(units)


def wide_group_tail(units: torch.Tensor, wg: torch.Tensor) -> torch.Tensor:
    if not units.is_cuda:
        raise SpxError("scaleprotoseg_amd runs on an AMD GPU only; there is no CPU fallback")
    return _WideGroupTailFn.apply(units, wg)


def wide_linear(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    if not a.is_cuda:
        raise SpxError("scaleprotoseg_amd runs on an AMD GPU only; there is no CPU fallback")
    return _WideLinearFn.apply(a, w)


class _Packs:
    """Device buffers holding the MFMA-ordered operands of one forward."""

    def __init__(self, plan: SpxPlan, bank2d: torch.Tensor, head: Optional[torch.Tensor], need_bwd: bool,
                 tail: Optional[torch.Tensor] = None):
        lib = _lib.load()
        dev = bank2d.device
        pp = C.byref(plan)
        u8 = dict(dtype=torch.uint8, device=dev)
        self.bank = torch.empty(lib.spx_packed_bank_bytes(pp), **u8)
        self.bankT = torch.empty(lib.spx_packed_bankT_bytes(pp), **u8) if need_bwd else None
        self.p2 = torch.empty(lib.spx_packed_p2_bytes(pp) // 4, dtype=torch.float32, device=dev)
        self.head = self.headT = self.tail = self.tailT = None
        if head is not None:
            self.head = torch.empty(lib.spx_packed_head_bytes(pp), **u8)
            self.headT = torch.empty(lib.spx_packed_headT_bytes(pp), **u8) if need_bwd else None
        if tail is not None:
            # grouping-head tail: W_g fragments; the backward takes head^T with the unit index in accumulator order
            self.tail = torch.empty(lib.spx_packed_tail_bytes(pp), **u8)
            self.tailT = torch.empty(lib.spx_packed_tail_bytes(pp), **u8) if need_bwd else None
        _lib.check(
            lib.spx_pack_all(
                pp, _lib.ptr(bank2d), _lib.ptr(head), _lib.ptr(tail), int(tail.shape[0]) if tail is not None else 0,
                _lib.ptr(self.bank), _lib.ptr(self.bankT), _lib.ptr(self.p2), _lib.ptr(self.head), _lib.ptr(self.headT),
                _lib.ptr(self.tail), _lib.ptr(self.tailT), _lib.stream_ptr(),
            )
        )


# ---- pack cache: the MFMA-ordered operands depend on the parameters only, so an unchanged bank / head /
# tail (inference, evaluation, several forwards per optimizer step) reuses them instead of re-running the pack kernels and
# their five allocations per forward.  Key: (object, ``_version``, ``data_ptr``) of each parameter + the plan + what was packed.
# ``_version`` follows every in-place edit made THROUGH the parameter (optimizer steps, ``copy_``, ``add_``); a new Parameter
# object (``prune_prototypes``) or new storage (``weight.data = ...``: the simplex projection) changes the key too.  The one
# edit none of them sees is an in-place write through ``.data`` (``p.data.copy_(...)``: ``.data`` has its own version counter):
# the package's own such sites (the push commit, the ``_initialize_weights`` fills) call ``invalidate_pack_cache()``, and so must
# foreign code that edits a parameter that way.
import collections as _collections
import weakref as _weakref

_PACK_CACHE: "_collections.OrderedDict[tuple, tuple]" = _collections.OrderedDict()
_PACK_CACHE_MAX = 8
PACK_CACHE_STATS = {"hits": 0, "misses": 0}


def invalidate_pack_cache() -> None:
    """Drop every cached pack (call after editing a parameter in place through ``.data``)."""
    _PACK_CACHE.clear()


def _tensor_key(t: Optional[torch.Tensor]):
    return None if t is None else (id(t), int(t._version), int(t.data_ptr()), tuple(t.shape), t.dtype)


def _cached_packs(plan_key, plan, bank, head, tail, bank2d, head2d, tail2d, need_bwd):
    if torch.cuda.is_current_stream_capturing():
        # a captured step is replayed after its parameters have been edited in place (the static-buffer protocol of HIP graphs):
        # the pack kernels must be part of the graph
        return _Packs(plan, bank2d, head2d, need_bwd, tail2d)
    key = (plan_key, _tensor_key(bank), _tensor_key(head), _tensor_key(tail), bool(need_bwd))
    hit = _PACK_CACHE.get(key)
    if hit is not None:
        refs, packs = hit
        if all((r is None and t is None) or (r is not None and r() is t) for r, t in zip(refs, (bank, head, tail))):
            _PACK_CACHE.move_to_end(key)
            PACK_CACHE_STATS["hits"] += 1
            return packs
        del _PACK_CACHE[key]
    PACK_CACHE_STATS["misses"] += 1
    packs = _Packs(plan, bank2d, head2d, need_bwd, tail2d)
    refs = tuple(None if t is None else _weakref.ref(t) for t in (bank, head, tail))
    _PACK_CACHE[key] = (refs, packs)
    while len(_PACK_CACHE) > _PACK_CACHE_MAX:
        _PACK_CACHE.popitem(last=False)
    return packs


class _ProtoHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bank, head, layout, want_dist, want_act, epsilon, act_fn, gather=None, tail=None, ce_labels=None):
        lib = _lib.load()
        ctx.set_materialize_grads(False)      # an output nobody differentiates arrives as None, not as a zero tensor
        B, HW = _check_x(x, layout)
        P, K = layout.num_prototypes, layout.num_classes
        xd = _x_dtype_code(x)
        x = x.contiguous()
        bank2d = bank.detach().reshape(P, layout.channels_per_scale).contiguous().float()
        head2d = head.detach().contiguous().float() if head is not None else None
        if head2d is not None and tuple(head2d.shape) != (K, P):
            raise SpxError(f"head matrix must be [{K}, {P}], got {tuple(head2d.shape)}")
        plan = layout.plan()
        need_bwd = any(t is not None and t.requires_grad for t in (x, bank, head, tail))
        if need_bwd and bank.requires_grad and layout.channels_per_scale > 256:
            # the parameter-side backward tiles at most 8 channel blocks: say so BEFORE a full forward has run
            raise SpxError(f"prototype gradients need channels_per_scale <= 256, got {layout.channels_per_scale} "
                           "(freeze the bank or run the forward under torch.no_grad())")
        tail2d = tail.detach().contiguous().float() if tail is not None else None
        if tail2d is not None:
            if head2d is None or gather is not None:
                raise SpxError("the fused group tail needs the dense head and no class gather")
            if tail2d.dim() != 2 or tail2d.shape[1] != K or tail2d.shape[0] > 32:
                raise SpxError(f"group tail must be [K2 <= 32, {K}], got {tuple(tail2d.shape)}")
        plan_key = (layout.num_prototypes, layout.num_classes, layout.num_scales, layout.channels_per_scale,
                    tuple(tuple(int(v) for v in r_) for r_ in layout.scale_ranges))
        packs = _cached_packs(plan_key, plan, bank, head, tail, bank2d, head2d, tail2d, need_bwd)
        f32 = dict(dtype=torch.float32, device=x.device)
        act = torch.empty((B * HW, P), **f32) if want_act else None
        logits = torch.empty((B * HW, K), **f32) if head is not None else None
        gact = None
        ce = ce_state = None
        if ce_labels is not None:
            if head is None:
                raise SpxError("the fused cross entropy needs a head (logits)")
            if tuple(ce_labels.shape) != (B, HW) or ce_labels.dtype != torch.int32 or not ce_labels.is_cuda:
                raise SpxError(f"ce labels must be int32 [{B}, {HW}] on the GPU (class 0..K-1, anything else = ignored)")
            ce_labels = ce_labels.contiguous()
            lse = torch.empty((B * HW,), **f32)
            pred = torch.empty((B * HW,), dtype=torch.int32, device=x.device)
            # the grouping tail's kernel is pixel-per-thread (flat partial count); the plain head's epilogue is per tile
            n_part = lib.spx_ce_partials_flat(B * HW) if tail2d is not None else lib.spx_ce_partials(B, HW)
            partials = torch.empty((n_part, 2), **f32)
            ce = _lib.SpxCe(labels=_lib.ptr(ce_labels), lse=_lib.ptr(lse), pred=_lib.ptr(pred), partials=_lib.ptr(partials))
            ce_state = (ce_labels, lse, pred, partials)
        if tail2d is not None:
            K2 = int(tail2d.shape[0])
            logits = torch.empty((B * HW, K2), **f32)
            gact = torch.empty((B * HW, K), **f32)
            dist = torch.empty((B, P) + tuple(x.shape[2:]), **f32) if want_dist else None
            # the tail as its own (fp32) kernel when the cross entropy rides on the logits; otherwise fused into the
            # distance kernel's epilogue
            tail_ws = ce is not None
            with _timed("spx_dist_fwd"):
                if tail_ws:
                    ws = torch.empty(lib.spx_group_tail_workspace_bytes(C.byref(plan), B, HW), dtype=torch.uint8, device=x.device)
                    _lib.check(
                        lib.spx_dist_fwd_group_ws(
                            C.byref(plan), _lib.ptr(x), xd, B, HW, _lib.ptr(packs.bank), _lib.ptr(packs.p2),
                            _lib.ptr(packs.head), _lib.ptr(tail2d), K2, _lib.ptr(dist), _lib.ptr(act), _lib.ptr(gact),
                            _lib.ptr(logits), C.byref(ce) if ce is not None else None, _lib.ptr(ws), float(epsilon),
                            ACT_FN[act_fn], _lib.stream_ptr(),
                        )
                    )
                else:
                    _lib.check(
                        lib.spx_dist_fwd_group(
                            C.byref(plan), _lib.ptr(x), xd, B, HW, _lib.ptr(packs.bank), _lib.ptr(packs.p2),
                            _lib.ptr(packs.head), _lib.ptr(packs.tail), K2, _lib.ptr(dist), _lib.ptr(act),
                            _lib.ptr(gact), _lib.ptr(logits), float(epsilon), ACT_FN[act_fn], _lib.stream_ptr(),
                        )
                    )
        else:
            if gather is not None:
                if tuple(gather.labels.shape) != (B, HW) or gather.labels.dtype != torch.int32:
                    raise SpxError(f"gather labels must be int32 [{B}, {HW}]")
                # slots no prototype maps to (and pixels without a class) are written as 0 by the kernel itself (class ids
                # below 1024: include/spx_hip.h); beyond that the planes are zero-filled here
                few = gather.table is not None and int(gather.table.shape[0]) <= 1024
                dist = (torch.empty if few else torch.zeros)((B, gather.width, HW), **f32)
                g_args = (_lib.ptr(gather.labels), _lib.ptr(gather.keys), gather.width, _lib.ptr(dist), None)
            else:
                dist = torch.empty((B, P) + tuple(x.shape[2:]), **f32) if want_dist else None
                g_args = (None, None, 0, None, _lib.ptr(dist))
            # small pixel grids run one workgroup per (tile, scale) (include/spx_hip.h, "Scale-parallel forward"); the
            # logits then need a workspace for the per-scale partials, and the cross entropy (which wants the SUMMED
            # logits) runs as the stand-alone kernel right behind
            ws_bytes = lib.spx_fwd_split_workspace_bytes(C.byref(plan), B, HW) if head is not None else 0
            with _timed("spx_dist_fwd"):
                if ce is not None and ws_bytes == 0:
                    _lib.check(
                        lib.spx_dist_fwd_ce(
                            C.byref(plan), _lib.ptr(x), xd, B, HW, _lib.ptr(packs.bank), _lib.ptr(packs.p2),
                            _lib.ptr(packs.head), *g_args, _lib.ptr(act), _lib.ptr(logits), C.byref(ce), float(epsilon),
                            ACT_FN[act_fn], _lib.stream_ptr(),
                        )
                    )
                else:
                    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
                    _lib.check(
                        lib.spx_dist_fwd_ws(
                            C.byref(plan), _lib.ptr(x), xd, B, HW, _lib.ptr(packs.bank), _lib.ptr(packs.p2),
                            _lib.ptr(packs.head), *g_args, _lib.ptr(act), _lib.ptr(logits), _lib.ptr(ws), float(epsilon),
                            ACT_FN[act_fn], _lib.stream_ptr(),
                        )
                    )
                    if ce is not None:
                        ce_labels_, lse_, pred_, _ = ce_state
                        partials = torch.empty((lib.spx_ce_partials_flat(B * HW), 2), **f32)
                        _lib.check(lib.spx_ce_fwd(_lib.ptr(logits), _lib.ptr(ce_labels_), B * HW, K, _lib.ptr(lse_),
                                                  _lib.ptr(pred_), _lib.ptr(partials), _lib.stream_ptr()))
                        ce_state = (ce_labels_, lse_, pred_, partials)
        ctx.gather = gather
        ctx.tail2d = tail2d
        ctx.layout, ctx.plan, ctx.packs = layout, plan, packs
        ctx.epsilon, ctx.act_fn = float(epsilon), act_fn
        ctx.have = (logits is not None, dist is not None, act is not None)
        # (the logits are an OUTPUT: kept through save_for_backward, never as a ctx attribute - that would be a reference
        # cycle node -> ctx -> tensor -> node that outlives the backward and keeps the leaves' AccumulateGrad nodes alive)
        # (gact too: it is a differentiable OUTPUT since round 4 - compute_group's list - and as a ctx attribute it would close the
        # same cycle)
        ctx.save_for_backward(x, bank2d, head2d, logits if ce_state is not None else None, gact)
        ctx.bank_shape = tuple(bank.shape)
        outs = tuple(t if t is not None else x.new_empty(0) for t in (logits, dist, act))
        extra = gact if tail2d is not None else x.new_empty(0)     # exp(units): compute_group's list; a gradient on it enters the backward's dUnits
        ce_loss, ce_pred = x.new_empty(0), x.new_empty(0)
        ctx.ce_state = ctx.ce_count = None
        if ce_state is not None:
            ce_loss, tot = _ce_finish(ce_state[3])                # mean (0 / 0 = nan when every pixel is ignored, as torch's), (count, sum)
            ce_pred = ce_state[2]
            ctx.ce_state, ctx.ce_count = ce_state, tot[0]
        ctx.mark_non_differentiable(*([o for o, h in zip(outs, ctx.have) if not h] + ([extra] if tail2d is None else []) + [ce_pred]
                                      + ([ce_loss] if ce_state is None else [])))
        return outs + (extra, ce_loss, ce_pred)

    @staticmethod
    def backward(ctx, g_logits, g_dist, g_act, g_gact=None, g_ce=None, _g_pred=None):
        lib = _lib.load()
        x, bank2d, head2d, ce_logits, gact = ctx.saved_tensors
        layout, plan, packs = ctx.layout, ctx.plan, ctx.packs
        B, HW = _check_x(x, layout)
        P, K, Cs = layout.num_prototypes, layout.num_classes, layout.channels_per_scale
        need_x, need_bank, need_head = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        have_l, have_d, have_a = ctx.have
        gl = g_logits.contiguous().float() if (have_l and g_logits is not None) else None
        gd = g_dist.contiguous().float() if (have_d and g_dist is not None) else None
        ga = g_act.contiguous().float() if (have_a and g_act is not None) else None
        if packs.bankT is None:
            raise SpxError("backward requested but the forward ran without gradient packs")
        pp = C.byref(plan)
        dev = x.device
        xd = _x_dtype_code(x)
        scr = lib.spx_bwd_scratch_bytes(pp, B, HW)
        with_ce = ctx.ce_state is not None and g_ce is not None
        need_head = need_head and (gl is not None or with_ce or (ctx.tail2d is not None and g_gact is not None))
        s = _lib.stream_ptr()
        ce = d_logits_ce = None
        if with_ce:
            ce_labels, lse, _, _ = ctx.ce_state
            coef = (g_ce.float() / ctx.ce_count).reshape(1).contiguous()       # on the device: no host read-back
            if gl is not None:
                # the logits ALSO carry a gradient of their own: form the cross entropy's part with the stand-alone
                # kernel, add, and take the ordinary d_logits path
                dl = torch.empty_like(ce_logits)
                _lib.check(lib.spx_ce_bwd(_lib.ptr(ce_logits), _lib.ptr(lse), _lib.ptr(ce_labels), _lib.ptr(coef),
                                          B * HW, int(ce_logits.shape[1]), _lib.ptr(dl), s))
                gl = gl.reshape(dl.shape) + dl
            else:
                d_logits_ce = torch.empty_like(ce_logits) if (need_head or (ctx.tail2d is not None and ctx.needs_input_grad[9])) else None
                ce = _lib.SpxCe(labels=_lib.ptr(ce_labels), lse=_lib.ptr(lse), logits=_lib.ptr(ce_logits),
                                coef=_lib.ptr(coef), d_logits_out=_lib.ptr(d_logits_ce))
        dx = torch.empty_like(x) if need_x else None
        # bf16 features and a scale of more than 192 prototypes (several panels): the panels' shares of dX are summed in fp32
        n_acc = lib.spx_bwd_dx_scratch_bytes(pp, xd, B, HW) if need_x else 0
        dx_acc = torch.empty(n_acc, dtype=torch.uint8, device=dev) if n_acc else None
        g_scr = torch.empty(scr, dtype=torch.uint8, device=dev) if need_bank else None
        a_scr = torch.empty(lib.spx_bwd_head_scratch_bytes(pp, B, HW), dtype=torch.uint8, device=dev) if need_head else None
        tail2d, d_units, d_tail = ctx.tail2d, None, None
        gg = g_gact.contiguous().float() if (tail2d is not None and g_gact is not None) else None
        if tail2d is not None and gl is None and ce is None:
            if gg is None:
                raise SpxError("backward through the fused group tail without a logits gradient")
            gl = torch.zeros((B * HW, int(tail2d.shape[0])), dtype=torch.float32, device=dev)   # only the group activations carry one
        d_bank = d_head = None
        with _timed("spx_dist_bwd"):
            if tail2d is not None and ce is not None:
                d_units = torch.empty((B * HW, K), dtype=torch.float32, device=dev)
                _lib.check(
                    lib.spx_dist_bwd_group_ce(
                        pp, _lib.ptr(x), xd, B, HW, _lib.ptr(packs.bank), _lib.ptr(packs.bankT), _lib.ptr(packs.p2),
                        _lib.ptr(packs.headT), _lib.ptr(packs.tailT), int(tail2d.shape[0]), _lib.ptr(gact),
                        _lib.ptr(gd), _lib.ptr(ga), C.byref(ce), _lib.ptr(gg), _lib.ptr(d_units), _lib.ptr(dx), _lib.ptr(dx_acc), _lib.ptr(g_scr),
                        _lib.ptr(a_scr), ctx.epsilon, ACT_FN[ctx.act_fn], s,
                    )
                )
            elif tail2d is not None:
                d_units = torch.empty((B * HW, K), dtype=torch.float32, device=dev)
                _lib.check(
                    lib.spx_dist_bwd_group(
                        pp, _lib.ptr(x), xd, B, HW, _lib.ptr(packs.bank), _lib.ptr(packs.bankT), _lib.ptr(packs.p2),
                        _lib.ptr(packs.headT), _lib.ptr(packs.tailT), int(tail2d.shape[0]), _lib.ptr(gact),
                        _lib.ptr(gd), _lib.ptr(ga), _lib.ptr(gl), _lib.ptr(gg), _lib.ptr(d_units), _lib.ptr(dx), _lib.ptr(dx_acc), _lib.ptr(g_scr),
                        _lib.ptr(a_scr), ctx.epsilon, ACT_FN[ctx.act_fn], s,
                    )
                )
            elif ce is not None:
                g = ctx.gather
                _lib.check(
                    lib.spx_dist_bwd_ce(
                        pp, _lib.ptr(x), xd, B, HW, _lib.ptr(packs.bank), _lib.ptr(packs.bankT), _lib.ptr(packs.p2),
                        _lib.ptr(packs.headT), _lib.ptr(g.labels) if g is not None else None,
                        _lib.ptr(g.keys) if g is not None else None, g.width if g is not None else 0,
                        _lib.ptr(gd) if g is None else None, _lib.ptr(gd) if g is not None else None, _lib.ptr(ga),
                        C.byref(ce), _lib.ptr(dx), _lib.ptr(dx_acc), _lib.ptr(g_scr), _lib.ptr(a_scr), ctx.epsilon, ACT_FN[ctx.act_fn], s,
                    )
                )
            elif ctx.gather is not None:
                g = ctx.gather
                _lib.check(
                    lib.spx_dist_bwd_cls(
                        pp, _lib.ptr(x), xd, B, HW, _lib.ptr(packs.bank), _lib.ptr(packs.bankT), _lib.ptr(packs.p2),
                        _lib.ptr(packs.headT) if gl is not None else None, _lib.ptr(g.labels), _lib.ptr(g.keys),
                        g.width, _lib.ptr(gd), _lib.ptr(ga), _lib.ptr(gl), _lib.ptr(dx), _lib.ptr(dx_acc), _lib.ptr(g_scr),
                        _lib.ptr(a_scr), ctx.epsilon, ACT_FN[ctx.act_fn], s,
                    )
                )
            else:
                _lib.check(
                    lib.spx_dist_bwd(
                        pp, _lib.ptr(x), xd, B, HW, _lib.ptr(packs.bank), _lib.ptr(packs.bankT), _lib.ptr(packs.p2),
                        _lib.ptr(packs.headT) if gl is not None else None, _lib.ptr(gd), _lib.ptr(ga), _lib.ptr(gl),
                        _lib.ptr(dx), _lib.ptr(dx_acc), _lib.ptr(g_scr), _lib.ptr(a_scr), ctx.epsilon, ACT_FN[ctx.act_fn], s,
                    )
                )
        if ce is not None:
            gl = d_logits_ce                          # formed by the pixel kernel's prologue
        if tail2d is not None:
            if ctx.needs_input_grad[9]:
                d_tail = _pixel_outer(gl, gact)   # d W_g [K2, U]: one small product over the pixels
            gl = d_units                              # the parameter kernel's d_logits operand
        if need_bank or need_head:
            ws = torch.empty(lib.spx_bank_bwd_workspace_bytes(pp, B, HW) // 4, dtype=torch.float32, device=dev)
            d_bank = torch.empty((P, Cs), dtype=torch.float32, device=dev) if need_bank else None
            d_head = torch.empty((K, P), dtype=torch.float32, device=dev) if need_head else None
            with _timed("spx_bank_bwd"):
                _lib.check(
                    lib.spx_bank_bwd(
                        pp, _lib.ptr(x), xd, B, HW, _lib.ptr(bank2d), _lib.ptr(g_scr), _lib.ptr(a_scr),
                        _lib.ptr(gl), _lib.ptr(d_bank), _lib.ptr(d_head), _lib.ptr(ws), s,
                    )
                )
            if d_bank is not None:
                d_bank = d_bank.reshape(ctx.bank_shape)
        elif ctx.needs_input_grad[2] and head2d is not None:
            d_head = torch.zeros_like(head2d)
        if ctx.needs_input_grad[2] and d_head is None and head2d is not None:
            d_head = torch.zeros_like(head2d)
        if tail2d is not None and ctx.needs_input_grad[9] and d_tail is None:
            d_tail = torch.zeros_like(tail2d)
        return dx, d_bank, d_head, None, None, None, None, None, None, d_tail, None


def proto_head_forward(
    x: torch.Tensor,
    bank: torch.Tensor,
    head: Optional[torch.Tensor],
    layout: BankLayout,
    *,
    want_distances: bool = True,
    want_activations: bool = False,
    epsilon: float = 1e-4,
    activation: str = "log",
    class_gather: Optional[ClassGather] = None,
    group_tail: Optional[torch.Tensor] = None,
    ce_labels: Optional[torch.Tensor] = None,
):
    """(logits [B*H*W, K] | None, distances [B,P,H,W] | None, activations [B*H*W, P] | None).

    ``x``: the reference's ``conv_features`` (a Sigmoid's output, values in (0, 1)); any bf16 / fp32 values are accepted, but
    the prototype gradient takes features beyond +-65504 as saturated (its product runs in fp16; include/spx_hip.h).

    With ``class_gather`` the distance output is the class-gathered tensor [B, J, H*W] (slot planes) instead of the P-wide map
    (the only entries the reference's KLDLoss reads, segmentation/model/loss.py:89-107)."""
    if activation not in ACT_FN:
        raise SpxError(f"activation {activation!r} has no fused kernel (use 'log' or 'linear')")
    if not x.is_cuda:
        raise SpxError("scaleprotoseg_amd runs on an AMD GPU only; there is no CPU fallback")
    logits, dist, act, gact, ce_loss, ce_pred = _ProtoHeadFn.apply(
        x, bank, head, layout, want_distances or class_gather is not None, want_activations, epsilon, activation,
        class_gather, group_tail, ce_labels,
    )
    if ce_labels is not None:
        # fused cross entropy: a 4th entry (loss, argmax prediction per pixel); the loss is
        # differentiable through the same backward as the logits
        out = (logits, dist if (want_distances or class_gather is not None) else None,
               act if want_activations else None, FusedCrossEntropy(ce_loss, ce_pred, ce_labels))
        return out + (gact,) if group_tail is not None else out
    if group_tail is not None:
        # fused grouping head (model_multiscale_group.py:283-308): logits = exp(act . head^T) . group_tail^T;
        # the last entry is exp(act . head^T) [B*H*W, U] (cat of compute_group's list); a gradient on it joins dUnits
        return logits, (dist if want_distances else None), (act if want_activations else None), gact
    return (
        logits if head is not None else None,
        dist if (want_distances or class_gather is not None) else None,
        act if want_activations else None,
    )


def push_masked_argmin(
    distances: torch.Tensor,
    labels: torch.Tensor,
    class_identity: torch.Tensor,
    *,
    void_class: Optional[int] = 0,
    max_dist: float = 1e10,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """(indices int64 [B,P], values f32 [B,P]); push_multiscale_optimization.py:74-91 on the GPU."""
    lib = _lib.load()
    if distances.dim() != 4:
        raise SpxError("distances must be [B, P, H, W]")
    B, P, H, W = distances.shape
    K = class_identity.shape[1]
    if tuple(labels.shape) != (B, H, W):
        raise SpxError(f"labels must be [{B}, {H}, {W}], got {tuple(labels.shape)}")
    d = distances.detach().contiguous().float()
    lab = labels.to(device=d.device, dtype=torch.int32).contiguous()
    ident = class_identity.to(device=d.device, dtype=torch.float32).contiguous()
    idx = torch.empty((B, P), dtype=torch.int64, device=d.device)
    val = torch.empty((B, P), dtype=torch.float32, device=d.device)
    scratch = torch.empty((B * P,), dtype=torch.int64, device=d.device)
    _lib.check(
        lib.spx_push_argmin(
            _lib.ptr(d), _lib.ptr(lab), _lib.ptr(ident), B, P, K, H * W, -1 if void_class is None else int(void_class),
            float(max_dist), _lib.ptr(idx), _lib.ptr(val), _lib.ptr(scratch), _lib.stream_ptr(),
        )
    )
    return idx, val


def identity_is_one_hot(class_identity: torch.Tensor) -> bool:
    """True when every row of prototype_class_identity is one-hot or all zero (what the reference builds,
    model_multiscale.py:89-97): the fused push compares class indices instead of multiplying by the mask."""
    ident = class_identity.detach()
    return bool((((ident == 0) | (ident == 1)).all() & (ident.sum(dim=1) <= 1).all()).item())


def push_min_from_features(
    conv_features: torch.Tensor,
    bank: torch.Tensor,
    layout: BankLayout,
    labels: torch.Tensor,
    class_identity: torch.Tensor,
    *,
    void_class: Optional[int] = 0,
    max_dist: float = 1e10,
    keys: Optional[torch.Tensor] = None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """(indices int64 [B,P], values f32 [B,P]) of the class-masked per-prototype minimum over each image's latent grid
    (push_multiscale_optimization.py:68-91), computed INSIDE the distance kernel from the features: the [B, P, H, W] map
    is never written (spx_dist_push_min).  Bit-identical to ``push_masked_argmin`` on the map ``proto_head_forward`` writes.
    ``labels`` [B, H, W] in the push's convention (``void_class`` as in ``push_masked_argmin``); ``class_identity`` must be
    one-hot per row (``identity_is_one_hot``); ``keys``: a cached ``class_gather_table(layout, class_identity, device)[0]``."""
    lib = _lib.load()
    if not conv_features.is_cuda:
        raise SpxError("scaleprotoseg_amd runs on an AMD GPU only; there is no CPU fallback")
    B, HW = _check_x(conv_features, layout)
    H, W = conv_features.shape[2], conv_features.shape[3]
    if tuple(labels.shape) != (B, H, W):
        raise SpxError(f"labels must be [{B}, {H}, {W}], got {tuple(labels.shape)}")
    if keys is None:
        if not identity_is_one_hot(class_identity):
            raise SpxError("the fused push needs a one-hot prototype_class_identity (use push_masked_argmin on the distance map)")
        keys = class_gather_table(layout, class_identity, conv_features.device)[0]
    P, K = layout.num_prototypes, int(class_identity.shape[1])
    x = conv_features.detach().contiguous()
    dev = x.device
    lab = labels.to(device=dev, dtype=torch.int32).reshape(B, HW).contiguous()
    bank2d = bank.detach().reshape(P, layout.channels_per_scale).contiguous().float()
    plan = layout.plan()
    plan_key = (layout.num_prototypes, layout.num_classes, layout.num_scales, layout.channels_per_scale,
                tuple(tuple(int(v) for v in r_) for r_ in layout.scale_ranges))
    packs = _cached_packs(plan_key, plan, bank, None, None, bank2d, None, None, False)
    idx = torch.empty((B, P), dtype=torch.int64, device=dev)
    val = torch.empty((B, P), dtype=torch.float32, device=dev)
    scratch = torch.empty((B * P,), dtype=torch.int64, device=dev)
    _lib.check(lib.spx_dist_push_min(C.byref(plan), _lib.ptr(x), _x_dtype_code(x), B, HW, _lib.ptr(packs.bank), _lib.ptr(packs.p2),
                                     _lib.ptr(lab), -1 if void_class is None else int(void_class), K, _lib.ptr(keys), float(max_dist),
                                     _lib.ptr(idx), _lib.ptr(val),
                                     _lib.ptr(scratch), _lib.stream_ptr()))
    return idx, val


def argmin_over_images(values: torch.Tensor) -> torch.Tensor:
    """tot_dist.argmin(dim=0) with lowest-image tie-break; push_multiscale_optimization.py:135-137."""
    lib = _lib.load()
    v = values.detach().contiguous().float()
    N, P = v.shape
    best = torch.empty((P,), dtype=torch.int64, device=v.device)
    _lib.check(lib.spx_argmin_images(_lib.ptr(v), N, P, _lib.ptr(best), _lib.stream_ptr()))
    return best


def upsample_argext(src: torch.Tensor, size: Tuple[int, int], largest: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """(indices int64 [N,H,W], values f32 [N,H,W]) of ``F.interpolate(src, size, mode="bilinear",
    align_corners=False)`` reduced with argmin (``largest=False``) / argmax over the channels, fused: the upsampled
    [N,C,H,W] tensor is never materialised (segmentation/eval_valid_multiscale.py:229-234, :375-383)."""
    lib = _lib.load()
    if src.dim() != 4:
        raise SpxError("src must be [N, C, h, w]")
    s = src.detach().contiguous().float()
    N, Cc, h, w = s.shape
    H, W = int(size[0]), int(size[1])
    idx = torch.empty((N, H, W), dtype=torch.int64, device=s.device)
    val = torch.empty((N, H, W), dtype=torch.float32, device=s.device)
    _lib.check(lib.spx_upsample_argext(_lib.ptr(s), N, Cc, h, w, H, W, 1 if largest else 0, _lib.ptr(idx), _lib.ptr(val), _lib.stream_ptr()))
    return idx, val
