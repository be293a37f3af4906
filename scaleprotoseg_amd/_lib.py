"""ctypes binding of libspx_hip.so (include/spx_hip.h).

The HIP extension is the only compute path of this package: loading fails
loudly when the shared library is missing, and every operator raises when its
tensors are not on an AMD GPU.  There is no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspx_hip.so")
SPX_MAX_PANELS = 64
ABI_VERSION = 15


class SpxError(RuntimeError):
    """A libspx_hip.so entry point returned a non-zero status."""


class SpxPlan(C.Structure):
    _fields_ = [
        ("num_prototypes", C.c_int32),
        ("num_classes", C.c_int32),
        ("num_scales", C.c_int32),
        ("channels_per_scale", C.c_int32),
        ("kc", C.c_int32),
        ("npb", C.c_int32),
        ("ncb", C.c_int32),
        ("npanels", C.c_int32),
        ("panel_ch0", C.c_int32 * SPX_MAX_PANELS),
        ("panel_p0", C.c_int32 * SPX_MAX_PANELS),
        ("panel_np", C.c_int32 * SPX_MAX_PANELS),
    ]


class SpxCe(C.Structure):
    """spx_ce of include/spx_hip.h: the cross-entropy attachment of spx_dist_fwd_ce / spx_dist_bwd_ce."""

    _fields_ = [
        ("labels", C.c_void_p),
        ("lse", C.c_void_p),
        ("pred", C.c_void_p),
        ("partials", C.c_void_p),
        ("logits", C.c_void_p),
        ("coef", C.c_void_p),
        ("d_logits_out", C.c_void_p),
    ]


_PP = C.POINTER(SpxPlan)
_PCE = C.POINTER(SpxCe)
_V = C.c_void_p
_I = C.c_int32
_F = C.c_float

# name -> (restype, argtypes); must list every symbol include/spx_hip.h declares
SIGNATURES = {
    "spx_version": (C.c_int, []),
    "spx_last_error": (C.c_char_p, []),
    "spx_make_plan": (C.c_int, [_I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), _PP]),
    "spx_packed_bank_bytes": (C.c_size_t, [_PP]),
    "spx_packed_bankT_bytes": (C.c_size_t, [_PP]),
    "spx_packed_p2_bytes": (C.c_size_t, [_PP]),
    "spx_packed_head_bytes": (C.c_size_t, [_PP]),
    "spx_packed_headT_bytes": (C.c_size_t, [_PP]),
    "spx_pack_bank": (C.c_int, [_PP, _V, _V, _V, _V, _V]),
    "spx_pack_head": (C.c_int, [_PP, _V, _V, _V, _V]),
    "spx_pack_all": (C.c_int, [_PP, _V, _V, _V, _I, _V, _V, _V, _V, _V, _V, _V, _V]),
    "spx_dist_fwd": (C.c_int, [_PP, _V, _I, _I, _I, _V, _V, _V, _V, _V, _V, _F, _I, _V]),
    "spx_dist_bwd": (C.c_int, [_PP, _V, _I, _I, _I, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V, _F, _I, _V]),
    "spx_dist_fwd_cls": (C.c_int, [_PP, _V, _I, _I, _I, _V, _V, _V, _V, _V, _I, _V, _V, _V, _F, _I, _V]),
    "spx_dist_bwd_cls": (C.c_int, [_PP, _V, _I, _I, _I, _V, _V, _V, _V, _V, _V, _I, _V, _V, _V, _V, _V, _V, _V, _F, _I, _V]),
    "spx_group_dense": (C.c_int, [_V, _V, _I, _V, _V, _V, _V, _I, _I, _V, _V]),
    "spx_group_dense_bwd": (C.c_int, [_V, _V, _V, C.c_int64, _I, _V, _V]),
    "spx_packed_tail_bytes": (C.c_size_t, [_PP]),
    "spx_pack_group_tail": (C.c_int, [_PP, _V, _I, _V, _V, _V]),
    "spx_pack_headT_units": (C.c_int, [_PP, _V, _V, _V]),
    "spx_dist_fwd_group": (C.c_int, [_PP, _V, _I, _I, _I, _V, _V, _V, _V, _I, _V, _V, _V, _V, _F, _I, _V]),
    "spx_dist_bwd_group": (C.c_int, [_PP, _V, _I, _I, _I, _V, _V, _V, _V, _V, _I, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V, _F, _I, _V]),
    "spx_bwd_scratch_bytes": (C.c_size_t, [_PP, _I, _I]),
    "spx_bwd_head_scratch_bytes": (C.c_size_t, [_PP, _I, _I]),
    "spx_bwd_dx_scratch_bytes": (C.c_size_t, [_PP, _I, _I, _I]),
    "spx_bank_bwd_workspace_bytes": (C.c_size_t, [_PP, _I, _I]),
    "spx_bank_bwd": (C.c_int, [_PP, _V, _I, _I, _I, _V, _V, _V, _V, _V, _V, _V, _V]),
    "spx_push_argmin": (C.c_int, [_V, _V, _V, _I, _I, _I, _I, _I, _F, _V, _V, _V, _V]),
    "spx_dist_push_min": (C.c_int, [_PP, _V, _I, _I, _I, _V, _V, _V, _I, _I, _V, _F, _V, _V, _V, _V]),
    "spx_argmin_images": (C.c_int, [_V, _I, _I, _V, _V]),
    "spx_kld_segment_max": (C.c_int, [_V, _V, _I, _I, _I, _I, _I, _V, _V, _V, _V]),
    "spx_kld_segment_sumexp": (C.c_int, [_V, _V, _I, _I, _I, _I, _I, _V, _V, _V]),
    "spx_kld_segment_lse": (C.c_int, [_V, _V, _I, _V, _V, _I, _V, _V]),
    "spx_kld_gram_loss": (C.c_int, [_V, _V, _V, _V, _I, _I, _I, _V, _V, _V, _V, _V]),
    "spx_kld_pair_sums": (C.c_int, [_V, _V, _I, _I, _I, _I, _I, _V, _V, _V, _V]),
    "spx_kld_backward": (C.c_int, [_V, _V, _I, _I, _I, _I, _V, _V, _V, _V, _V, _V]),
    "spx_upsample_argext": (C.c_int, [_V, _I, _I, _I, _I, _I, _I, _I, _V, _V, _V]),
    "spx_fwd_split_groups": (C.c_int32, [_PP, _I, _I]),
    "spx_fwd_split_workspace_bytes": (C.c_size_t, [_PP, _I, _I]),
    "spx_dist_fwd_ws": (C.c_int, [_PP, _V, _I, _I, _I, _V, _V, _V, _V, _V, _I, _V, _V, _V, _V, _V, _F, _I, _V]),
    "spx_group_tail_workspace_bytes": (C.c_size_t, [_PP, _I, _I]),
    "spx_dist_fwd_group_ws": (C.c_int, [_PP, _V, _I, _I, _I, _V, _V, _V, _V, _I, _V, _V, _V, _V, _PCE, _V, _F, _I, _V]),
    "spx_dist_bwd_group_ce": (C.c_int, [_PP, _V, _I, _I, _I, _V, _V, _V, _V, _V, _I, _V, _V, _V, _PCE, _V, _V, _V, _V, _V, _V, _F, _I, _V]),
    "spx_exp": (C.c_int, [_V, _V, C.c_int64, _V]),
    "spx_exp_bwd": (C.c_int, [_V, _V, _V, C.c_int64, _V]),
    "spx_pixel_outer_workspace_bytes": (C.c_size_t, [C.c_int64, _I, _I]),
    "spx_pixel_outer": (C.c_int, [_V, _V, C.c_int64, _I, _I, _V, _V, _V]),
    "spx_rows_gemm_workspace_bytes": (C.c_size_t, [_I, _I, _I, _I]),
    "spx_rows_gemm": (C.c_int, [_V, C.c_int64, C.c_int64, _V, C.c_int64, C.c_int64, _V, C.c_int64, _I, _I, _I, _I, _V, C.c_int64, _V, _V]),
    "spx_ce_partials": (C.c_size_t, [_I, _I]),
    "spx_ce_partials_flat": (C.c_size_t, [C.c_int64]),
    "spx_ce_finish": (C.c_int, [_V, C.c_int64, _V, _V, _V]),
    "spx_shift_labels": (C.c_int, [_V, _I, C.c_int64, _V, _V]),
    "spx_dist_fwd_ce": (C.c_int, [_PP, _V, _I, _I, _I, _V, _V, _V, _V, _V, _I, _V, _V, _V, _V, _PCE, _F, _I, _V]),
    "spx_dist_bwd_ce": (C.c_int, [_PP, _V, _I, _I, _I, _V, _V, _V, _V, _V, _V, _I, _V, _V, _V, _PCE, _V, _V, _V, _V, _F, _I, _V]),
    "spx_ce_fwd": (C.c_int, [_V, _V, C.c_int64, _I, _V, _V, _V, _V]),
    "spx_ce_bwd": (C.c_int, [_V, _V, _V, _V, C.c_int64, _I, _V, _V]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libspx_hip.so (built by scaleprotoseg_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SpxError(
            f"{LIB_PATH} is missing: build it with `python -m scaleprotoseg_amd.build` "
            "(hipcc --offload-arch=gfx950).  scaleprotoseg_amd has no CPU fallback."
        )
    # development A/B only (tools/ab_variants.sh): another build of the SAME library; never a different backend
    lib = C.CDLL(os.environ.get("SPX_LIB_OVERRIDE") or LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.spx_version() != ABI_VERSION:
        raise SpxError(f"libspx_hip.so ABI {lib.spx_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        raise SpxError(load().spx_last_error().decode("utf-8", "replace"))


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Raw device pointer of a dense GPU tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SpxError(
            "scaleprotoseg_amd operators run on an AMD GPU only (tensor is on "
            f"{t.device}); there is no CPU fallback"
        )
    if not t.is_contiguous():
        raise SpxError("tensor must be contiguous")
    return t.data_ptr()


def stream_ptr() -> int:
    """hipStream_t of torch's current stream (the raw getter: ``torch.cuda.current_stream()`` builds a Stream object,
    ~20 us per call, which is visible at the launch-bound training shapes)."""
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is None:
        return torch.cuda.current_stream().cuda_stream
    return raw(torch.cuda.current_device())


def make_plan(P: int, K: int, S: int, Cs: int, scale_lo, scale_hi) -> SpxPlan:
    lib = load()
    plan = SpxPlan()
    lo = (C.c_int32 * S)(*[int(v) for v in scale_lo])
    hi = (C.c_int32 * S)(*[int(v) for v in scale_hi])
    check(lib.spx_make_plan(P, K, S, Cs, lo, hi, C.byref(plan)))
    return plan
