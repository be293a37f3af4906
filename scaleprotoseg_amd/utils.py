"""Small host-side helpers the hot path's callers need."""
from __future__ import annotations

import numpy as np
import torch


@torch.no_grad()
def projection_simplex_sort(v: torch.Tensor, z: float = 1.0) -> torch.Tensor:
    """Row-wise Euclidean projection onto the probability simplex (segmentation/utils.py:113-124).

    Init-time / once-per-optimizer-step work on [G, n_k] matrices; stays stock PyTorch on v's device."""
    n = v.size(1)
    u, _ = torch.sort(v, descending=True)
    css = torch.cumsum(u, 1) - z
    k = torch.arange(n, device=v.device).type_as(v) + 1
    cond = (u - css / k) > 0
    rho, rho_idx = (k * cond).max(1)
    theta = torch.gather(css, 1, rho_idx[:, None]) / rho[:, None]
    return torch.clamp(v - theta, min=0)


def _pil_nearest_index(n_in: int, n_out: int) -> np.ndarray:
    """Source index per output index of PIL's NEAREST resize: position (i + 0.5) * n_in / n_out, produced by
    repeated double-precision addition of the step (the accumulation order decides ties), truncated."""
    step = float(n_in) / float(n_out)
    inc = np.full(n_out, step, dtype=np.float64)
    inc[0] = 0.0 + step * 0.5
    pos = np.add.accumulate(inc)  # sequential, like the C loop
    return np.minimum(pos.astype(np.int64), n_in - 1)


def resize_label(label: np.ndarray, size) -> torch.Tensor:
    """Nearest-neighbour label resize with PIL's sampling rule (segmentation/data/dataset.py:22-30).

    size is (W, H).  The reference resizes through ``PIL.Image.resize(..., NEAREST)`` and warns that other
    nearest rules misalign labels; the rule is restated here (checked against PIL on 3000 random sizes and
    against the reference's own resize_label in tests/golden/push_argmin.npz) so the push path needs no PIL."""
    label = np.asarray(label)
    h_in, w_in = label.shape
    w_out, h_out = int(size[0]), int(size[1])
    rows = _pil_nearest_index(h_in, h_out)
    cols = _pil_nearest_index(w_in, w_out)
    return torch.from_numpy(np.ascontiguousarray(label[np.ix_(rows, cols)]).astype(np.int64))
