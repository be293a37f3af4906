"""Data-parallel layer for the prototype path: one process per GPU, RCCL over xGMI.

The reference is single-GPU; this is new capability:
  * training: the image batch is sharded over ranks, every rank runs the fused forward/backward on its
    own images, and the (tiny: 0.08-1.6 MB) parameter gradients are summed with ONE all-reduce of a flat
    fp32 bucket per optimizer step — the exchange is latency-bound, so one message, not one per tensor;
  * push: the image list is cut into contiguous ranges, each rank reduces its range on the GPU, and one
    all-gather of (value, image, flat index) per prototype is followed by a local lexicographic minimum
    that keeps the reference's tie-break (lowest image index, push_multiscale_optimization.py:137).
Backend: "nccl" (= RCCL on ROCm) for GPU tensors; the same code runs over "gloo" on CPU tensors, which is
how tests/test_dp_gloo.py covers the world_size > 1 logic without a GPU.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def _all_reduce_sum(t: torch.Tensor, group=None) -> None:
    """In-place sum all-reduce.  RCCL ("nccl") reduces device tensors directly; under the gloo rehearsal backend a
    device tensor is staged through the host (gloo's device-tensor support depends on the build)."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def _all_gather(t: torch.Tensor, group=None) -> List[torch.Tensor]:
    world = dist.get_world_size(group)
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.detach().cpu()
        out = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(out, h, group=group)
        return [o.to(t.device) for o in out]
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return out


def shard_range(n_items: int, rank: int, world_size: int) -> range:
    """Contiguous, balanced shard of ``range(n_items)`` for ``rank``."""
    base, rem = divmod(n_items, world_size)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


class FlatGradBucket:
    """One flat fp32 buffer holding the gradients of a fixed parameter list.

    ``attach()`` makes every parameter's ``.grad`` a VIEW of the flat buffer: autograd then accumulates straight into the
    bucket and ``all_reduce`` is the collective alone - no gather / scatter copy kernels around it (2 x n_params launches
    per step otherwise).  A parameter whose ``.grad`` was re-assigned or set to None since (``zero_grad(set_to_none=True)``,
    ``p.grad = None``) is copied in and re-attached; use ``zero()`` to clear gradients while keeping the views."""

    def __init__(self, params: Sequence[torch.nn.Parameter], attach: bool = False):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradBucket needs at least one trainable parameter")
        self.sizes = [p.numel() for p in self.params]
        dev = self.params[0].device
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p, n in zip(self.params, self.sizes):
            self.views.append(self.flat[off : off + n].view_as(p))
            off += n
        if attach:
            self.attach()

    def _is_view(self, p: torch.nn.Parameter, v: torch.Tensor) -> bool:
        g = p.grad
        return g is not None and g.data_ptr() == v.data_ptr() and g.shape == v.shape and g.dtype == v.dtype and g.is_contiguous()

    def attach(self):
        """Point every parameter's ``.grad`` at its slice of the flat buffer (existing gradient values are kept)."""
        for p, v in zip(self.params, self.views):
            if self._is_view(p, v):
                continue
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
            p.grad = v

    def zero(self):
        """Clear all gradients in one fill, keeping the views attached."""
        self.flat.zero_()

    def gather(self):
        for p, v in zip(self.params, self.views):
            if self._is_view(p, v):
                continue                     # autograd accumulated into the bucket already
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)

    def scatter(self):
        for p, v in zip(self.params, self.views):
            if not self._is_view(p, v):
                p.grad = v                   # from now on the gradient lives in the bucket

    def all_reduce(self, group=None, average: bool = False):
        """Sum (or mean) the bucket over the ranks with ONE collective; afterwards every ``.grad`` is a view of it."""
        self.gather()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            _all_reduce_sum(self.flat, group)
            if average:
                self.flat.div_(dist.get_world_size(group))
        self.scatter()


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None, average: bool = False) -> FlatGradBucket:
    bucket = FlatGradBucket(list(params))
    bucket.all_reduce(group=group, average=average)
    return bucket


def reduce_push_candidates(
    local_best_img: torch.Tensor,
    local_values: torch.Tensor,
    local_flat_idx: torch.Tensor,
    image_offset: int,
    group=None,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Combine per-rank push winners into the global ones.

    local_best_img [P] (position inside this rank's shard), local_values [P] (its masked minimum),
    local_flat_idx [P] (its flat latent index).  Returns (global image index [P] int64, value [P] f32,
    flat index [P] int64), identical on every rank.  Lexicographic minimum on (value, global image index):
    because shards are contiguous ranges, this equals the single-process ``argmin`` over all images."""
    gimg = local_best_img.to(torch.int64) + int(image_offset)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return gimg, local_values, local_flat_idx.to(torch.int64)
    world = dist.get_world_size(group)
    # pack to one int64 triple per prototype so a single all_gather moves everything (values via bit cast)
    vbits = local_values.to(torch.float32).contiguous().view(torch.int32).to(torch.int64)
    payload = torch.stack([vbits, gimg, local_flat_idx.to(torch.int64)], dim=0).contiguous()
    allp = torch.stack(_all_gather(payload, group), dim=0)   # [world, 3, P]
    vals = allp[:, 0].to(torch.int32).view(torch.float32)    # [world, P]
    imgs = allp[:, 1]
    flats = allp[:, 2]
    # ranks hold increasing image ranges, so the first rank attaining the minimum value holds the lowest image
    vmin = vals.min(dim=0).values
    is_min = vals == vmin[None]
    big = torch.iinfo(torch.int64).max
    cand_img = torch.where(is_min, imgs, torch.full_like(imgs, big))
    win_rank = cand_img.argmin(dim=0)
    ar = torch.arange(vals.shape[1], device=vals.device)
    return imgs[win_rank, ar], vals[win_rank, ar], flats[win_rank, ar]


def gather_push_patches(local_patches: torch.Tensor, owner_mask: torch.Tensor, group=None) -> torch.Tensor:
    """[P, Cs] feature vectors: each rank fills the rows it owns (owner_mask), zeros elsewhere; a sum
    all-reduce assembles the full bank (exact: every row has exactly one non-zero contributor)."""
    out = torch.where(owner_mask[:, None], local_patches, torch.zeros_like(local_patches)).contiguous()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        _all_reduce_sum(out, group)
    return out


class DataParallelStep:
    """Minimal data-parallel optimizer step for the prototype path: every rank runs forward + backward
    on its own images, the parameter gradients are averaged with ONE all-reduce of a flat bucket after the ``iter_size``
    accumulation window, then every rank applies the same optimizer step and - in the group phase - the same simplex
    re-projection of the group projections (segmentation/model/module_multiscale_group_train.py:323-338:
    ``manual_backward(loss / iter_size)``, ``optimizer.step()``, ``projection_simplex_sort`` per projection).
    Kernels, all-reduce order and the update are deterministic, so replicas stay bit-identical.

    Gradient semantics: the mean over ranks of per-rank gradients, i.e. the gradient of the mean of the per-rank losses
    (equal to the single-process loss over the union batch when each rank's loss is a mean over equally many terms)."""

    def __init__(self, net: torch.nn.Module, optimizer: torch.optim.Optimizer, iter_size: int = 1, group=None,
                 grad_hook=None, scheduler=None):
        """``grad_hook`` (optional, no arguments) runs on every rank AFTER the all-reduce and BEFORE ``optimizer.step()`` - the
        place of the reference's gradient edits, e.g. ``last_layer_group.weight.grad *= group_class_identity.T`` when
        ``incorrect_strength == 0`` (module_multiscale_group_train.py:327-328); ``scheduler.step()`` (optional) follows the
        optimizer step and precedes the simplex re-projection, as upstream (:333-338)."""
        self.net = net
        self.optimizer = optimizer
        self.iter_size = int(iter_size)
        self.group = group
        self.grad_hook = grad_hook
        self.scheduler = scheduler
        self.iter_steps = 0
        self.bucket = FlatGradBucket(self._trainable(), attach=True)

    def _trainable(self):
        return [p for g in self.optimizer.param_groups for p in g["params"] if p.requires_grad]

    def _sync_bucket(self):
        """The optimizer's trainable parameters can change between steps (the warm-up -> joint phase switch unfreezes the
        backbone, a param group is added): a parameter outside the bucket would be neither all-reduced nor cleared and the
        replicas would drift apart silently.  Rebuild the bucket when the set differs (existing gradients are carried over)."""
        now = self._trainable()
        if len(now) != len(self.bucket.params) or any(a is not b for a, b in zip(now, self.bucket.params)):
            self.bucket = FlatGradBucket(now, attach=True)

    def backward(self, loss: torch.Tensor) -> bool:
        """Accumulate ``loss / iter_size``; at the end of the window all-reduce, step and re-project.  Returns True when
        the optimizer stepped."""
        (loss / self.iter_size).backward()
        self.iter_steps += 1
        if self.iter_steps < self.iter_size:
            return False
        self.iter_steps = 0
        self._sync_bucket()
        self.bucket.all_reduce(group=self.group, average=True)
        if self.grad_hook is not None:
            self.grad_hook()
        self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()
        projections = getattr(self.net, "group_projection", None)
        if projections is not None:
            from .utils import projection_simplex_sort

            for gp in projections:
                gp.weight.data = projection_simplex_sort(gp.weight.data)
        self.bucket.zero()                   # (not zero_grad(set_to_none=True): the .grad views stay attached)
        return True
