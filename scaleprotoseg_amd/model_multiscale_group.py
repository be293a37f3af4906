"""Drop-in group-phase ``PPNetMultiScale`` (ScaleProtoSeg phase 2) on the MI355X kernels.

Mirrors segmentation/model/model_multiscale_group.py:82-519: same constructor, ``group_projection``
(ModuleList of bias-free Linear(n_k -> G)), ``last_layer_group``, ``group_class_identity``,
``compute_group`` (returns a list), ``state_dict`` keys (no ``last_layer``), simplex initialisation.

Fused form of the grouping head: every class's projection is a row block of ONE dense [G*K', P] matrix
(zeros outside the class's prototype columns), so ``cat_k(act[:, idx_k] @ W_k^T)`` is the kernel's head
contraction act @ Wd^T; ``exp`` and the ``last_layer_group`` product follow inside the same kernel (a second
split-bf16 MFMA on the unit tiles), and the backward forms dUnits in registers.  The reference instead gathers per
class with a host sync per class per forward (:297-301).
"""
from __future__ import annotations

import os
from typing import Any, List, Optional, Tuple

import torch
import torch.nn as nn

from .functional import (MAX_FUSED_HEAD_ROWS, MAX_FUSED_TAIL_CLASSES, GroupTables, SpxError, cross_entropy_from_logits,
                         group_dense, group_exp, proto_head_forward, shifted_labels_i32, wide_group_tail, wide_linear)
from .model_multiscale import _PrototypeBankMixin, _build_add_on, _first_add_on_channels
from .utils import projection_simplex_sort


class PPNetMultiScale(_PrototypeBankMixin, nn.Module):
    """Group-phase module; the reference names it ``PPNetMultiScale`` in model_multiscale_group.py (:82), so
    ``from scaleprotoseg_amd.model_multiscale_group import PPNetMultiScale`` works as upstream.  The package exports it
    as ``PPNetMultiScaleGroup`` to keep it apart from the prototype-phase class."""

    def __init__(
        self,
        features: nn.Module,
        img_size: int,
        prototype_shape: Tuple[int, int, int, int],
        proto_layer_rf_info: List[float],
        num_classes: int,
        init_weights: bool = True,
        prototype_activation_function: str = "log",
        add_on_layers_type: str = "bottleneck",
        bottleneck_stride: Optional[int] = None,
        patch_classification: bool = False,
        num_scales: int = 4,
        scale_head_type: Optional[str] = None,
        num_groups: int = 3,
        incorrect_strength: float = -0.5,
        equiv_path: Optional[os.PathLike] = None,
        equiv_scale_weight: float = 0.25,
    ):
        super().__init__()
        self.img_size = img_size
        self.bottleneck_stride = bottleneck_stride
        self.patch_classification = patch_classification
        self.num_groups = num_groups
        self.incorrect_strength = incorrect_strength
        if scale_head_type is not None:
            raise SpxError("scale_head_type is not supported: no reference config uses it")
        if equiv_path is not None:
            raise SpxError("equiv_path initialisation is deprecated in the reference (NOT USED) and not built")
        self.scale_head = None
        self.prototype_activation_function = prototype_activation_function
        self._init_bank(prototype_shape, num_classes, num_scales)
        self.proto_layer_rf_info = proto_layer_rf_info
        self.features = features
        in_ch = _first_add_on_channels(features)
        self.add_on_layers = _build_add_on(add_on_layers_type, in_ch, self.prototype_shape[1], bottleneck_stride)
        self._group_index_cache = None
        self._initialize_groups()
        if init_weights:
            self._initialize_weights()

    # ---- group bookkeeping ------------------------------------------------------------------------
    def _present_classes(self) -> List[int]:
        ident = self.prototype_class_identity
        return [k for k in range(self.num_classes) if int(ident[:, k].sum().item()) > 0]

    def _initialize_groups(self):
        """group_projection / group_class_identity / last_layer_group (model_multiscale_group.py:249-269)."""
        ident = self.prototype_class_identity
        present = self._present_classes()
        self.group_projection = nn.ModuleList(
            [nn.Linear(int(ident[:, k].sum().item()), self.num_groups, bias=False) for k in present]
        )
        n_groups = self.num_groups * len(present)
        self.group_class_identity = torch.zeros(n_groups, self.num_classes)
        for j, k in enumerate(present):
            self.group_class_identity[j * self.num_groups : (j + 1) * self.num_groups, k] = 1
        self.last_layer_group = nn.Linear(n_groups, self.num_classes, bias=False)
        dev = self.prototype_vectors.device          # re-initialisation after .cuda() (finetune_wandb_group.py:80, prune)
        self.group_projection.to(dev)
        self.last_layer_group.to(dev)
        self._group_index_cache = None

    def _group_index(self, device):
        """(rows, cols, number of units, GroupTables) of every group-projection weight inside the dense [NG, P] head matrix."""
        ident = self.prototype_class_identity
        key = (self._tables_version, ident._version, tuple(gp.weight.shape for gp in self.group_projection), str(device))
        c = self._group_index_cache
        if c is not None and c[0] is ident and c[1] == key:
            return c[2]
        P = self.num_prototypes
        rows, cols, r0 = [], [], 0
        col_block, col_local = torch.full((P,), -1, dtype=torch.int32), torch.zeros(P, dtype=torch.int32)
        row_block, row_local, block_cols = [], [], []
        for j, k in enumerate(self._present_classes()):
            idx = torch.nonzero(ident[:, k]).flatten()
            g = self.group_projection[j].weight.shape[0]
            rr = torch.arange(r0, r0 + g).repeat_interleave(idx.numel())
            cc = idx.repeat(g)
            rows.append(rr)
            cols.append(cc)
            col_block[idx] = j
            col_local[idx] = torch.arange(idx.numel(), dtype=torch.int32)
            row_block += [j] * g
            row_local += list(range(g))
            block_cols.append(int(idx.numel()))
            r0 += g
        rows_, cols_ = torch.cat(rows), torch.cat(cols)
        i32 = lambda t: torch.as_tensor(t, dtype=torch.int32).to(device).contiguous()
        tables = None
        if torch.device(device).type == "cuda" and len(block_cols) <= 192:
            tables = GroupTables(i32(row_block), i32(row_local), i32(col_block), i32(col_local), i32(rows_), i32(cols_),
                                 block_cols, r0, P)
        out = (rows_.to(device), cols_.to(device), r0, tables)
        self._group_index_cache = (ident, key, out)
        return out

    def _dense_group_matrix(self) -> torch.Tensor:
        dev = self.prototype_vectors.device
        rows, cols, ng, tables = self._group_index(dev)
        if tables is not None:
            return group_dense(tables, [gp.weight for gp in self.group_projection])      # one launch (spx_group_dense)
        vals = torch.cat([gp.weight.reshape(-1) for gp in self.group_projection])
        return torch.zeros(ng, self.num_prototypes, device=dev, dtype=vals.dtype).index_put((rows, cols), vals)

    def _group_units(self, prototype_activations: torch.Tensor):
        """(units, g) behind ``prototype_activations``: what the forward that produced this very tensor left on it (the
        kernels' own unit product / group activations, still attached to the autograd graph), else the dense unit product
        act . Wd^T on the fp32 MFMA product kernels.  One of the two is None."""
        tag = getattr(prototype_activations, "spx_group", None)
        if tag is not None and tag[0] == prototype_activations._version:
            return tag[1], tag[2]
        if prototype_activations.dim() != 2 or prototype_activations.shape[1] != self.num_prototypes:
            raise SpxError(f"prototype activations must be [M, {self.num_prototypes}], got {tuple(prototype_activations.shape)}")
        return wide_linear(prototype_activations, self._dense_group_matrix()), None

    def compute_group(self, prototype_activations: torch.Tensor) -> List[torch.Tensor]:
        """List of per-class group activations exp(act[:, idx_k] @ W_k^T) (model_multiscale_group.py:283-303), as column
        blocks of the dense form exp(act . Wd^T): for the activations a forward of this module returned, views of the
        [M, U] tensor the fused kernel wrote (a gradient on them enters its backward as part of dUnits - the path
        KLDLossGroup takes, module_multiscale_group_train.py:242-262); for any other activations the same product on the
        fp32 MFMA kernels followed by the exp kernel.  No per-class gathers, no host syncs."""
        units, g = self._group_units(prototype_activations)
        if g is None:
            g = group_exp(units)
        return list(torch.split(g, [int(gp.weight.shape[0]) for gp in self.group_projection], dim=1))

    def run_last_layer(self, prototype_activations: torch.Tensor) -> torch.Tensor:
        """last_layer_group(cat(compute_group(act))) (model_multiscale_group.py:305-308): exp fused into the product kernel's
        operand staging."""
        units, g = self._group_units(prototype_activations)
        if units is None:
            return wide_linear(g, self.last_layer_group.weight)
        return wide_group_tail(units, self.last_layer_group.weight)

    # ---- forward ------------------------------------------------------------------------------------
    def forward_from_conv_features(
        self, conv_features, return_activations: bool = False, return_distances: bool = False, ce_target=None
    ) -> Any:
        """Same return-tuple rules as model_multiscale_group.py:404-452.  Extension: ``ce_target`` ([B, H, W] at the latent
        resolution, 0 = void, 1..K) computes the pixel-wise cross entropy in the kernel that produces the logits and hangs
        it on the returned logits (``logits.spx_ce``), exactly as the prototype-phase module does."""
        if isinstance(conv_features, list):
            return [self.forward_from_conv_features(c) for c in conv_features]
        if not (hasattr(self, "patch_classification") and self.patch_classification):
            raise Exception("Original Prototype Network Implementation")
        self._check_fusable()
        B, _, H, W = conv_features.shape
        wd = self._dense_group_matrix()
        if callable(self.prototype_activation_function):
            # model_multiscale_group.py:393-394: the user's function on the distance map as ordinary torch code; both head
            # products on the fp32 MFMA product kernels
            if ce_target is not None:
                raise SpxError("ce_target needs a built-in similarity ('log' or 'linear')")
            _, dist, _ = proto_head_forward(conv_features, self.prototype_vectors, None, self._layout(1), want_distances=True,
                                            epsilon=self.epsilon, activation="log")
            act = self.prototype_activation_function(dist).permute(0, 2, 3, 1).reshape(B * H * W, -1)
            units = wide_linear(act, wd)
            act.spx_group = (act._version, units, None)
            logits = wide_group_tail(units, self.last_layer_group.weight).reshape(B, H, W, -1)
            if return_activations and not return_distances:
                return logits, act
            if return_activations and return_distances:
                return logits, dist, act
            return logits, dist
        want_dist = return_distances or not return_activations
        wg = self.last_layer_group.weight
        rows, k2 = int(wd.shape[0]), int(wg.shape[0])
        kw = dict(want_distances=want_dist, epsilon=self.epsilon, activation=self.prototype_activation_function)
        ce_labels = fused_ce = None
        if ce_target is not None:
            if tuple(ce_target.shape) != (B, H, W):
                raise SpxError(f"ce_target must be [{B}, {H}, {W}] (latent grid), got {tuple(ce_target.shape)}")
            ce_labels = shifted_labels_i32(ce_target.reshape(B, -1), conv_features.device)
        if rows <= MAX_FUSED_HEAD_ROWS and k2 <= MAX_FUSED_TAIL_CLASSES:
            # whole grouping head in the kernels: units = act . Wd^T, exp, last_layer_group (:303-308)
            out = proto_head_forward(
                conv_features, self.prototype_vectors, wd, self._layout(rows), want_activations=return_activations,
                group_tail=wg, ce_labels=ce_labels, **kw,
            )
            logits, dist, act = out[:3]
            if ce_labels is not None:
                fused_ce = out[3]
            if act is not None:
                act.spx_group = (act._version, None, out[-1])        # exp(units) [M, U], written by the kernel
        elif rows <= MAX_FUSED_HEAD_ROWS:
            # up to 160 units but more than 32 classes: the unit product stays in the kernel, the tail is the fp32 MFMA product kernel (exp fused into its operand staging)
            units, dist, act = proto_head_forward(
                conv_features, self.prototype_vectors, wd, self._layout(rows), want_activations=return_activations, **kw,
            )
            logits = wide_group_tail(units, wg)
            if act is not None:
                act.spx_group = (act._version, units, None)
        else:
            # group_scaleproto_ade.gin (150 classes x 3 groups = 450 units) / _coco.gin (546): the kernel hands out the
            # [pixel][P] activations once, both products are the fp32 MFMA product kernels (csrc/spx_gemm.hip) on them
            _, dist, act = proto_head_forward(
                conv_features, self.prototype_vectors, None, self._layout(1), want_activations=True, **kw,
            )
            units = wide_linear(act, wd)
            act.spx_group = (act._version, units, None)
            logits = wide_group_tail(units, wg)
        if ce_labels is not None and fused_ce is None:
            fused_ce = cross_entropy_from_logits(logits, ce_labels)       # heads the fused kernels do not carry
        logits = logits.reshape(B, H, W, -1)
        if fused_ce is not None:
            fused_ce.target = ce_target
            fused_ce.target_version = ce_target._version      # an in-place edit of the labels afterwards voids the attachment
            logits.spx_ce = fused_ce
        if return_activations and not return_distances:
            return logits, act
        if return_activations and return_distances:
            return logits, dist, act
        return logits, dist

    def prune_prototypes(self, prototypes_to_prune: List[int]):
        """Not in the upstream group class (it receives an already pruned bank by attribute assignment,
        finetune_wandb_group.py:74-80); provided so a post-push prototype-phase checkpoint loads through
        ``checkpoint.load_reference_state_dict``: prunes the bank and the class / scale tables, then rebuilds
        ``group_projection`` / ``group_class_identity`` / ``last_layer_group`` for the new table (fresh default-initialised
        layers, as ``_initialize_groups`` upstream: call ``_initialize_weights`` or load a state_dict next)."""
        self._prune_bank(prototypes_to_prune)
        self._initialize_groups()

    # ---- init ---------------------------------------------------------------------------------------
    def set_last_layer_incorrect_connection(self):
        pos = torch.t(self.group_class_identity).to(self.last_layer_group.weight.device)  # :480-491
        self.last_layer_group.weight.data.copy_(1 * pos + self.incorrect_strength * (1 - pos))
        from .functional import invalidate_pack_cache

        invalidate_pack_cache()

    def _initialize_weights(self, equiv_path=None, equiv_scale_weight: float = 0.25):
        for m in self.add_on_layers.modules():  # :493-519
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if equiv_path is not None:
            raise SpxError("equiv_path initialisation is deprecated in the reference (NOT USED) and not built")
        for gp in self.group_projection:
            gp.weight.data = projection_simplex_sort(gp.weight.data)
        self.set_last_layer_incorrect_connection()

    def __repr__(self):
        return (
            "PPNet(\n\tfeatures: {},\n\timg_size: {},\n\tprototype_shape: {},\n\tproto_layer_rf_info: {},\n"
            "\tnum_classes: {},\n\tepsilon: {}\n)"
        ).format(self.features, self.img_size, self.prototype_shape, self.proto_layer_rf_info, self.num_classes, self.epsilon)


PPNetMultiScaleGroup = PPNetMultiScale


def construct_PPNet_Group(
    features: nn.Module,
    img_size: int = 224,
    prototype_shape: Tuple[int, int, int, int] = (2000, 512, 1, 1),
    num_classes: int = 200,
    prototype_activation_function: str = "log",
    add_on_layers_type: str = "bottleneck",
    scale_head_type: Optional[str] = None,
    **kwargs,
) -> PPNetMultiScale:
    """Factory with the reference's argument meaning (model_multiscale_group.py:589-624)."""
    return PPNetMultiScale(
        features=features, img_size=img_size, prototype_shape=prototype_shape, proto_layer_rf_info=[],
        num_classes=num_classes, init_weights=True, prototype_activation_function=prototype_activation_function,
        add_on_layers_type=add_on_layers_type, scale_head_type=scale_head_type, **kwargs,
    )
