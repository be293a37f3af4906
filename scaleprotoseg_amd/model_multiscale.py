"""Drop-in ``PPNetMultiScale`` (prototype phase) on the MI355X kernels.

Mirrors the public surface of the reference class
(segmentation/model/model_multiscale.py:71-477: constructor, properties,
``forward`` / ``forward_from_conv_features`` / ``push_forward`` / ``prune_prototypes``,
``state_dict`` keys ``prototype_vectors``, ``ones``, ``last_layer.weight``,
``features.*``) while the distance -> similarity -> head chain runs in one HIP
kernel (scaleprotoseg_amd.functional).  There is no CPU path: tensors must live
on an AMD GPU when the distance methods are called.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from .functional import (MAX_FUSED_HEAD_ROWS, BankLayout, ClassGather, SpxError, class_gather_table,
                         cross_entropy_from_logits, proto_head_forward, shifted_labels_i32, wide_linear)
from .loss import ClassDistances


def _first_add_on_channels(features: nn.Module) -> int:
    """Backbone output width, by the reference's name dispatch (model_multiscale.py:153-171)."""
    name = str(features).upper()
    convs = lambda m: [i for i in m.modules() if isinstance(i, nn.Conv2d)]
    if name.startswith("VGG") or name.startswith("RES"):
        return convs(features)[-1].out_channels
    if name.startswith("DENSE"):
        return [i for i in features.modules() if isinstance(i, nn.BatchNorm2d)][-1].num_features
    if name.startswith("DEEPLAB"):
        return convs(features)[-2].out_channels
    if name.startswith("MSC"):
        return convs(features.base)[-2].out_channels
    raise Exception(f"{name[:10]} base_architecture NOT implemented")


def _build_add_on(kind: str, in_ch: int, proto_ch: int, bottleneck_stride: Optional[int]) -> nn.Sequential:
    """Add-on stack between backbone and prototype layer (model_multiscale.py:173-218)."""
    if kind == "deeplab_simple":
        return nn.Sequential(nn.Sigmoid())
    layers: List[nn.Module] = []
    if kind == "bottleneck_pool":
        layers += [nn.Conv2d(in_ch, in_ch, kernel_size=3, padding=1, stride=bottleneck_stride), nn.ReLU()]
    if kind.startswith("bottleneck"):
        cur = in_ch
        while cur > proto_ch or len(layers) == 0:
            out = max(proto_ch, cur // 2)
            layers += [nn.Conv2d(cur, out, kernel_size=1), nn.ReLU(), nn.Conv2d(out, out, kernel_size=1)]
            if out > proto_ch:
                layers.append(nn.ReLU())
            else:
                assert out == proto_ch
                layers.append(nn.Sigmoid())
            cur = cur // 2
        return nn.Sequential(*layers)
    return nn.Sequential(
        nn.Conv2d(in_ch, proto_ch, kernel_size=1), nn.ReLU(), nn.Conv2d(proto_ch, proto_ch, kernel_size=1), nn.Sigmoid()
    )


class _PrototypeBankMixin:
    """State and helpers shared by the prototype-phase and group-phase modules."""

    _tables_version = 0

    def __setattr__(self, name, value):
        # prototype_class_identity / scale_num_prototypes are plain attributes that callers re-assign
        # (finetune_wandb_group.py:77-78, prune_prototypes): every derived cache is keyed on this counter
        if name in ("prototype_class_identity", "scale_num_prototypes"):
            object.__setattr__(self, "_tables_version", self._tables_version + 1)
        super().__setattr__(name, value)

    def _init_bank(self, prototype_shape, num_classes: int, num_scales: int):
        self.epsilon = 1e-4  # model_multiscale.py:106
        self.num_scales = num_scales
        self.prototype_vectors = nn.Parameter(torch.rand(prototype_shape), requires_grad=True)  # :111
        P = self.prototype_vectors.shape[0]
        # one-hot prototype -> class table, scale-major / class-minor blocks (:129-141)
        self.prototype_class_identity = torch.zeros(P, num_classes)
        per_scale = P // num_scales
        per_cs = P // num_classes // num_scales
        for s in range(num_scales):
            for k in range(num_classes):
                self.prototype_class_identity[s * per_scale + k * per_cs : s * per_scale + (k + 1) * per_cs, k] = 1
        self.scale_num_prototypes: Dict[int, Tuple[int, int]] = {
            s: (s * per_scale, (s + 1) * per_scale) for s in range(num_scales)
        }  # :146-149
        # kept for state_dict parity only: the kernels never read it (|x|^2 is computed from the staged tile)
        self.ones = nn.Parameter(torch.ones(prototype_shape), requires_grad=False)  # :222

    @property
    def prototype_shape(self):
        return self.prototype_vectors.shape

    @property
    def num_prototypes(self) -> int:
        return self.prototype_vectors.shape[0]

    @property
    def num_classes(self) -> int:
        return self.prototype_class_identity.shape[1]

    def _layout(self, head_rows: int) -> BankLayout:
        cs = int(self.prototype_vectors.shape[1]) * int(self.prototype_vectors.shape[2]) * int(self.prototype_vectors.shape[3])
        return BankLayout(
            num_prototypes=self.num_prototypes,
            num_classes=head_rows,
            num_scales=self.num_scales,
            channels_per_scale=cs,
            scale_ranges=tuple(tuple(int(v) for v in self.scale_num_prototypes[s]) for s in range(self.num_scales)),
        )

    def _class_gather(self, target_labels: torch.Tensor, layout: BankLayout, device) -> ClassGather:
        """Kernel-side description of 'only my class's prototypes' for labels in the reference's convention
        (0 = void, 1..K = class; module_multiscale.py:234-242, loss.py:73).  The (class, slot) table is cached
        until prototype_class_identity is re-assigned (prune_prototypes, attribute assignment), edited in place
        (tensor version counter) or the scale table changes."""
        ident = self.prototype_class_identity
        tag = (self._tables_version, ident._version, layout.scale_ranges, str(device))
        cache = getattr(self, "_gather_cache", None)
        if cache is None or cache[0] is not ident or cache[1] != tag:
            cache = (ident, tag, class_gather_table(layout, ident, device))      # holds ident: its id cannot be recycled
            self._gather_cache = cache
        keys, width, table = cache[2]
        B = target_labels.shape[0]
        labels0 = shifted_labels_i32(target_labels.reshape(B, -1), device)
        return ClassGather(labels=labels0, keys=keys, width=width, table=table)

    def _check_fusable(self):
        if getattr(self, "scale_head", None) is not None:
            raise SpxError("scale_head aggregation has no fused kernel (every reference config sets scale_head_type=None)")
        if self.prototype_vectors.shape[2] != 1 or self.prototype_vectors.shape[3] != 1:
            raise SpxError("only 1x1 prototypes are supported (all reference configs)")

    def _prune_bank(self, prototypes_to_prune) -> List[int]:
        """The bank half of ``prune_prototypes`` (model_multiscale.py:400-423, :428-432): shrinks ``prototype_vectors``,
        ``ones`` and ``prototype_class_identity`` to the kept rows and re-packs ``scale_num_prototypes``.  Returns the
        kept indices (ascending).  The parameters are re-created, as upstream (optimizers holding the old ones go stale)."""
        drop = set(int(i) for i in prototypes_to_prune)
        keep = sorted(set(range(self.num_prototypes)) - drop)
        prev_hi = 0
        for s in range(self.num_scales):
            lo, hi = self.scale_num_prototypes[s]
            n = len(set(range(lo, hi)) - drop)
            self.scale_num_prototypes[s] = (prev_hi, prev_hi + n)
            prev_hi += n
        self.prototype_vectors = nn.Parameter(self.prototype_vectors.data[keep, ...], requires_grad=True)
        self.ones = nn.Parameter(self.ones.data[keep, ...], requires_grad=False)
        self.prototype_class_identity = self.prototype_class_identity[keep, :]      # bumps _tables_version
        return keep

    # -- reference methods on the distance path ---------------------------------------------------
    def conv_features(self, x):
        """features -> add_on_layers, list-aware for MSC training inputs (model_multiscale.py:246-253)."""
        x = self.features(x)
        if isinstance(x, list):
            return [self.add_on_layers(xs) for xs in x]
        return self.add_on_layers(x)

    def _scale_l2_convolution(self, x: torch.Tensor) -> torch.Tensor:
        """[B,S*Cs,H,W] -> distances [B,P,H,W] (model_multiscale.py:283-317), one HIP launch."""
        self._check_fusable()
        _, d, _ = proto_head_forward(
            x, self.prototype_vectors, None, self._layout(1), want_distances=True, epsilon=self.epsilon,
            activation="linear",
        )
        return d

    def prototype_distances(self, x: torch.Tensor) -> torch.Tensor:
        return self._scale_l2_convolution(self.conv_features(x))  # :319-322

    def distance_2_similarity(self, distances: torch.Tensor) -> torch.Tensor:
        """Stand-alone similarity for callers outside the fused path (model_multiscale.py:324-330)."""
        if self.prototype_activation_function == "log":
            return torch.log((distances + 1) / (distances + self.epsilon))
        if self.prototype_activation_function == "linear":
            return -distances
        return self.prototype_activation_function(distances)

    def push_forward(self, x):
        """(conv_features, distances) for the push (model_multiscale.py:390-398)."""
        conv = self.conv_features(x)
        if isinstance(conv, list):
            return [(c, self._scale_l2_convolution(c)) for c in conv]
        return conv, self._scale_l2_convolution(conv)

    def push_min_distances(self, x, labels_for_grid, void_class=None, max_dist: float = 1e10):
        """Class-masked per-prototype minimum of the distance map of ``x`` over its latent grid, taken INSIDE the distance
        kernel (push_multiscale_optimization.py:68-91 without the [B, P, H, W] map): (indices int64 [B, P], values [B, P]).
        ``labels_for_grid((H, W))`` returns the label map [B, H, W] at the latent resolution (the grid is only known after
        the backbone has run).  Returns None when the fused kernel does not apply (MSC list input, a
        prototype_class_identity that is not one-hot, features off the GPU): the caller then reduces the written map."""
        from .functional import class_gather_table, identity_is_one_hot, push_min_from_features

        conv = self.conv_features(x)
        if isinstance(conv, list) or not conv.is_cuda:
            return None
        self._check_fusable()
        layout = self._layout(1)
        ident = self.prototype_class_identity
        tag = (self._tables_version, ident._version, layout.scale_ranges, str(conv.device))
        cache = getattr(self, "_push_key_cache", None)
        if cache is None or cache[0] is not ident or cache[1] != tag:
            keys = class_gather_table(layout, ident, conv.device)[0] if identity_is_one_hot(ident) else None
            cache = (ident, tag, keys)
            self._push_key_cache = cache
        if cache[2] is None:
            return None
        labels = labels_for_grid((conv.shape[2], conv.shape[3]))
        return push_min_from_features(conv, self.prototype_vectors, layout, labels, ident, void_class=void_class,
                                      max_dist=max_dist, keys=cache[2])

    def forward(self, x, **kwargs):
        conv = self.conv_features(x)
        if isinstance(conv, list):  # MSC
            return [self.forward_from_conv_features(c, **kwargs) for c in conv]
        return self.forward_from_conv_features(conv, **kwargs)


class PPNetMultiScale(_PrototypeBankMixin, nn.Module):
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
    ):
        super().__init__()
        self.img_size = img_size
        self.bottleneck_stride = bottleneck_stride
        self.patch_classification = patch_classification
        if scale_head_type is not None:
            raise SpxError("scale_head_type is not supported: no reference config uses it (SURVEY.md §2 #5)")
        self.scale_head = None
        self.prototype_activation_function = prototype_activation_function
        self._init_bank(prototype_shape, num_classes, num_scales)
        self.proto_layer_rf_info = proto_layer_rf_info
        self.features = features
        in_ch = _first_add_on_channels(features)
        self.add_on_layers = _build_add_on(add_on_layers_type, in_ch, self.prototype_shape[1], bottleneck_stride)
        self.last_layer = nn.Linear(self.num_prototypes, self.num_classes, bias=False)  # :225
        if init_weights:
            self._initialize_weights()

    def run_last_layer(self, prototype_activations: torch.Tensor) -> torch.Tensor:
        """last_layer on given activations (model_multiscale.py:243-244; callers outside the fused path): the fp32 MFMA
        product kernel."""
        return wide_linear(prototype_activations, self.last_layer.weight)

    def forward_from_conv_features(
        self, conv_features, return_activations: bool = False, return_distances: bool = False, target_labels=None,
        ce_target=None,
    ) -> Any:
        """Same return-tuple rules as model_multiscale.py:340-388.

        Extension: with ``target_labels`` ([B, H, W] at the latent resolution, 0 = void, 1..K, i.e.
        the tensor the training step passes to the losses, module_multiscale.py:234-242) the distance entry of the
        tuple is a ``ClassDistances`` ([B, J, H*W]: per pixel only the distances to its own class's prototypes, the
        entries KLDLoss reads) and the P-wide fp32 map is never written; ``scaleprotoseg_amd.loss.KLDLoss`` takes it
        as is.  With ``ce_target`` (same convention and grid) the pixel-wise cross entropy of the logits is computed in the
        kernel's logits epilogue and travels on the returned logits as ``logits.spx_ce`` (loss, argmax prediction);
        ``scaleprotoseg_amd.loss.PixelWiseCrossEntropyLoss(ignore_index=-1)`` called with that ``logits`` and the same
        ``ce_target`` returns it instead of running a second pass (loss.py:9-48, caller module_multiscale.py:239)."""
        if isinstance(conv_features, list):
            return [self.forward_from_conv_features(c) for c in conv_features]  # flags dropped, as in :359
        if not (hasattr(self, "patch_classification") and self.patch_classification):
            raise Exception("Original Prototype Network Implementation")
        self._check_fusable()
        B, _, H, W = conv_features.shape
        if callable(self.prototype_activation_function):
            return self._forward_callable_similarity(conv_features, return_activations, return_distances, target_labels, ce_target)
        want_dist = return_distances or not return_activations
        wide = self.num_classes > MAX_FUSED_HEAD_ROWS          # e.g. scaleproto_coco.gin: 182 classes
        layout = self._layout(1 if wide else self.num_classes)
        gather = None
        if target_labels is not None and want_dist:
            if tuple(target_labels.shape) != (B, H, W):
                raise SpxError(f"target_labels must be [{B}, {H}, {W}] (latent grid), got {tuple(target_labels.shape)}")
            gather = self._class_gather(target_labels, layout, conv_features.device)
        ce_labels = fused_ce = None
        if ce_target is not None:
            if tuple(ce_target.shape) != (B, H, W):
                raise SpxError(f"ce_target must be [{B}, {H}, {W}] (latent grid), got {tuple(ce_target.shape)}")
            ce_labels = shifted_labels_i32(ce_target.reshape(B, -1), conv_features.device)
        out = proto_head_forward(
            conv_features, self.prototype_vectors, None if wide else self.last_layer.weight, layout,
            want_distances=want_dist and gather is None, want_activations=return_activations or wide,
            epsilon=self.epsilon, activation=self.prototype_activation_function, class_gather=gather,
            ce_labels=None if wide else ce_labels,
        )
        logits, dist, act = out[:3]
        if wide:      # the kernel hands out the activations once; the 182-row head is the fp32 MFMA product kernel (csrc/spx_gemm.hip) on them
            logits = wide_linear(act, self.last_layer.weight)
            if ce_labels is not None:
                fused_ce = cross_entropy_from_logits(logits, ce_labels)
        elif ce_labels is not None:
            fused_ce = out[3]
        if gather is not None:
            dist = ClassDistances(values=dist, labels=gather.labels, table=gather.table, grid=(H, W), target=target_labels,
                                  target_version=target_labels._version)
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

    def _forward_callable_similarity(self, conv_features, return_activations, return_distances, target_labels, ce_target):
        """A user-supplied ``prototype_activation_function`` (model_multiscale.py:329-330: any callable on the [B, P, H, W]
        distance map): the distance kernel writes the map, the user's function runs on it as ordinary torch code (autograd
        included), and the head is the fp32 MFMA product kernel on its NHWC view - the reference's op order with its two
        GEMM-shaped pieces on this package's kernels."""
        if target_labels is not None or ce_target is not None:
            raise SpxError("target_labels / ce_target need a built-in similarity ('log' or 'linear')")
        B, _, H, W = conv_features.shape
        _, dist, _ = proto_head_forward(conv_features, self.prototype_vectors, None, self._layout(1), want_distances=True,
                                        epsilon=self.epsilon, activation="log")
        act = self.prototype_activation_function(dist).permute(0, 2, 3, 1).reshape(B * H * W, -1)
        logits = wide_linear(act, self.last_layer.weight).reshape(B, H, W, -1)
        if return_activations and not return_distances:
            return logits, act
        if return_activations and return_distances:
            return logits, dist, act
        return logits, dist

    def prune_prototypes(self, prototypes_to_prune: List[int]):
        """Drop prototype rows and re-pack the scale table (model_multiscale.py:400-432)."""
        keep = self._prune_bank(prototypes_to_prune)
        self.last_layer.in_features = self.num_prototypes
        self.last_layer.out_features = self.num_classes
        self.last_layer.weight.data = self.last_layer.weight.data[:, keep]

    def set_last_layer_incorrect_connection(self, incorrect_strength: float):
        """+1 own class / incorrect_strength elsewhere (model_multiscale.py:449-464)."""
        pos = torch.t(self.prototype_class_identity).to(self.last_layer.weight.device)
        self.last_layer.weight.data.copy_(1 * pos + incorrect_strength * (1 - pos))
        from .functional import invalidate_pack_cache

        invalidate_pack_cache()

    def _initialize_weights(self):
        for m in self.add_on_layers.modules():  # :466-476
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self.set_last_layer_incorrect_connection(incorrect_strength=-0.5)

    def __repr__(self):
        return (
            "PPNet(\n\tfeatures: {},\n\timg_size: {},\n\tprototype_shape: {},\n\tproto_layer_rf_info: {},\n"
            "\tnum_classes: {},\n\tepsilon: {}\n)"
        ).format(self.features, self.img_size, self.prototype_shape, self.proto_layer_rf_info, self.num_classes, self.epsilon)


def construct_PPNet(
    features: nn.Module,
    img_size: int = 224,
    prototype_shape: Tuple[int, int, int, int] = (2000, 512, 1, 1),
    num_classes: int = 200,
    prototype_activation_function: str = "log",
    add_on_layers_type: str = "bottleneck",
    scale_head_type: Optional[str] = None,
    **kwargs,
) -> PPNetMultiScale:
    """Factory with the reference's argument meaning (model_multiscale.py:480-515); the backbone is passed
    in as a module because the reference's backbone zoo (absent submodule, URL downloads) is out of scope."""
    return PPNetMultiScale(
        features=features, img_size=img_size, prototype_shape=prototype_shape, proto_layer_rf_info=[],
        num_classes=num_classes, init_weights=True, prototype_activation_function=prototype_activation_function,
        add_on_layers_type=add_on_layers_type, scale_head_type=scale_head_type, **kwargs,
    )
