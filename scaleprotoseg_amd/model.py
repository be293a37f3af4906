"""Single-scale ``PPNet`` (ProtoSeg baseline) as the S = 1 case of the multi-scale module.

Mirrors segmentation/model/model.py:73-463 for ``patch_classification=True`` (the only mode any config
uses).  Differences from PPNetMultiScale that callers can observe and that are kept: the class table has
no scale dimension (model.py:108-110), ``num_prototypes_per_class`` exists, and
``forward_from_conv_features(..., return_activations=True)`` returns ``(logits, activations)`` whatever
``return_distances`` says (model.py:357-360).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch.nn as nn

from .model_multiscale import PPNetMultiScale


class PPNet(PPNetMultiScale):
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
    ):
        assert prototype_shape[0] % num_classes == 0  # model.py:112
        super().__init__(
            features, img_size, prototype_shape, proto_layer_rf_info, num_classes, init_weights=init_weights,
            prototype_activation_function=prototype_activation_function, add_on_layers_type=add_on_layers_type,
            bottleneck_stride=bottleneck_stride, patch_classification=patch_classification, num_scales=1,
        )
        self.num_prototypes_per_class = self.num_prototypes // self.num_classes

    def _l2_convolution(self, x):
        return self._scale_l2_convolution(x)  # model.py:250-268

    def forward_from_conv_features(self, conv_features, return_activations=False, return_distances=False, **extensions):
        if isinstance(conv_features, list):
            return [self.forward_from_conv_features(c) for c in conv_features]
        if not (hasattr(self, "patch_classification") and self.patch_classification):
            # ProtoPNet global-min-pool branch (model.py:331-344): unused by every config, not built
            raise NotImplementedError("PPNet without patch_classification is outside the hot path")
        return super().forward_from_conv_features(conv_features, return_activations=return_activations,
                                                  return_distances=False, **extensions)

    def forward_with_features(self, x, **kwargs):
        conv = self.conv_features(x)  # model.py:317-326
        if isinstance(conv, list):
            res = [self.forward_from_conv_features(c, **kwargs) for c in conv]
            return [(r[0], r[1], c) for r, c in zip(res, conv)]
        logits, distances = self.forward_from_conv_features(conv, **kwargs)
        return logits, distances, conv
