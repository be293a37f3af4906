"""scaleprotoseg_amd — MI355X-native prototype-distance hot path of ScaleProtoSeg.

Public surface mirrors the reference's module/function names for this path:
    PPNetMultiScale, construct_PPNet                (segmentation/model/model_multiscale.py)
    PPNetMultiScaleGroup, construct_PPNet_Group     (segmentation/model/model_multiscale_group.py)
    PPNet                                           (segmentation/model/model.py, S = 1)
    compute_distances, min_across_dataset, global_min, push_prototypes_multiscale
                                                    (segmentation/push_multiscale_optimization.py)
    projection_simplex_sort, resize_label           (segmentation/utils.py, segmentation/data/dataset.py)
    KLDLoss, KLDLossGroup                           (segmentation/model/loss.py; KLDLoss also takes the class-gathered
                                                     ClassDistances of forward_from_conv_features(target_labels=...))
Arithmetic runs in libspx_hip.so (hand-written gfx950 HIP); there is no CPU fallback.
"""
from ._lib import SpxError, load as load_library  # noqa: F401
from .functional import (  # noqa: F401
    BankLayout,
    ClassGather,
    FusedCrossEntropy,
    cross_entropy_from_logits,
    argmin_over_images,
    class_gather_table,
    proto_head_forward,
    push_masked_argmin,
    push_min_from_features,
    upsample_argext,
)
from .checkpoint import export_state, import_state, load_reference_state_dict  # noqa: F401
from .loss import ClassDistances, KLDLoss, KLDLossGroup, PixelWiseCrossEntropyLoss  # noqa: F401
from .model import PPNet  # noqa: F401
from .model_multiscale import PPNetMultiScale, construct_PPNet  # noqa: F401
from .model_multiscale_group import PPNetMultiScaleGroup, construct_PPNet_Group  # noqa: F401
from .push import (  # noqa: F401
    compute_distances,
    global_min,
    min_across_dataset,
    push_prototypes_multiscale,
)
from .utils import projection_simplex_sort, resize_label  # noqa: F401

__version__ = "0.1.0"
