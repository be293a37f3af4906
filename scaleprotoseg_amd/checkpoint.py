"""On-disk format helpers: load reference ``state_dict``s into the drop-in modules.

The reference saves whole-module pickles (``torch.save(ppnet)``, loaded with ``torch.load`` in
segmentation/finetune_wandb_group.py:74-80), which bind to its class path.  What travels between the two code bases
is the ``state_dict`` (same keys) plus the two attributes that are NOT serialised in it and that the
push's de-duplication changes: ``prototype_class_identity`` and ``scale_num_prototypes``.  The push writes the kept
prototype indices to ``unique_prototypes.json`` (segmentation/push_multiscale_optimization.py:327-335); from that
list this module rebuilds both attributes exactly as ``prune_prototypes`` does (model_multiscale.py:400-432).
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, List, Optional, Union

import torch
import torch.nn as nn


def _kept_indices(unique_prototypes: Union[None, str, os.PathLike, Iterable[int]]) -> Optional[List[int]]:
    if unique_prototypes is None:
        return None
    if isinstance(unique_prototypes, (str, os.PathLike)):
        with open(unique_prototypes) as fp:
            unique_prototypes = json.load(fp)
    return sorted(int(i) for i in unique_prototypes)


def load_reference_state_dict(
    net: nn.Module,
    state_dict: Dict[str, torch.Tensor],
    unique_prototypes: Union[None, str, os.PathLike, Iterable[int]] = None,
    strict: bool = False,
):
    """Load a reference ``state_dict`` into a freshly constructed drop-in module.

    ``net`` must have been built with the ORIGINAL prototype shape of the run (the gin ``prototype_shape``).  If the
    checkpoint was written after a push, pass the kept indices (the list in ``unique_prototypes.json`` or its path):
    the module's bank, ``ones``, ``last_layer`` columns, ``prototype_class_identity`` and ``scale_num_prototypes``
    are pruned to them first, so shapes and class tables match the checkpoint.  Returns what
    ``nn.Module.load_state_dict`` returns (the group phase has no ``last_layer``, the prototype phase no
    ``group_projection`` - hence ``strict=False`` by default, as in finetune_wandb_group.py:76)."""
    keep = _kept_indices(unique_prototypes)
    P_ckpt = int(state_dict["prototype_vectors"].shape[0]) if "prototype_vectors" in state_dict else None
    if keep is not None:
        P0 = net.num_prototypes
        if keep and (keep[0] < 0 or keep[-1] >= P0):
            raise ValueError(f"unique_prototypes index outside the module's bank of {P0} prototypes")
        drop = sorted(set(range(P0)) - set(keep))
        if drop:
            net.prune_prototypes(drop)
    if P_ckpt is not None and P_ckpt != net.num_prototypes:
        raise ValueError(
            f"checkpoint holds {P_ckpt} prototypes, the module {net.num_prototypes}: pass the run's "
            "unique_prototypes.json (kept indices after the push) so the class tables can be rebuilt"
        )
    if hasattr(net, "group_projection") and hasattr(net, "_initialize_groups") and keep is not None:
        net._initialize_groups()        # per-class projection shapes follow the pruned class table
    return net.load_state_dict(state_dict, strict=strict)


def export_state(net: nn.Module) -> Dict[str, object]:
    """``state_dict`` plus the non-serialised attributes, as one picklable dict of tensors and plain containers."""
    out: Dict[str, object] = {"state_dict": {k: v.detach().cpu() for k, v in net.state_dict().items()}}
    out["prototype_class_identity"] = net.prototype_class_identity.detach().cpu()
    out["scale_num_prototypes"] = {int(s): tuple(int(v) for v in r) for s, r in net.scale_num_prototypes.items()}
    return out


def import_state(net: nn.Module, blob: Dict[str, object], strict: bool = False):
    """Inverse of ``export_state`` for a module built with the original prototype shape."""
    ident = blob["prototype_class_identity"]
    ranges = blob["scale_num_prototypes"]
    P = int(ident.shape[0])
    if P != net.num_prototypes:
        # shrink to the stored bank: any P rows do (load_state_dict overwrites them), the tables come from the blob
        net.prune_prototypes(list(range(P, net.num_prototypes)))
    net.prototype_class_identity = ident.clone()
    net.scale_num_prototypes = {int(s): tuple(int(v) for v in r) for s, r in ranges.items()}
    if hasattr(net, "group_projection") and hasattr(net, "_initialize_groups"):
        net._initialize_groups()
    return net.load_state_dict(blob["state_dict"], strict=strict)
