"""Model factory -- mirror of ``src/dprt/models/__init__.py:10-18``."""
import os
from typing import Tuple

import torch

from dpft_amd.models.dprt import build_dprt


def build(model: str, *args, **kwargs):
    if model == "dprt":
        return build_dprt(*args, **kwargs)
    raise ValueError(f"unknown model {model!r}")


def load(checkpoint: str, *args, **kwargs) -> Tuple[torch.nn.Module, int, str]:
    """``<timestamp>_checkpoint_<epoch>.pt`` -> (module, epoch, timestamp).  Whole-module pickles
    need ``weights_only=False`` on torch >= 2.6.

    Accepts this package's own ``torch.save(model)`` files AND the reference's (classes of ``dprt.*`` / ``torchvision.*``,
    e.g. the published checkpoints): those are rebuilt as ``dpft_amd`` modules from the pickled tensors and
    hyper-parameters (dpft_amd/models/checkpoint.py)."""
    from dpft_amd.models import checkpoint as _ck
    filename = os.path.splitext(os.path.basename(checkpoint))[0]
    timestamp, _, epoch = filename.split("_")
    obj = _ck.read_foreign(checkpoint)
    if isinstance(obj, _ck.ForeignModule):
        return _ck.load_reference_checkpoint(obj, kwargs.get("config")), int(epoch), timestamp
    return obj, int(epoch), timestamp
