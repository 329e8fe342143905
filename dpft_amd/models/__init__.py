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
    need ``weights_only=False`` on torch >= 2.6."""
    filename = os.path.splitext(os.path.basename(checkpoint))[0]
    timestamp, _, epoch = filename.split("_")
    return torch.load(checkpoint, weights_only=False), int(epoch), timestamp
