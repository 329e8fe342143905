"""Learning-rate schedules, built from ``config['train']['scheduler']`` like src/dprt/training/scheduler.py:8-36: a
``name`` from ``torch.optim.lr_scheduler`` plus its keyword arguments; the two composite kinds (``ChainedScheduler``,
``SequentialLR``) take a list of such dicts under ``schedulers``.  Returns a factory ``optimizer -> scheduler`` (the
reference binds the optimizer late too: trainer.py:236).  ``FusedAdamW`` reads ``group['lr']`` on every launch, so any
torch scheduler drives it unchanged."""
from __future__ import annotations

from typing import Any, Callable, Dict

import torch

_COMPOSITE = {"ChainedScheduler", "SequentialLR"}


def _make(optimizer, spec: Dict[str, Any]):
    spec = dict(spec)                        # the caller's config stays intact (the reference pops from it)
    name = spec.pop("name")
    cls = getattr(torch.optim.lr_scheduler, name)
    if name in _COMPOSITE:
        members = [_make(optimizer, s) for s in spec.pop("schedulers")]
        if name == "SequentialLR":
            return cls(optimizer, members, **spec)
        return cls(members, **spec)
    return cls(optimizer, **spec)


def build_scheduler(name: str, *args, **kwargs) -> Callable:
    if args:
        raise TypeError("build_scheduler takes the scheduler's options as keyword arguments")
    return lambda optimizer: _make(optimizer, dict(kwargs, name=name))
