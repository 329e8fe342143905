"""Data-agnostic static query reference points.

Mirror of ``src/dprt/models/queries/data_agnostic.py`` (DataAgnosticStaticQueries :13-172) with
the ``spher2cart`` transformation of ``src/dprt/models/utils/transformations.py:212-281``.  The
grid is a constant of the configuration: it is evaluated once per (device, dtype) with the
reference's exact op sequence (linspace -> min-max scaling -> meshgrid('ij') -> spher2cart) and
then only expanded to the batch size.
"""
from __future__ import annotations

from collections import OrderedDict
from functools import partial
from typing import Any, Callable, Dict, List, Sequence, Union

import torch
from torch import nn

from dpft_amd.models.utils.transformations import build_transformation


class DataAgnosticStaticQueries(nn.Module):
    def __init__(self, resolution: List[int] = None, minimum: List[float] = None, maximum: List[float] = None,
                 transformation: nn.Module = None, distribution: Union[str, List[str]] = None, **kwargs):
        super().__init__()
        self.resolution = resolution if resolution is not None else []
        self.minimum = minimum if minimum is not None else []
        self.maximum = maximum if maximum is not None else []
        self.transformation = transformation if transformation is not None else nn.Identity()
        if distribution is None:
            self.distribution = ["linear"] * len(self.resolution)
        elif isinstance(distribution, (list, tuple)):
            self.distribution = distribution
        else:
            self.distribution = [distribution] * len(self.resolution)
        assert len(self.resolution) == len(self.minimum) == len(self.maximum) == len(self.distribution)
        self._dist_fns: List[Callable] = [
            getattr(torch, d) if d != "linear" else partial(torch.mul, 1) for d in self.distribution]
        self._cache = {}

    @classmethod
    def from_config(cls, config: Dict[str, Any]):
        return cls(config["resolution"], config["minimum"], config["maximum"],
                   transformation=build_transformation(config.get("transformation")),
                   distribution=config.get("distribution"))

    @staticmethod
    def _first(inp):
        # batch size / dtype / device come from the FIRST entry (data_agnostic.py:70-99)
        if isinstance(inp, torch.Tensor):
            return inp
        if isinstance(inp, dict):
            return inp[list(inp.keys())[0]]
        return inp[0]

    @staticmethod
    def _min_max_scaling(x: torch.Tensor, mi: float, ma: float) -> torch.Tensor:
        den = torch.max(x) - torch.min(x)
        if torch.isclose(den, torch.zeros_like(den)):
            den = 1.0
        return (x - torch.min(x)) / den * (ma - mi) + mi

    def _grid(self, dtype, device) -> torch.Tensor:
        key = (dtype, str(device))
        g = self._cache.get(key)
        if g is None:
            qs = [torch.linspace(0.0, 1.0, r, dtype=dtype) for r in self.resolution]
            qs = [fn(q) for q, fn in zip(qs, self._dist_fns)]
            qs = [self._min_max_scaling(q, mi, ma) for q, mi, ma in zip(qs, self.minimum, self.maximum)]
            qs = torch.meshgrid(*tuple(qs), indexing="ij")
            g = torch.stack([torch.flatten(q) for q in qs], dim=-1)
            g = self.transformation(g.unsqueeze(0))[0].contiguous().to(device)
            self._cache[key] = g
        return g

    def __getstate__(self):          # caches are rebuilt on demand (torch.save(model), deepcopy)
        st = self.__dict__.copy()
        st.pop("_batched", None)
        st["_cache"] = {}
        return st

    def forward(self, batch: Union[torch.Tensor, Sequence[torch.Tensor], Dict[str, torch.Tensor]]):
        first = self._first(batch)
        g = self._grid(first.dtype, first.device)
        # the batch of query centres is a constant of (B, dtype, device): built once, handed out read-only (the heads write
        # their refined centres into new tensors) -- a captured decoder graph can then take it as a static input as it is
        key = (first.shape[0], first.dtype, str(first.device))
        cache = self.__dict__.setdefault("_batched", {})
        c = cache.get(key)
        if c is None or c.data_ptr() == 0:
            c = cache[key] = g.unsqueeze(0).repeat(first.shape[0], 1, 1)
        return OrderedDict({"center": c})


def build_data_agnostic_query(name: str, *args, **kwargs):
    if "static" in name.lower() or "linear" in name.lower():
        return DataAgnosticStaticQueries.from_config(*args, **kwargs)
    raise ValueError(f"unknown data agnostic querent {name!r}")
