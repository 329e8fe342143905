"""Sinusoidal positional embedding, MI355X-native.

Mirror of ``src/dprt/models/embeddings/sinusoidal.py`` (SinusoidalEmbedding :11-110,
MultiLevelSinusoidalEmbedding :113-153).  The embedding is a pure function of (H, W): the two
(W,C) / (H,C) tables are computed once per shape on the host (fp32, same op order as the
reference: cumsum from 1, /(last+eps)*scale, /dim_t, interleaved sin/cos) and added IN PLACE by
one HIP kernel (``x += pos_x; x += pos_y``, the reference's two fp32 adds, :107-108).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Any, Dict

import torch
from torch import nn

from dpft_amd.hip import ops


def _axis_table(n: int, num_feats: int, temperature: float, normalize: bool, scale: float, eps: float,
                offset: float) -> torch.Tensor:
    embed = torch.arange(1, n + 1, dtype=torch.float32)
    if normalize:
        embed = (embed + offset) / (embed[-1:] + eps) * scale
    dim_t = torch.arange(num_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_feats)
    pos = embed[:, None] / dim_t
    return torch.stack((pos[:, 0::2].sin(), pos[:, 1::2].cos()), dim=2).view(n, -1).contiguous()


class _AddPosFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos_x, pos_y):
        ctx.mark_dirty(x)
        ops.add_pos_(x, pos_x, pos_y)
        return x

    @staticmethod
    def backward(ctx, g):
        return g, None, None


class SinusoidalEmbedding(nn.Module):
    def __init__(self, num_feats: int, temperature: int = 10000, normalize: bool = False,
                 scale: float = 2 * math.pi, eps: float = 1e-6, offset: float = 0., **kwargs):
        super().__init__()
        if normalize:
            assert isinstance(scale, (float, int)), "when normalize is set, scale should be float or int"
        self.num_feats, self.temperature, self.normalize = num_feats, temperature, normalize
        self.scale, self.eps, self.offset = scale, eps, offset
        self._tables = {}

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "SinusoidalEmbedding":
        return cls(**config)

    def __getstate__(self):          # the per-(H, W, device) tables are device tensors: rebuilt on demand
        st = self.__dict__.copy()
        st["_tables"] = {}
        return st

    def tables(self, H: int, W: int, device):
        key = (H, W, str(device))
        t = self._tables.get(key)
        if t is None:
            args = (self.num_feats, float(self.temperature), self.normalize, self.scale, self.eps, self.offset)
            t = (_axis_table(W, *args).to(device), _axis_table(H, *args).to(device))
            self._tables[key] = t
        return t

    def forward(self, batch: torch.Tensor) -> torch.Tensor:
        """(B,H,W,C) -> same tensor, embedded in place."""
        B, H, W, C = batch.shape
        if C != self.num_feats:
            raise ValueError(f"SinusoidalEmbedding: C={C} != num_feats={self.num_feats}")
        pos_x, pos_y = self.tables(H, W, batch.device)
        if not batch.is_contiguous():
            raise ValueError("SinusoidalEmbedding expects a contiguous NHWC tensor")
        return _AddPosFn.apply(batch, pos_x, pos_y)


class MultiLevelSinusoidalEmbedding(nn.Module):
    def __init__(self, n_levels: int = 1, **kwargs):
        super().__init__()
        self.n_levels = n_levels
        self.embedding_layers = nn.ModuleDict(
            {"embedding" + str(i): SinusoidalEmbedding(**kwargs) for i in range(n_levels)})

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "MultiLevelSinusoidalEmbedding":
        return cls(**config)

    def forward(self, batches: "OrderedDict[str, torch.Tensor]") -> "OrderedDict[str, torch.Tensor]":
        it = zip(batches.items(), self.embedding_layers.values())
        return OrderedDict({k: layer(b) for (k, b), layer in it})

    def level_tables(self, batches: "OrderedDict[str, torch.Tensor]", channels: int):
        """Per level the (pos_x (W,C), pos_y (H,C)) tables ``forward`` would add to NHWC tensors of these maps' sizes with ``channels``
        channels -- for a producer that adds them itself (necks/fpn.py: the output conv's epilogue); None where that cannot replace
        ``forward`` exactly."""
        levels, layers = list(batches.values()), list(self.embedding_layers.values())
        if len(levels) != len(layers) or any(t.dim() != 4 or channels != l.num_feats for t, l in zip(levels, layers)):
            return None
        return [l.tables(t.shape[1], t.shape[2], t.device) for t, l in zip(levels, layers)]


def build_sinusoidal_embedding(*args, **kwargs) -> nn.Module:
    return MultiLevelSinusoidalEmbedding.from_config(*args, **kwargs)
