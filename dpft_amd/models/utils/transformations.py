"""Coordinate transformations on the hot path (``src/dprt/models/utils/transformations.py``):
``cart2spher`` (:71-120, used by IMPFusion.get_reference_points) and ``Spher2Cart`` (:212-281,
used by the querent).  The unused polar variants are out of scope."""
from __future__ import annotations

from typing import Tuple

import torch
from torch import nn


def cart2spher(x, y, z, degrees: bool = True) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    r = torch.linalg.norm(torch.dstack((x, y, z)), dim=-1).reshape_as(x)
    phi = torch.arctan2(y, x)
    mask = r != 0
    c = torch.where(mask, z / torch.where(mask, r, torch.ones_like(r)), torch.zeros_like(z))
    roh = torch.arcsin(c)
    if degrees:
        phi = torch.rad2deg(phi)
        roh = torch.rad2deg(roh)
    return r, phi, roh


def spher2cart(r, phi, roh, degrees: bool = True):
    if degrees:
        phi, roh = torch.deg2rad(phi), torch.deg2rad(roh)
    return r * torch.cos(phi) * torch.cos(roh), r * torch.sin(phi) * torch.cos(roh), r * torch.sin(roh)


class Spher2Cart(nn.Module):
    def __init__(self, dim: int = -1, degrees: bool = True, **kwargs):
        super().__init__()
        self.dim, self.degrees = dim, degrees

    def forward(self, batch: torch.Tensor):
        return torch.cat(spher2cart(*batch.split(1, self.dim), self.degrees), self.dim)


class Cart2Spher(nn.Module):
    def __init__(self, dim: int = -1, degrees: bool = True, **kwargs):
        super().__init__()
        self.dim, self.degrees = dim, degrees

    def forward(self, batch: torch.Tensor):
        return torch.cat(cart2spher(*batch.split(1, self.dim), self.degrees), self.dim)


def build_transformation(name: str, *args, **kwargs):
    if name is None:
        return None
    if "spher2cart" in name.lower():
        return Spher2Cart(*args, **kwargs)
    if "cart2spher" in name.lower():
        return Cart2Spher(*args, **kwargs)
    raise ValueError(f"transformation {name!r} is outside the dpft_amd hot path")
