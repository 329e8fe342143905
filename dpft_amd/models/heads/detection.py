"""Linear detection head (``src/dprt/models/heads/detection.py:149-275``): four bias-free MLP
branches (center/size/angle/class), activations Identity/ReLU/Tanh/Identity, ``center += ref``.
Identical module tree => identical state-dict names (``layers.{center,size,angle,class}_head.{0,3,6}``)."""
from __future__ import annotations

from collections import OrderedDict
from typing import Any, Dict, Optional

import torch
from torch import nn


class LinearDetectionHead(nn.Module):
    def __init__(self, in_channels: int, num_classes: int, num_reg_layers: int = 1, num_cls_layers: int = 1,
                 bias: Optional[bool] = False, dropout: float = 0.0, channels_last: Optional[bool] = True,
                 **kwargs) -> None:
        super().__init__()
        self.in_channels, self.num_classes = in_channels, num_classes
        self.num_reg_layers, self.num_cls_layers = num_reg_layers, num_cls_layers
        self.bias, self.dropout, self.channels_last = bias, dropout, channels_last
        self.activations = {"center": "Identity", "size": "ReLU", "angle": "Tanh", "class": "Identity"}
        self.layers = nn.ModuleDict({
            "center_head": self._branch(3, self.num_reg_layers),
            "size_head": self._branch(3, self.num_reg_layers),
            "angle_head": self._branch(2, self.num_reg_layers),
            "class_head": self._branch(self.num_classes, self.num_cls_layers),
        })
        self.activation_fn = nn.ModuleDict({k: getattr(nn, v)() for k, v in self.activations.items()})

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "LinearDetectionHead":
        return cls(config["in_channels"], config["num_classes"], config.get("num_reg_layers", 1),
                   config.get("num_cls_layers", 1), config.get("bias", False), config.get("dropout", 0.0),
                   config.get("channels_last", True))

    def _branch(self, out_channels: int, n_layers: int) -> nn.Module:
        layers = []
        for _ in range(n_layers - 1):
            layers += [nn.Linear(self.in_channels, self.in_channels, bias=self.bias), nn.ReLU(),
                       nn.Dropout(self.dropout)]
        layers.append(nn.Linear(self.in_channels, out_channels, bias=self.bias))
        return nn.Sequential(*layers)

    def forward(self, batch: torch.Tensor, ref: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        it = zip(self.activation_fn.items(), self.layers.values())
        out = OrderedDict({k: act(layer(batch)) for (k, act), layer in it})
        out["center"] = out["center"] + ref["center"][..., :3]        # detection.py:273
        return out


def build_detection_head(name: str, *args, **kwargs) -> nn.Module:
    if "linear" in name.lower():
        return LinearDetectionHead.from_config(*args, **kwargs)
    raise ValueError(f"head {name!r} is outside the dpft_amd hot path (linear_detection_head only)")
