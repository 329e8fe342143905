"""Configuration dictionaries with the reference's schema (``config/*.json`` of TUMFTM/DPFT:
sections computing / data / train / model / evaluate).  They are generated here instead of being
shipped as files; the only deliberate difference from the reference's JSONs is
``model.backbones.*.weights = ""`` (the torchvision ``IMAGENET1K_V2`` enum needs a download).
Any reference JSON can be passed unchanged through ``load_config(path)`` as well.
"""
from __future__ import annotations

import copy
import json
import os
from typing import Any, Dict, List

_VIEWS = {
    "camera_mono": dict(backbone="ResNet101", in_channels=None, raw_channels=3),
    "radar_bev": dict(backbone="ResNet50", in_channels=6, raw_channels=6),
    "radar_front": dict(backbone="ResNet50", in_channels=6, raw_channels=6),
}
_NAMES = {
    "kradar": ["camera_mono", "radar_bev", "radar_front"],
    "kradar_camera_mono": ["camera_mono"],
    "kradar_radar_bev": ["radar_bev"],
    "kradar_radar_front": ["radar_front"],
    "kradar_radar": ["radar_bev", "radar_front"],
}


def make_config(inputs: List[str], weights: str = "") -> Dict[str, Any]:
    n = len(inputs)
    backbones, necks, embeddings = {}, {}, {}
    for v in inputs:
        spec = _VIEWS[v]
        bb = {"name": spec["backbone"], "weights": weights}
        if spec["in_channels"] is not None:
            bb["in_channels"] = spec["in_channels"]
        bb.update({"multi_scale": 4, "norm_layer": "BatchNorm2d"})
        backbones[v] = bb
        necks[v] = {"name": "FPN", "in_channels_list": [spec["raw_channels"], 256, 512, 1024, 2048],
                    "out_channels": 16}
        embeddings[v] = {"name": "sinusoidal_embedding", "num_feats": 16, "n_levels": 5, "normalize": True}
    data = {
        "revision": "v2", "image_size": 512, "num_classes": 2,
        "categories": {"Sedan": 0, "Bus or Truck": -1, "Motorcycle": -1, "Bicycle": -1, "Bicycle Group": -1,
                       "Pedestrian": -1, "Pedestrian Group": -1, "Background": -1},
        "fov": {"x": [0.0, 72.0], "y": [-6.4, 6.4], "z": [-2.0, 6.0], "azimuth": [-50, 50]},
    }
    return {
        "dataset": "kradar",
        "computing": {"dtype": "float32", "seed": 42, "workers": 16, "device": "cuda"},
        "data": data,
        "train": {
            "batch_size": 4, "shuffle": True, "epochs": 200, "logging": "epoch",
            "optimizer": {"name": "AdamW", "lr": 0.0001},
            "anassigner": "HungarianAnassigner", "criterion": "SetCriterion",
            "losses": {"class": "FocalLoss", "center": "L1Loss", "size": "L1Loss", "angle": "L1Loss"},
            "loss_inputs": {"class": ["class"], "center": ["center"], "size": ["size"], "angle": ["angle"]},
            "loss_weights": {"total_class": 1.0, "object_class": 0.0, "center": 1.0, "size": 1.0, "angle": 1.0},
            "scheduler": {"name": "ConstantLR", "factor": 1.0},
        },
        "model": {
            "name": "dprt", "inputs": list(inputs), "skiplinks": {v: True for v in inputs},
            "backbones": backbones, "necks": necks, "embeddings": embeddings,
            "querent": {"name": "data_agnostic_static_querent", "transformation": "spher2cart",
                        "resolution": [20, 20, 1], "minimum": [4, -50, 0], "maximum": [72, 50, 0]},
            "fuser": {"name": "IMPFusion", "i_iter": 4, "m_views": n, "d_model": 16, "d_ffn": 32,
                      "n_queries": 400, "n_levels": [5] * n, "n_heads": [8] * n, "n_points": [4] * n,
                      "norm": True, "dropout": 0.1, "reduction": "linear", "activation": "Mish"},
            "head": {"name": "linear_detection_head", "in_channels": 16, "num_classes": 2,
                     "num_reg_layers": 3, "num_cls_layers": 3},
        },
        "evaluate": {"logging": "epoch", "metrics": {"mAP": "mAP3D", "mGIoU": "mGIoU3D"},
                     "exporter": {"name": "kradar"}},
    }


def load_config(name_or_path: str, offline: bool = False, weight_files: Dict[str, str] = None) -> Dict[str, Any]:
    """``load_config('kradar')`` -> built-in equivalent of config/kradar.json; a path loads that JSON unchanged
    (mirror of src/dprt/utils/config.py:8-20).

    Backbone ``weights`` entries that are torchvision enums (``IMAGENET1K_V2`` ...) cannot be downloaded here.  Nothing
    is blanked silently: ``weight_files={"ResNet101": "/path/r101.pt", "IMAGENET1K_V2": ...}`` maps a backbone name (or
    an enum) to a local state-dict file; ``offline=True`` is the explicit opt-in to random init and warns per backbone;
    otherwise the enum stays in the config and ``Backbone`` raises when the model is built."""
    if name_or_path in _NAMES:
        return make_config(_NAMES[name_or_path])
    with open(name_or_path, "r") as f:
        cfg = json.load(f)
    for view, bb in cfg.get("model", {}).get("backbones", {}).items():
        w = bb.get("weights")
        if not w or os.path.exists(w):
            continue
        local = (weight_files or {}).get(bb.get("name")) or (weight_files or {}).get(w)
        if local:
            bb["weights"] = local
        elif offline:
            import warnings
            warnings.warn(f"load_config(offline=True): backbone {view!r} ({bb.get('name')}) weights {w!r} are not "
                          "available offline -> RANDOM initialisation (the reference starts from pretrained weights)")
            bb["weights"] = ""
    return cfg


def available() -> List[str]:
    return sorted(_NAMES)
