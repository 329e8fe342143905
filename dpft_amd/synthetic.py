"""Seeded synthetic K-Radar-shaped batches -- there is no dataset offline.

Shapes/contracts follow the reference's dataset code: camera frame 1280x720 resized so that the short
side is ``image_size`` (src/dprt/datasets/kradar/dataset.py:319-341), radar RA map 256x107x6 and EA
map 37x107x6 scaled to [0,255] (:295-317), ``X_shape`` recorded before the resize (:164-169),
camera ``label_to_camera_mono_t`` all zero (:204-205), radar P matrices 3x4 (:259-293), labels as a
list of per-sample dicts (src/dprt/datasets/loader.py:10-34).
Tensors are generated on the CPU with a seeded generator and then moved, so CPU oracle and GPU path
see identical bits.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

INPUT_SHAPES = {            # (H, W, C) as fed to the model, and the recorded original shape
    "camera_mono": ((512, 910, 3), (720, 1280, 3)),
    "radar_bev": ((256, 107, 6), (256, 107, 6)),
    "radar_front": ((37, 107, 6), (37, 107, 6)),
}


def make_batch(inputs: List[str], batch_size: int, seed: int = 42, device="cpu",
               shapes: Dict[str, Tuple[int, int, int]] = None) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    B = batch_size
    batch: Dict[str, torch.Tensor] = {}
    for name in inputs:
        (H, W, C), orig = INPUT_SHAPES[name]
        if shapes is not None and name in shapes:
            H, W, C = shapes[name]
        batch[name] = torch.rand(B, H, W, C, generator=g) * 255.0
        batch[f"{name}_shape"] = torch.tensor([list(orig)] * B, dtype=torch.int64)
        if name == "camera_mono":
            f, cx, cy = 560.0, 640.0, 360.0
            p = torch.tensor([[cx, -f, 0, 0], [cy, 0, -f, 0], [1, 0, 0, 0], [0, 0, 0, 1.0]]).repeat(B, 1, 1)
            p = p + (torch.rand(B, 4, 4, generator=g) * 10 - 5) * (p.abs() > 1.5)
            batch[f"label_to_{name}_t"] = torch.zeros(B, 4, 4)
            batch[f"label_to_{name}_p"] = p
        else:
            t = torch.eye(4).repeat(B, 1, 1)
            t[:, 0, 3] = torch.rand(B, generator=g) * 6 - 3
            t[:, 1, 3] = torch.rand(B, generator=g) * 2 - 1
            if name == "radar_bev":
                p = torch.tensor([[0, -1, 0, 53], [256 / 118.03710938, 0, 0, 0], [0, 0, 0, 1.0]])
            else:
                p = torch.tensor([[0, -1, 0, 53], [0, 0, 1, 18], [0, 0, 0, 1.0]])
            batch[f"label_to_{name}_t"] = t
            batch[f"label_to_{name}_p"] = p.repeat(B, 1, 1)
    return {k: v.to(device) for k, v in batch.items()}


def make_labels(batch_size: int, seed: int = 42, device="cpu", max_boxes: int = 8) -> List[Dict[str, torch.Tensor]]:
    """Per-sample label dicts: gt_center (M,3), gt_size (M,3), gt_angle (M,2)=[sin,cos], gt_class (M,2)=[0,1]."""
    g = torch.Generator().manual_seed(seed + 1000)
    labels = []
    for _ in range(batch_size):
        M = int(torch.randint(1, max_boxes + 1, (1,), generator=g))
        u = torch.rand(M, 3, generator=g)
        center = torch.stack((5 + u[:, 0] * 65, -6 + u[:, 1] * 12, -1.5 + u[:, 2] * 3.5), -1)
        s = torch.rand(M, 3, generator=g)
        size = torch.stack((3.5 + s[:, 0] * 1.5, 1.6 + s[:, 1] * 0.6, 1.4 + s[:, 2] * 0.6), -1)
        yaw = (torch.rand(M, generator=g) * 2 - 1) * 3.14159
        labels.append({
            "gt_center": center.to(device), "gt_size": size.to(device),
            "gt_angle": torch.stack((torch.sin(yaw), torch.cos(yaw)), -1).to(device),
            "gt_class": torch.tensor([[0.0, 1.0]]).repeat(M, 1).to(device),
        })
    return labels
