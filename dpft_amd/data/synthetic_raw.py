"""Seeded synthetic samples at the RAW sizes the K-Radar dataset delivers before the online transforms (there is no
dataset offline): camera frame 720x1280x3 as decoded u8, radar RA map 256x107x6 / EA map 37x107x6 in dB (unscaled),
``X_shape`` of the raw frame, transformation / projection matrices and the detection label dict (SURVEY App. A;
src/dprt/datasets/kradar/dataset.py:120-181)."""
from __future__ import annotations

from typing import Dict, Tuple

import torch
from torch.utils.data import Dataset

from dpft_amd.synthetic import make_batch, make_labels

RAW = {"camera_mono": (720, 1280, 3), "radar_bev": (256, 107, 6), "radar_front": (37, 107, 6)}


class SyntheticRawDataset(Dataset):
    def __init__(self, n: int = 64, seed: int = 0, inputs=("camera_mono", "radar_bev", "radar_front"),
                 camera_u8: bool = True, raw_shapes: Dict[str, Tuple[int, int, int]] = None):
        self.n, self.seed, self.inputs, self.camera_u8 = n, seed, tuple(inputs), camera_u8
        self.raw = dict(RAW, **(raw_shapes or {}))

    def __len__(self) -> int:
        return self.n

    def __getitem__(self, index: int):
        if not 0 <= index < self.n:
            raise IndexError(index)
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + index)
        meta = make_batch(list(self.inputs), 1, seed=self.seed * 7919 + index)      # matrices + shapes of one sample
        sample: Dict[str, torch.Tensor] = {}
        for name in self.inputs:
            H, W, C = self.raw[name]
            if name.startswith("camera"):
                img = torch.randint(0, 256, (H, W, C), generator=g, dtype=torch.uint8)
                sample[name] = img if self.camera_u8 else img.float()
            else:
                sample[name] = 60.0 + torch.rand(H, W, C, generator=g) * 180.0       # dB, partly outside [100, 200]
            sample[f"{name}_shape"] = torch.tensor([H, W, C], dtype=torch.int64)
            sample[f"label_to_{name}_t"] = meta[f"label_to_{name}_t"][0]
            sample[f"label_to_{name}_p"] = meta[f"label_to_{name}_p"][0]
        label = make_labels(1, seed=self.seed * 31 + index)[0]
        return sample, label
