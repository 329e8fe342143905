"""Reader of the pre-processed K-Radar folder tree -- the real-data half of the input pipeline (SURVEY 8f NEXT-2).

Counterpart of ``KRadarDataset`` (src/dprt/datasets/kradar/dataset.py:19-118 constructor / split naming, :120-181
``__getitem__``, :183-257 transformation / projection / shape entries, :259-293 radar grid projections, :343-395 the
detection label and its field-of-view filter, :397-481 folder walk, :483-508 file loading, :510-536 modality dropout):
the same constructor arguments and config keys (``from_config(config)`` = ``computing | data``), the same folder layout

    <src>/<split>/<sequence>/<sample>/{mono.jpg, mono_info.npy, stereo.jpg, stereo_info.npy, ra.npy, ra_info.npy,
                                       ea.npy, ea_info.npy, os1.npy | os2.npy, labels.npy, description.npy}

the same sample dict keys in the same order and the same label dicts, so ``listed_collating`` / ``PrefetchLoader`` /
``GpuPreprocessor`` take its output unchanged.

What differs is WHERE the two arithmetic transforms run.  ``device_transforms=True`` (the default here) hands out what the
files hold -- the camera frame as decoded uint8 HWC, the radar maps in dB -- and leaves ``scale_radar_data`` and
``resize_image`` to the HIP kernels on the upload stream (dpft_amd/data/preprocess.py); a worker then only decodes and
copies.  ``device_transforms=False`` reproduces the reference's host-side sample (float frame, scaled radar maps; the
resize via ``F.interpolate``: torchvision is not a dependency) for comparisons.

JPEG / PNG decoding uses Pillow (the reference: ``torchvision.io.read_image``; both sit on libjpeg, neither is pinned by
the reference's tests).
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Sequence, Tuple, Union

import numpy as np
import torch
from torch.utils.data import Dataset

# Radar grid of the pre-processed maps (src/dprt/datasets/kradar/utils/radar_info.py): 107 azimuth bins, 37 elevation
# bins, 256 range bins up to 118.03710938 m; received power limits used by the scaling (:109,:113).
N_AZIMUTH, N_ELEVATION, N_RANGE, MAX_RANGE = 107, 37, 256, 118.03710938
MIN_POWER, MAX_POWER = 100.0, 200.0

_CAMERAS = (("M", "camera_mono", "mono"), ("S", "camera_stereo", "stereo"))
_RADARS = (("B", "radar_bev", "ra"), ("F", "radar_front", "ea"))
_IMAGE_EXT = (".jpg", ".png")


def read_image_hwc(path: str) -> torch.Tensor:
    """Decoded frame as a uint8 (H, W, C) tensor (grey-scale files get C = 1, like ``read_image``'s (C, H, W) moved last)."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ("RGB", "L"):
            im = im.convert("RGB")
        arr = np.array(im)                                   # (a writable copy: the tensor owns it)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(arr))


def ra_projection(dtype=torch.float32) -> torch.Tensor:
    """(u, v, 1) = P (r, phi, rho, 1): azimuth bin, range bin of the range-azimuth map (dataset.py:277-293)."""
    return torch.tensor([[0.0, -1.0, 0.0, (N_AZIMUTH - 1) / 2], [N_RANGE / MAX_RANGE, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]],
                        dtype=dtype)


def ea_projection(dtype=torch.float32) -> torch.Tensor:
    """Azimuth bin, elevation bin of the elevation-azimuth map (dataset.py:259-275)."""
    return torch.tensor([[0.0, -1.0, 0.0, (N_AZIMUTH - 1) / 2], [0.0, 0.0, 1.0, (N_ELEVATION - 1) / 2], [0.0, 0.0, 0.0, 1.0]],
                        dtype=dtype)


def detection_label(raw: torch.Tensor, num_classes: int, fov: Dict[str, Sequence[float]], dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Rows [x, y, z, theta, l, w, h, category, id] -> centre / size / (sin, cos) / one-hot (category + 1: slot 0 is the
    ignore class), restricted to boxes whose centre lies strictly inside the configured field of view in x, y, z and
    azimuth (degrees) -- dataset.py:343-395."""
    center, size = raw[:, 0:3], raw[:, 4:7]
    theta = raw[:, 3:4]
    label = {"gt_center": center, "gt_size": size, "gt_angle": torch.cat((torch.sin(theta), torch.cos(theta)), dim=-1),
             "gt_class": torch.nn.functional.one_hot(raw[:, 7].long() + 1, num_classes).to(dtype)}
    keep = torch.ones(raw.shape[0], dtype=torch.bool)
    azimuth = torch.rad2deg(torch.atan2(center[:, 1], center[:, 0]))
    for name, value in (("x", center[:, 0]), ("y", center[:, 1]), ("z", center[:, 2]), ("azimuth", azimuth)):
        if name in fov:
            lo, hi = fov[name]
            keep &= (value > lo) & (value < hi)
    return {k: v[keep] for k, v in label.items()}


class KRadarFolderDataset(Dataset):
    def __init__(self, src: str, version: str = "", split: str = "train", camera: str = "M", camera_dropout: float = 0.0,
                 image_size: Union[int, Tuple[int, int], None] = None, radar: str = "BF", radar_dropout: float = 0.0,
                 lidar: int = 0, label: str = "detection", num_classes: int = 1, sequential: bool = False,
                 scale: bool = True, fov: Dict[str, Sequence[float]] = None, dtype: str = "float32",
                 device_transforms: bool = True, **kwargs):
        super().__init__()
        if camera_dropout + radar_dropout > 1.0:
            raise ValueError("camera_dropout + radar_dropout must not exceed 1")
        if sequential:
            raise NotImplementedError("sequential K-Radar items are not implemented upstream either (dataset.py:171-175)")
        self.src, self.version = src, version
        self.split = f"{version}_{split}" if version else split
        self.camera, self.radar, self.lidar = camera or "", radar or "", lidar
        self.camera_dropout, self.radar_dropout = camera_dropout, radar_dropout
        self.image_size, self.label, self.num_classes, self.scale = image_size, label, num_classes, scale
        self.fov = dict(fov) if fov is not None else {}
        self.dtype = getattr(torch, dtype)
        self.device_transforms = device_transforms
        self.samples = self._walk()

    @classmethod
    def from_config(cls, config: Dict[str, Any], *args, **kwargs) -> "KRadarFolderDataset":
        return cls(*args, **{**config["computing"], **config["data"]}, **kwargs)

    # ------------------------------------------------------------------------------------------ folder tree
    def _files_of(self, folder: str) -> Dict[str, str]:
        """Entry name -> file of one sample folder, in the order the sample dict is built."""
        files: Dict[str, str] = {}
        for flag, key, stem in _CAMERAS:
            if flag in self.camera:
                files[key] = os.path.join(folder, stem + ".jpg")
                files["label_to_" + key] = os.path.join(folder, stem + "_info.npy")
        for flag, key, stem in _RADARS:
            if flag in self.radar:
                files[key] = os.path.join(folder, stem + ".npy")
                files["label_to_" + key] = os.path.join(folder, stem + "_info.npy")
        if self.lidar in (1, 2):
            files["lidar_top"] = os.path.join(folder, f"os{self.lidar}.npy")
        if self.label == "detection":
            files["label"] = os.path.join(folder, "labels.npy")
        files["description"] = os.path.join(folder, "description.npy")
        return files

    def _walk(self) -> List[Dict[str, str]]:
        root = os.path.join(self.src, self.split)
        out: List[Dict[str, str]] = []
        # sorted: the index -> sample map feeds ShardedSampler on every rank, and os.listdir order is file-system dependent
        # (upstream takes directory order; it is single-process, and the order is not observable after shuffling)
        for sequence in sorted(os.listdir(root)):
            seq_dir = os.path.join(root, sequence)
            out += [self._files_of(os.path.join(seq_dir, s)) for s in sorted(os.listdir(seq_dir))]
        return out

    def __len__(self) -> int:
        return len(self.samples)

    # ------------------------------------------------------------------------------------------ one sample
    def _load(self, files: Dict[str, str]) -> Dict[str, torch.Tensor]:
        sample: Dict[str, torch.Tensor] = {}
        for key, path in files.items():
            ext = os.path.splitext(path)[-1]
            if ext in _IMAGE_EXT:
                img = read_image_hwc(path)
                sample[key] = img if self.device_transforms else img.to(self.dtype)
            elif ext == ".npy":
                sample[key] = torch.from_numpy(np.load(path)).to(self.dtype)
        return sample

    def _dropout(self, sample: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """One draw per sample: nothing / the cameras / the radar maps are zeroed with the configured probabilities."""
        p_none = 1.0 - (self.camera_dropout + self.radar_dropout)
        pick = int(np.random.choice(3, replace=True, p=[p_none, self.camera_dropout, self.radar_dropout]))
        drop = ((), ("camera_mono", "camera_stereo"), ("radar_bev", "radar_front"))[pick]
        for key in drop:
            if key in sample:
                sample[key] = torch.zeros_like(sample[key])
        return sample

    def __getitem__(self, index: int):
        sample = self._load(self.samples[index])
        if self.scale and not self.device_transforms:
            for key in ("radar_bev", "radar_front"):
                if key in sample:
                    sample[key] = torch.clip((sample[key] - MIN_POWER) / (MAX_POWER - MIN_POWER) * 255.0, 0, 255)
        sample = self._dropout(sample)
        label: Dict[str, Any] = {}
        if self.label == "detection":
            label = detection_label(sample.pop("label"), self.num_classes, self.fov, self.dtype)
        label["description"] = sample.pop("description")
        # transformations (cartesian space): identity placeholder (zeros) for cameras, the calibration for radar grids
        for flag, key, _ in _CAMERAS:
            if flag in self.camera:
                sample[f"label_to_{key}_t"] = torch.zeros_like(sample[f"label_to_{key}"])
        for flag, key, _ in _RADARS:
            if flag in self.radar:
                sample[f"label_to_{key}_t"] = sample.pop(f"label_to_{key}")
        # projections (sensor space): the camera matrix from the file, the fixed grid projections for the radar maps
        for flag, key, _ in _CAMERAS:
            if flag in self.camera:
                sample[f"label_to_{key}_p"] = sample.pop(f"label_to_{key}")
        if "B" in self.radar:
            sample["label_to_radar_bev_p"] = ra_projection(self.dtype)
        if "F" in self.radar:
            sample["label_to_radar_front_p"] = ea_projection(self.dtype)
        # shapes of the inputs BEFORE any resize: the reference points are normalised by them
        for flag, key, _ in _CAMERAS + _RADARS:
            if flag in (self.camera if key.startswith("camera") else self.radar):
                sample[f"{key}_shape"] = torch.as_tensor(sample[key].shape)
        if self.image_size is not None and not self.device_transforms:
            from dpft_amd.data.preprocess import resized_output_size
            for flag, key, _ in _CAMERAS:
                if flag in self.camera:
                    img = sample[key].movedim(-1, 0).unsqueeze(0)
                    size = resized_output_size(img.shape[2], img.shape[3], self.image_size)
                    sample[key] = torch.nn.functional.interpolate(img, size=size, mode="bilinear", align_corners=False)[0].movedim(0, -1)
        return sample, label


def initialize_kradar(*args, **kwargs) -> KRadarFolderDataset:
    return KRadarFolderDataset.from_config(*args, **kwargs)
