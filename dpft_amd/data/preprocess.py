"""Online sample transforms of ``KRadarDataset.__getitem__`` on the device.

Reference order (src/dprt/datasets/kradar/dataset.py:140-169): load -> ``scale_radar_data`` (:295-317) -> labels ->
transformations / projections / ``_add_shape`` (shape recorded BEFORE the resize) -> ``resize_image`` (:319-341).
Here the loader ships raw frames (camera as decoded u8 or float HWC, radar maps in dB) and this module runs the two
arithmetic transforms as HIP kernels on the upload stream; everything else in the sample dict passes through.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple, Union

import torch

from dpft_amd.hip.lib import HipLibraryError, lib, ptr, stream

MIN_POWER, MAX_POWER = 100.0, 200.0        # src/dprt/datasets/kradar/utils/radar_info.py:109,113


def resized_output_size(h: int, w: int, size: Union[int, Sequence[int]]) -> Tuple[int, int]:
    """torchvision.transforms.functional.resize size rule: an int matches the SHORT side, the long side is
    int(size * long / short); a pair is taken as (H, W)."""
    if isinstance(size, (tuple, list)):
        if len(size) == 2:
            return int(size[0]), int(size[1])
        size = size[0]
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = int(size), int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def _check(t: torch.Tensor):
    if not t.is_cuda:
        raise HipLibraryError("dpft_amd.data.GpuPreprocessor needs device tensors; there is no CPU path")


def resize_bilinear(frames: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """(B,H,W,C) float32 or uint8 -> (B,size[0],size[1],C) float32, bilinear, align_corners=False, no antialias."""
    _check(frames)
    frames = frames.contiguous()
    B, Hs, Ws, C = frames.shape
    out = torch.empty((B, size[0], size[1], C), dtype=torch.float32, device=frames.device)
    if frames.dtype == torch.uint8:
        fn = "dpft_resize_bilinear_nhwc_u8"
    elif frames.dtype == torch.float32:
        fn = "dpft_resize_bilinear_nhwc_f32"
    else:
        raise TypeError(f"resize_bilinear: unsupported dtype {frames.dtype}")
    lib.call(fn, ptr(frames), ptr(out), B, Hs, Ws, size[0], size[1], C, stream())
    return out


def scale_clip(x: torch.Tensor, in_lo: float = MIN_POWER, in_hi: float = MAX_POWER, out_lo: float = 0.0,
               out_hi: float = 255.0) -> torch.Tensor:
    _check(x)
    x = x.contiguous().float()
    y = torch.empty_like(x)
    lib.call("dpft_scale_clip_f32", ptr(x), ptr(y), x.numel(), float(in_lo), float(in_hi), float(out_lo), float(out_hi),
             stream())
    return y


class GpuPreprocessor:
    def __init__(self, image_size: Union[int, Sequence[int], None] = 512, scale: bool = True,
                 camera_keys: Sequence[str] = ("camera_mono", "camera_stereo"),
                 radar_keys: Sequence[str] = ("radar_bev", "radar_front")):
        self.image_size, self.scale = image_size, scale
        self.camera_keys, self.radar_keys = tuple(camera_keys), tuple(radar_keys)

    @classmethod
    def from_config(cls, config: Dict) -> "GpuPreprocessor":
        d = config.get("data", {})
        return cls(image_size=d.get("image_size", 512), scale=d.get("scale", True))

    def __call__(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        out = dict(batch)
        for k in self.radar_keys:
            if k in out and self.scale:
                out[k] = scale_clip(out[k])
        for k in self.camera_keys:
            if k in out:
                v = out[k]
                if self.image_size is not None:
                    v = resize_bilinear(v, resized_output_size(v.shape[1], v.shape[2], self.image_size))
                elif v.dtype != torch.float32:
                    v = v.float()
                out[k] = v
        return out
