"""4-D radar tesseract -> model inputs on the device.

``KRadarProcessor.get_radar_data`` (src/dprt/datasets/kradar/processor.py:588-633) reduces each (doppler 64, range 256,
elevation 37, azimuth 107) power cube to the range-azimuth map ``radar_bev`` (256,107,6) and the elevation-azimuth map
``radar_front`` (37,107,6) with numpy on the CPU while the dataset is prepared; ``dpft_radar_projection_f32`` does it in
three launches that read the 259 MB cube twice.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from dpft_amd.hip.lib import HipLibraryError, lib, ptr, stream

# K-Radar doppler raster in m/s (src/dprt/datasets/kradar/utils/radar_info.py:16-30): 64 uniform bins, bin 32 = 0,
# first bin -1.93259122, tabulated with 8 decimals
DOPPLER_MIN = -1.93259122
RANGE_CROP = (4, 252)          # processor.py:611


def doppler_raster(n: int = 64, device="cpu") -> torch.Tensor:
    k = torch.arange(n, dtype=torch.float64) - n // 2
    table = torch.round(k * (-DOPPLER_MIN / (n // 2)) * 1e8) / 1e8
    return table.to(torch.float32).to(device)


def radar_projection(tesseract: torch.Tensor, raster: Optional[torch.Tensor] = None,
                     crop: Tuple[int, int] = RANGE_CROP) -> Tuple[torch.Tensor, torch.Tensor]:
    """tesseract (D,R,E,A) linear power on the device -> (ra (R,A,6), ea (E,A,6)) float32."""
    if not tesseract.is_cuda:
        raise HipLibraryError("dpft_amd.data.radar_projection needs a device tensor; there is no CPU path")
    t = tesseract.contiguous().float()
    D, R, E, A = t.shape
    raster = doppler_raster(D, t.device) if raster is None else raster.to(t.device).contiguous().float()
    if raster.numel() != D:
        raise ValueError(f"doppler raster has {raster.numel()} entries for {D} doppler bins")
    ra = torch.empty((R, A, 6), dtype=torch.float32, device=t.device)
    ea = torch.empty((E, A, 6), dtype=torch.float32, device=t.device)
    scratch = torch.empty(int(lib.dpft_radar_projection_scratch_floats(D, R, E, A)), dtype=torch.float32, device=t.device)
    lib.call("dpft_radar_projection_f32", ptr(t), ptr(raster), ptr(ra), ptr(ea), ptr(scratch), D, R, E, A,
             int(crop[0]), int(min(crop[1], R)), stream())
    return ra, ea
