"""TEST INFRASTRUCTURE -- numpy restatement of the radar tesseract -> (range-azimuth, elevation-azimuth) projection of
``KRadarProcessor.get_radar_data`` (src/dprt/datasets/kradar/processor.py:588-633): power in dB, then per output cell six
features = (max, median, variance) of the radar cross-section and (argmax raster value, median / mean, variance) of the
doppler profile.  Kept quirks: the EA "doppler median" is a MEAN (:621), variances are variance-of-variance (:603,:617),
the range bins 0-3 and 252-255 are cropped for the EA map only (:611).  Pinned by tests/golden/radar_projection.npz,
produced by the reference's own method on a seeded synthetic tesseract (oracle/gen_golden.py)."""
from __future__ import annotations

import numpy as np


def _features(db: np.ndarray, axis: int, raster: np.ndarray, doppler_center: str):
    """db (D, ...) with `axis` the dimension that is folded away besides doppler (axis 0)."""
    peak = np.max(db, axis=axis)                       # (D, kept...)  max over the folded spatial dimension
    rcs_max = np.max(peak, axis=0)
    rcs_median = np.median(np.median(db, axis=axis), axis=0)
    rcs_var = np.var(np.var(db, axis=axis), axis=0)
    dop_max = raster[np.argmax(peak, axis=0)]
    dop_mid = np.median(peak, axis=0) if doppler_center == "median" else np.mean(peak, axis=0)
    dop_var = np.var(peak, axis=0)
    return np.dstack((rcs_max, rcs_median, rcs_var, dop_max, dop_mid, dop_var))


def radar_projection(tesseract: np.ndarray, doppler_raster) -> tuple:
    """tesseract (doppler, range, elevation, azimuth) linear power -> ra (range, azimuth, 6), ea (elevation, azimuth, 6)."""
    raster = np.asarray(doppler_raster)
    db = 10 * np.log10(tesseract)                      # processor.py:598
    ra = _features(db, 2, raster, "median")            # fold elevation            (:600-608)
    ea = _features(db[:, 4:252], 1, raster, "mean")    # crop, then fold range     (:611-622)
    return ra, ea
