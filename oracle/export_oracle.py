"""TEST INFRASTRUCTURE -- CPU restatement of the reference's K-Radar exporter
(src/dprt/evaluation/exporters/kradar.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this; the product path (dpft_amd/evaluation/exporters/kradar.py) selects objects with dpft_export_select_f32.

Pinned by tests/golden/export.json (file trees written by the reference's own KRadarExporter, oracle/gen_golden.py).
The exporter is restated as a pure function: instead of appending to files it returns ``{relative path: text}``
with the text the reference's ``write`` calls accumulate ('a+' mode, :226-229).
"""
from __future__ import annotations

import itertools
from typing import Dict, List

import numpy as np
import torch

DEFAULT_CATEGORIES = {0: "Sedan", 1: "Bus or Truck", 2: "Motorcycle", 3: "Bicycle", 4: "Bicycle Group",
                      5: "Pedestrian", 6: "Pedestrian Group", 7: "Background"}                     # :62-71
DEFAULT_ROADS = {0: "urban", 1: "highway", 2: "alleyway", 3: "suburban", 4: "university", 5: "mountain",
                 6: "parkinglots", 7: "shoulder", 8: "countryside"}                                # :100-110
DEFAULT_WEATHER = {0: "normal", 1: "overcast", 2: "fog", 3: "rain", 4: "sleet", 5: "lightsnow", 6: "heavysnow"}
DEFAULT_TIME = {0: "day", 1: "night"}
CATEGORY_TO_CLS = {"Sedan": "sed", "Bus or Truck": "bus", "Motorcycle": "mot", "Bicycle": "bic",
                   "Bicycle Group": "big", "Pedestrian": "ped", "Pedestrian Group": "peg", "Background": "bg"}  # :43-52
DUMMY = "dummy -1 -1 0 0 0 0 0 0 0 0 0 0 0 0 0"                                                    # :212


def invert(mapping, default):
    """The property setters keep ``{value: key}`` of a config mapping (:83-84), defaults are already inverted."""
    return dict(default) if mapping is None else {v: k for k, v in mapping.items()}


def selection_mask(cls: torch.Tensor, center: torch.Tensor, angle: torch.Tensor, conf_thr: float):
    """cls_mask & conf_mask & fov_mask of ONE sample (:259-277); also returns categories and yaw."""
    confidence, categories = torch.max(cls, dim=-1)
    yaw = torch.atan2(angle[..., 0], angle[..., 1])
    categories = categories - 1
    x, y, z = center[:, 0], center[:, 1], center[:, 2]
    fov = (0 < x) & (x < 72) & (-6.4 < y) & (y < 6.4) & (-2.0 < z) & (z < 6.0) & (-50.0 < yaw) & (yaw < 50.0)
    return (categories >= 0) & (confidence >= conf_thr) & fov, categories, yaw


def construct_objects(objects: Dict[str, torch.Tensor], conf_thr: float, pre: str = "") -> np.ndarray:
    """(n, 15) float64 rows: name, truncated, occluded, alpha, bbox x4, h, w, l, y, z, x, theta (:279-293; the hstack
    of int64 / float64 / float32 parts promotes to float64)."""
    pre = f"{pre}_" if pre else pre
    mask, categories, yaw = selection_mask(objects[f"{pre}class"], objects[f"{pre}center"], objects[f"{pre}angle"],
                                           conf_thr)
    n = int(mask.sum())
    out = np.zeros((n, 15), dtype=np.float64)
    out[:, 0] = categories[mask].numpy()
    out[:, 4:8] = [50, 50, 150, 150]
    out[:, 8:11] = objects[f"{pre}size"][mask][:, [2, 1, 0]].double().numpy()
    out[:, 11:14] = objects[f"{pre}center"][mask][:, [1, 2, 0]].double().numpy()
    out[:, 14] = yaw[mask].double().numpy()
    return out


def serialize_object(row: np.ndarray, categories: Dict[int, str]) -> str:
    """:315-347."""
    return " ".join([CATEGORY_TO_CLS[categories[row[0]]]] + [str(int(v)) for v in row[1:8]]
                    + [str(round(v, 2)) for v in row[8:15]])


def export_tree(outputs: Dict[str, torch.Tensor], targets: List[Dict[str, torch.Tensor]], step: int,
                conf_thrs=None, categories=None, roads=None, weather=None, time_zone=None) -> Dict[str, str]:
    """KRadarExporter.export (:485-514) as ``{path relative to dst: accumulated text}``."""
    conf_thrs = [0.0, 0.3, 0.5, 0.7, 0.9] if conf_thrs is None else conf_thrs                       # :39
    cats, roads = invert(categories, DEFAULT_CATEGORIES), invert(roads, DEFAULT_ROADS)
    weather, time_zone = invert(weather, DEFAULT_WEATHER), invert(time_zone, DEFAULT_TIME)
    tree: Dict[str, str] = {}

    def write(lines, path):
        tree[path] = tree.get(path, "") + "".join(s + "\n" for s in lines)

    def describe(d):                                                                                # :296-313
        d = d.detach().cpu().numpy()
        return [time_zone[int(d[1])], roads[int(d[0])], weather[int(d[2])]]

    for thr in conf_thrs:
        folder = "/".join(("exports", "kradar", str(thr)))
        for i, tgt in enumerate(targets):                                                           # :461-483, :393-425
            objs = [serialize_object(r, cats) for r in construct_objects(tgt, thr, pre="gt")] or [DUMMY]
            desc = describe(tgt["description"])
            for sub in itertools.chain(["all"], desc):
                name = f"{str(step + i).zfill(6)}.txt"
                write(desc, f"{folder}/{sub}/desc/{name}")
                write(objs, f"{folder}/{sub}/gts/{name}")
                write([str(step + i).zfill(6)], f"{folder}/{sub}/val.txt")
        for i, tgt in enumerate(targets):                                                           # :427-459, :362-391
            sample = {k: v[i] for k, v in outputs.items()}
            objs = [serialize_object(r, cats) for r in construct_objects(sample, thr)] or [DUMMY]
            for sub in itertools.chain(["all"], describe(tgt["description"])):
                write(objs, f"{folder}/{sub}/preds/{str(step + i).zfill(6)}.txt")
    return tree
