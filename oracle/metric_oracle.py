"""TEST INFRASTRUCTURE -- CPU restatement of the reference's per-step detection metrics
(src/dprt/evaluation/metric.py: mAP3D :16-151, mGIoU3D :154-253, Metric :256-345), line for line in torch, with
``dprt.utils.iou.iou3d / giou3d`` (:72-210, built on the absent third-party pytorch3d.box3d_overlap -- parity
unpinned for that call) replaced by the oracle's exact yaw-only box geometry.  The reference's quirks are kept on
purpose: the "interpolation" of the precision/recall curve only uses its first and last point (utils/misc.py:43-83),
the class selection drops the smallest PRESENT label rather than label 0 (:141,:243), masked-out boxes become
degenerate boxes (IoU 0 / GIoU -1).  Pinned by tests/golden/metric.npz, which is produced by the reference's own
classes (oracle/gen_golden.py) with box3d_overlap bound to ``box3d_overlap_from_corners`` below.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from oracle.dprt_oracle import _poly_clip_area, box_corners


def box3d_overlap_from_corners(b1: torch.Tensor, b2: torch.Tensor):
    """Stand-in for pytorch3d.ops.box3d_overlap on yaw-only boxes: (n,8,3),(m,8,3) -> (vol (n,m), iou (n,m)).
    Corner order of src/dprt/utils/bbox.py:4-74 (0-3 bottom face, 4-7 top face)."""
    n, m = b1.shape[0], b2.shape[0]
    vol = torch.zeros(n, m, dtype=b1.dtype)
    iou = torch.zeros(n, m, dtype=b1.dtype)
    a1, a2 = b1.double(), b2.double()

    def area(p):
        s = 0.0
        for i in range(4):
            x1, y1 = p[i]; x2, y2 = p[(i + 1) % 4]
            s += x1 * y2 - x2 * y1
        return abs(s) / 2
    for i in range(n):
        pa = [(float(a1[i, t, 0]), float(a1[i, t, 1])) for t in range(4)]
        v1 = area(pa) * float(a1[i, 4, 2] - a1[i, 0, 2])
        for j in range(m):
            pb = [(float(a2[j, t, 0]), float(a2[j, t, 1])) for t in range(4)]
            v2 = area(pb) * float(a2[j, 4, 2] - a2[j, 0, 2])
            zlo = max(float(a1[i, 0, 2]), float(a2[j, 0, 2])); zhi = min(float(a1[i, 4, 2]), float(a2[j, 4, 2]))
            v = _poly_clip_area(pa, pb) * max(0.0, zhi - zlo)
            vol[i, j] = v
            iou[i, j] = v / (v1 + v2 - v) if v > 0 else 0.0
    return vol, iou


def _valid(size: torch.Tensor) -> torch.Tensor:
    """_check_nonzero (src/dprt/utils/iou.py:39-69): every face-triangle area > 1e-4."""
    l, w, h = size[..., 0].double(), size[..., 1].double(), size[..., 2].double()
    return torch.minimum(torch.minimum(l * w, l * h), w * h) / 2 > 1e-4


def iou_giou(c1, s1, a1, c2, s2, a2):
    """(iou, giou) (N,M) of yaw-only boxes incl. the reference's invalid-box conventions (iou.py:72-210)."""
    N, M = c1.shape[0], c2.shape[0]
    k1, k2 = box_corners(c1.double(), s1.double(), a1.double()), box_corners(c2.double(), s2.double(), a2.double())
    iou = torch.zeros(N, M, dtype=torch.float64)
    giou = torch.zeros(N, M, dtype=torch.float64)
    v1, v2 = _valid(s1), _valid(s2)
    for i in range(N):
        for j in range(M):
            if not (bool(v1[i]) and bool(v2[j])):
                giou[i, j] = -1.0                      # evol keeps its -1 initialiser, iou = uni = 0
                continue
            vol, io = box3d_overlap_from_corners(k1[i:i + 1], k2[j:j + 1])
            vol, io = float(vol), float(io)
            allc = torch.cat((k1[i], k2[j]), 0)
            evol = float((allc.max(0).values - allc.min(0).values).prod())
            uni = vol / io if io != 0 else 0.0
            iou[i, j] = io
            giou[i, j] = (io - (evol - uni) / evol) if evol != 0 else 0.0
    return iou, giou


def _interp(x, xp, fp, left=None, right=None):
    """src/dprt/utils/misc.py:43-83 -- a straight line through the FIRST and LAST sample."""
    x0, x1, y0, y1 = xp[0], xp[-1], fp[0], fp[-1]
    left = left if left is not None else y0
    right = right if right is not None else y1
    if torch.isclose((x1 - x0), torch.zeros_like(x0)):
        y = torch.zeros_like(x)
    else:
        y = y0 + (x - x0) * (y1 - y0) / (x1 - x0)
    y[x < x0] = left
    y[x > x1] = right
    return y


def _pair_matrices(inp, tgt):
    ang = torch.atan2(inp["angle"][..., 0], inp["angle"][..., 1])
    gang = torch.atan2(tgt["gt_angle"][..., 0], tgt["gt_angle"][..., 1])
    return iou_giou(inp["center"], inp["size"], ang, tgt["gt_center"], tgt["gt_size"], gang)


def map3d(inp: Dict[str, torch.Tensor], tgt: Dict[str, torch.Tensor], threshold: float = 0.5, nelem: int = 101):
    """mAP3D.forward for ONE sample (tensors without the dummy batch dimension): metric.py:31-151."""
    C = tgt["gt_class"].shape[-1]
    label, gt_label = torch.argmax(inp["class"], -1), torch.argmax(tgt["gt_class"], -1)
    iou_full, _ = _pair_matrices(inp, tgt)
    aps = torch.zeros(C)
    for l in range(C):
        mask, gt_mask = label == l, gt_label == l
        # corners of other classes are zeroed -> degenerate -> IoU 0 (:76-85)
        iou = (iou_full * mask[:, None] * gt_mask[None, :]).float()
        npos = gt_mask.sum().float()
        sort_idx = torch.argsort(inp["class"][..., l], descending=True)
        iou, mask_s = iou[sort_idx, :], mask[sort_idx]
        thr_mask = iou > threshold
        iou_mask = torch.logical_and(*torch.meshgrid(mask_s, gt_mask, indexing="ij"))
        tp_c = torch.logical_and(iou_mask, thr_mask)
        tp, fp = torch.zeros(iou.shape[0]), torch.ones(iou.shape[0])
        if tp_c.shape[1] > 0:
            tp_value, tp_idx = torch.max(tp_c.to(torch.uint8), dim=0)
            tp_value = tp_value.bool()
            tp[tp_idx[tp_value]] = 1
            fp[tp_idx[tp_value]] = 0
        fp[~mask_s] = 0
        tp, fp = torch.cumsum(tp, 0), torch.cumsum(fp, 0)
        prec = torch.zeros_like(tp)
        div = (fp + tp != 0)
        prec[div] = tp[div] / (fp[div] + tp[div])
        rec = torch.ones_like(tp) if npos == 0 else tp / npos
        grid = torch.linspace(0, 1, nelem, dtype=rec.dtype)
        prec = _interp(grid, rec, prec, right=0)
        aps[l] = torch.sum(prec * 1 / (nelem - 1))
    selection = torch.sort(torch.unique(torch.cat([label, gt_label], 0)))[0][1:]
    if not selection.numel() or not selection.any():
        return torch.ones(())
    return torch.mean(aps[selection])


def mgiou3d(inp: Dict[str, torch.Tensor], tgt: Dict[str, torch.Tensor]):
    """mGIoU3D.forward for ONE sample: metric.py:160-253."""
    C = tgt["gt_class"].shape[-1]
    label, gt_label = torch.argmax(inp["class"], -1), torch.argmax(tgt["gt_class"], -1)
    _, giou_full = _pair_matrices(inp, tgt)
    gious = -torch.ones(C)
    for l in range(C):
        mask, gt_mask = label == l, gt_label == l
        pair = torch.logical_and(*torch.meshgrid(mask, gt_mask, indexing="ij"))
        giou = torch.where(pair, giou_full, -torch.ones_like(giou_full)).float()      # degenerate boxes: -1 (:203-205)
        sort_idx = torch.argsort(inp["class"][..., l], descending=True)
        giou, mask_s = giou[sort_idx, :], mask[sort_idx]
        giou_mask = torch.logical_and(*torch.meshgrid(mask_s, gt_mask, indexing="ij"))
        giou[~giou_mask] = -1
        if gt_mask.sum() == 0:
            gious[l] = 1.0
        if giou.shape[1] > 0 and giou.shape[0] > 0:
            match_giou, _ = torch.max(giou, dim=0)
            if match_giou.numel() > 0 and giou_mask.any():
                gious[l] = torch.mean(match_giou)
    selection = torch.sort(torch.unique(torch.cat([label, gt_label], 0)))[0][1:]
    if not selection.numel() or not selection.any():
        return torch.ones(())
    return torch.mean(gious[selection])


def metric_forward(inputs: Dict[str, torch.Tensor], targets: List[Dict[str, torch.Tensor]], reduction: str = "mean"):
    """Metric.forward (:307-345) with metrics {'mAP': mAP3D, 'mGIoU': mGIoU3D} (config/kradar.json:173-176)."""
    per = []
    for b, tgt in enumerate(targets):
        inp = {k: v[b] for k, v in inputs.items()}
        per.append({"mAP": map3d(inp, tgt), "mGIoU": mgiou3d(inp, tgt)})
    out = {k: torch.stack([p[k] for p in per]) for k in per[0]}
    if reduction != "none":
        out = {k: getattr(torch, reduction)(v) for k, v in out.items()}
    return out
