"""Per-step detection metrics, mirror of ``src/dprt/evaluation/metric.py`` (``Metric`` :256-345 with ``mAP3D`` :16-151
and ``mGIoU3D`` :154-253, built by ``build_metric(config['evaluate'])`` and called as ``eval_fn(output, labels)`` in
every training and validation step, src/dprt/training/trainer.py:134).

The reference evaluates sample by sample, class by class with ~50 small tensor ops each (and pytorch3d box overlaps);
here both metrics of the whole batch come from two HIP launches (``dpft_detection_metrics_f32``: pairwise IoU3D / GIoU3D,
then one block per sample) on padded targets.  Same values, including the reference's quirks (two-point precision /
recall "interpolation", smallest present label dropped from the class mean, degenerate boxes count as IoU 0 / GIoU -1).
"""
from __future__ import annotations

from typing import Any, Dict, List

import torch
from torch import nn

_KNOWN = {"mAP3D", "mGIoU3D"}


class Metric(nn.modules.loss._Loss):
    def __init__(self, metrics: Dict[str, str] = None, reduction: str = "mean", threshold: float = 0.5,
                 nelem: int = 101, **kwargs):
        super().__init__(**kwargs)
        if reduction not in {"none", "mean", "sum"}:
            raise ValueError(f"Invalid Value for arg 'reduction': '{reduction}'")
        self.metrics = dict(metrics) if metrics is not None else {}
        for name, kind in self.metrics.items():
            if kind not in _KNOWN:
                raise ValueError(f"dpft_amd Metric supports {sorted(_KNOWN)}, got {kind!r} for {name!r}")
        self.reduction, self.threshold, self.nelem = reduction, threshold, nelem

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "Metric":
        return cls(metrics=config.get("metrics"), reduction=config.get("reduction", "mean"))

    @torch.no_grad()
    def forward(self, inputs: Dict[str, torch.Tensor], targets: List[Dict[str, torch.Tensor]]):
        if not self.metrics:
            return torch.ones(1)
        import ctypes as C  # noqa: F401
        from dpft_amd.hip.lib import HipLibraryError, lib, stream
        from dpft_amd.training.loss import pack_targets
        cls = inputs["class"].detach().contiguous().float()
        if not cls.is_cuda:
            raise HipLibraryError("dpft_amd Metric needs device tensors; there is no CPU path")
        center, size = inputs["center"].detach().contiguous().float(), inputs["size"].detach().contiguous().float()
        angle = inputs["angle"].detach().contiguous().float()
        B, N, ncls = cls.shape
        dev = cls.device
        counts = [int(t["gt_class"].shape[0]) for t in targets]
        gt_box, gt_onehot, _gt_id, counts_t, Mmax = pack_targets(targets, counts, ncls, dev)
        scratch = torch.empty((B, N, Mmax, 2), dtype=torch.float32, device=dev)
        out = torch.empty((B, 2), dtype=torch.float32, device=dev)
        lib.call("dpft_detection_metrics_f32", cls.data_ptr(), center.data_ptr(), size.data_ptr(), angle.data_ptr(),
                 gt_box.data_ptr(), gt_onehot.data_ptr(), counts_t.data_ptr(), float(self.threshold), int(self.nelem),
                 scratch.data_ptr(), out.data_ptr(), B, N, Mmax, ncls, stream())
        col = {"mAP3D": 0, "mGIoU3D": 1}
        res = {name: out[:, col[kind]] for name, kind in self.metrics.items()}
        if self.reduction != "none":
            res = {k: getattr(torch, self.reduction)(v) for k, v in res.items()}
        return res


def build_metric(*args, **kwargs):
    return Metric.from_config(*args, **kwargs)
