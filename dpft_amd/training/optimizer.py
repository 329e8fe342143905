"""Optimizers.  ``build_optimizer`` mirrors src/dprt/training/optimizer.py:6-7 (``getattr(torch.optim,
name)``); for AdamW on CUDA it returns ``FusedAdamW``: the same update rule as torch.optim.AdamW
(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=1e-2, no amsgrad) executed by ONE HIP launch over all
parameter tensors (dpft_adamw_f32), with the moments in two flat fp32 buffers."""
from __future__ import annotations

import numpy as np
import torch

from dpft_amd.hip.lib import lib, note_weights_changed, ptr, stream

CHUNK = 16384


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(list(params), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = None
        self._step = 0

    @staticmethod
    def _layout(t: torch.Tensor):
        """Set of dense physical element orders a tensor has: 'plain' (row-major) and/or 'khwc'."""
        out = set()
        if t.is_contiguous():
            out.add("plain")
        if t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous():
            out.add("khwc")
        return out

    def _build(self):
        tables = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            total = sum(p.numel() for p in ps)
            dev = ps[0].device
            m = torch.zeros(total, dtype=torch.float32, device=dev)
            v = torch.zeros(total, dtype=torch.float32, device=dev)
            rows, off = [], 0
            for ti, p in enumerate(ps):
                assert p.grad is not None and (self._layout(p) & self._layout(p.grad)), \
                    "FusedAdamW expects persistent, dense gradient buffers laid out like their parameters"
                pp, gp = p.data_ptr(), p.grad.data_ptr()
                for c0 in range(0, p.numel(), CHUNK):
                    n = min(CHUNK, p.numel() - c0)
                    rows.append((pp + 4 * c0, gp + 4 * c0, m.data_ptr() + 4 * (off + c0), v.data_ptr() + 4 * (off + c0),
                                 n, ti))
                off += p.numel()
            arr = np.zeros(len(rows), dtype=np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"),
                                                      ("n", "<i4"), ("t", "<i4")]))
            for i, r in enumerate(rows):
                arr[i] = r
            chunks = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
            active = torch.ones(len(ps), dtype=torch.int32, device=dev)
            tables.append(dict(params=ps, ptrs=[(p.data_ptr(), p.grad.data_ptr()) for p in ps], m=m, v=v,
                               chunks=chunks, n_chunks=len(rows), active=active, active_host=None))
        self._tables = tables

    def set_active(self, active_ids):
        """ids of parameters that received a gradient this step (others are skipped like ``grad is None``)."""
        self._active_ids = active_ids

    @torch.no_grad()
    def step(self, closure=None):
        if self._tables is None or any(
                [(p.data_ptr(), p.grad.data_ptr()) for p in t["params"]] != t["ptrs"] for t in self._tables):
            self._build()
        self._step += 1
        ids = getattr(self, "_active_ids", None)
        for group, t in zip(self.param_groups, self._tables):
            if ids is not None:
                host = [int(id(p) in ids) for p in t["params"]]
                if host != t["active_host"]:
                    t["active"].copy_(torch.tensor(host, dtype=torch.int32))
                    t["active_host"] = host
            b1, b2 = group["betas"]
            lib.call("dpft_adamw_f32", ptr(t["chunks"]), t["n_chunks"], ptr(t["active"]) if ids is not None else None,
                     float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                     self._step, stream())
        note_weights_changed()                             # in-place through raw pointers: no _version bump
        return None


def build_optimizer(name: str, params, device=None, **kwargs):
    if name == "AdamW" and device is not None and torch.device(device).type == "cuda" and not kwargs.get("amsgrad"):
        return FusedAdamW(params, **kwargs)
    return getattr(torch.optim, name)(params, **kwargs)
