"""Optimizers.  ``build_optimizer`` mirrors src/dprt/training/optimizer.py:6-7 (``getattr(torch.optim,
name)``); for AdamW on CUDA it returns ``FusedAdamW``: the same update rule as torch.optim.AdamW
(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=1e-2, no amsgrad) executed by ONE HIP launch over all
parameter tensors (dpft_adamw_f32), with the moments in two flat fp32 buffers (per-parameter state kept)."""
from __future__ import annotations

import numpy as np
import torch

from dpft_amd.hip.lib import lib, note_weights_changed, ptr, stream

CHUNK = 16384


class FusedAdamW(torch.optim.Optimizer):
    """Drop-in for ``torch.optim.AdamW`` (no amsgrad / maximize / capturable).  The moments of a parameter group live in
    two flat fp32 buffers; ``self.state[p]`` holds views into them, so ``state_dict()`` / ``load_state_dict()`` round-trip
    in torch's format.  Fast path: persistent gradient buffers (the DP reducer's buckets) - the pointer table is built
    once.  With a plain ``zero_grad(set_to_none=True)`` loop the table is re-pointed whenever a ``.grad`` moves; the
    moments and step counts are keyed by parameter and survive that.  ``grad is None`` = the parameter sits the step out
    (its own step count does not advance, exactly like torch)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(list(params), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = None
        self._step = 0                 # launches so far; a tensor's own count is _step - skipped[t]
        self._restore = False          # self.state was replaced by load_state_dict: re-seed the flat buffers from it
        # Segments (round 4): the DP reducer's buckets.  A bucket whose gradients are final (all produced, all-reduced) can
        # be stepped at once, on the stream that finished it, while the backward goes on elsewhere -- step_segment();
        # step() then only launches what is left.  Rows of the chunk table are ordered by segment for that.
        self._segments = None          # list of lists of parameters (a parameter not listed belongs to the trailing segment)
        self._round_open = False       # a step_segment() of the current step has already advanced _step
        self._done = set()             # (table index, segment index) launched in the current step

    def attach_segments(self, param_lists):
        """Declare the segments (lists of parameters, e.g. GradBucketReducer buckets) step_segment() may be called with."""
        self._segments = [list(ps) for ps in param_lists]
        self._tables = None            # rebuilt (row order) at the next step

    @staticmethod
    def _layout(t: torch.Tensor):
        """Set of dense physical element orders a tensor has: 'plain' (row-major) and/or 'khwc'."""
        out = set()
        if t.is_contiguous():
            out.add("plain")
        if t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous():
            out.add("khwc")
        return out

    @staticmethod
    def _ptrs(ps):
        return [(p.data_ptr(), None if p.grad is None else p.grad.data_ptr()) for p in ps]

    @staticmethod
    def _flat_view(flat, off, p):
        """View of ``flat[off : off + numel]`` with p's shape AND physical element order."""
        v = flat[off:off + p.numel()]
        if p.dim() == 4 and not p.is_contiguous():                    # khwc
            o, i, kh, kw = p.shape
            return v.view(o, kh, kw, i).permute(0, 3, 1, 2)
        return v.view(p.shape)

    def _build(self):
        """(Re)build the chunk tables.  Moments: first build -> zeros (or the loaded state); later builds keep the flat
        buffers and only refresh the parameter / gradient pointers."""
        old = self._tables
        if self._restore:
            steps = [float(st["step"]) for st in self.state.values() if "step" in st]
            self._step = int(max(steps)) if steps else 0
        tables = []
        if old is not None and not self._restore:
            # a group whose set of trainable parameters changed gets new flat buffers: the per-parameter step counts live
            # only in the old tables' device-side `skipped` counters (state["step"] is written by state_dict() alone), so
            # they are read back here -- otherwise a carried-over parameter would restart its bias correction at 0
            for gi, group in enumerate(self.param_groups):
                ps = [p for p in group["params"] if p.requires_grad]
                if gi < len(old) and [id(p) for p in old[gi]["params"]] != [id(p) for p in ps]:
                    for p, sk in zip(old[gi]["params"], old[gi]["skipped"].tolist()):
                        if p in self.state:
                            self.state[p]["step"] = float(self._step - int(sk))
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.requires_grad]
            dev = ps[0].device
            keep = old is not None and not self._restore and gi < len(old) and \
                [id(p) for p in old[gi]["params"]] == [id(p) for p in ps]
            if keep:
                m, v, skipped = old[gi]["m"], old[gi]["v"], old[gi]["skipped"]
            else:
                total = sum(p.numel() for p in ps)
                m = torch.zeros(total, dtype=torch.float32, device=dev)
                v = torch.zeros(total, dtype=torch.float32, device=dev)
                skip_host = [0] * len(ps)
            rows, off = [], 0
            seg_of = {}
            if self._segments is not None:
                for si, sp in enumerate(self._segments):
                    for q in sp:
                        seg_of[id(q)] = si
            n_seg = (len(self._segments) if self._segments is not None else 0) + 1      # + the trailing segment
            seg_rows = [[] for _ in range(n_seg)]
            seg_tensors = [[] for _ in range(n_seg)]
            for ti, p in enumerate(ps):
                rows = seg_rows[seg_of.get(id(p), n_seg - 1)]
                seg_tensors[seg_of.get(id(p), n_seg - 1)].append(ti)
                mv, vv = self._flat_view(m, off, p), self._flat_view(v, off, p)
                if not keep:
                    st = self.state.get(p, {})
                    if "exp_avg" in st:                                # loaded / carried-over state
                        mv.copy_(st["exp_avg"])
                        vv.copy_(st["exp_avg_sq"])
                        skip_host[ti] = self._step - int(float(st.get("step", 0)))
                    else:
                        skip_host[ti] = self._step                      # joins now: its own count starts at 0
                    self.state[p] = {"exp_avg": mv, "exp_avg_sq": vv}
                rows.append((0, 0, 0, 0, 0, ti))                        # marker row (advances skipped[t] when inactive)
                if p.grad is not None:
                    assert p.grad.dtype == torch.float32 and (self._layout(p) & self._layout(p.grad)), \
                        "FusedAdamW expects dense fp32 gradients laid out like their parameters"
                    pp, gp = p.data_ptr(), p.grad.data_ptr()
                    for c0 in range(0, p.numel(), CHUNK):
                        n = min(CHUNK, p.numel() - c0)
                        rows.append((pp + 4 * c0, gp + 4 * c0, m.data_ptr() + 4 * (off + c0),
                                     v.data_ptr() + 4 * (off + c0), n, ti))
                off += p.numel()
            seg_range, rows = [], []
            for sr in seg_rows:                                         # rows of a segment are contiguous in the table
                seg_range.append((len(rows), len(sr)))
                rows += sr
            if not keep:
                skipped = torch.tensor(skip_host, dtype=torch.int32).to(dev)
            arr = np.zeros(len(rows), dtype=np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"),
                                                      ("n", "<i4"), ("t", "<i4")]))
            for i, r in enumerate(rows):
                arr[i] = r
            chunks = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
            active = torch.ones(len(ps), dtype=torch.int32, device=dev)
            tables.append(dict(params=ps, ptrs=self._ptrs(ps), m=m, v=v, skipped=skipped, chunks=chunks,
                               n_chunks=len(rows), active=active, active_host=None, seg_range=seg_range,
                               seg_tensors=seg_tensors))
        self._tables = tables
        self._restore = False

    def set_gate(self, loss: "torch.Tensor" = None):
        """Device-side `if loss > 0` (trainer.py:131) for the NEXT step(): a scalar tensor that stays on the device; the launch
        updates nothing when it is not positive.  Consumed by step()."""
        self._gate = loss

    def set_active(self, active_ids):
        """ids of parameters that received a gradient this step (others are skipped like ``grad is None``)."""
        self._active_ids = active_ids

    CHUNK_BYTES = 40

    def _launch(self, group, t, first: int, count: int):
        if count <= 0:
            return
        b1, b2 = group["betas"]
        import ctypes as C
        lib.call("dpft_adamw_f32", C.c_void_p(t["chunks"].data_ptr() + first * self.CHUNK_BYTES), count, ptr(t["active"]),
                 ptr(t["skipped"]),
                 float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                 self._step, ptr(getattr(self, "_gate", None)), stream())

    @torch.no_grad()
    def step_segment(self, si: int) -> bool:
        """Update the parameters of segment ``si`` NOW, on the current stream (their gradients are final there).  Returns
        False -- and leaves the segment to step() -- when the tables are stale or a tensor of the segment sat the previous
        step out (its `active` flag on the device would have to change first).  The step count advances once per step."""
        if self._segments is None or self._restore or self._tables is None:
            return False
        if getattr(self, "gate_required", False) and getattr(self, "_gate", None) is None:
            return False      # the step's `loss > 0` decision is taken on the device: never launch ahead of its gate
        if any(self._ptrs(t["params"]) != t["ptrs"] for t in self._tables):
            return False
        todo = []
        for gi, (group, t) in enumerate(zip(self.param_groups, self._tables)):
            first, count = t["seg_range"][si]
            if count == 0 or (gi, si) in self._done:
                continue
            host = t["active_host"]
            if host is None or any(host[ti] != 1 for ti in t["seg_tensors"][si]):
                return False
            todo.append((gi, group, t, first, count))
        if not self._round_open:
            self._step += 1
            self._round_open = True
        for gi, group, t, first, count in todo:
            self._launch(group, t, first, count)
            self._done.add((gi, si))
        return True

    @torch.no_grad()
    def step(self, closure=None):
        if self._restore or self._tables is None or any(self._ptrs(t["params"]) != t["ptrs"] for t in self._tables):
            assert not self._round_open, "FusedAdamW: gradients moved between step_segment() and step()"
            self._build()
        if not self._round_open:
            self._step += 1
        ids = getattr(self, "_active_ids", None)
        for gi, (group, t) in enumerate(zip(self.param_groups, self._tables)):
            host = [int(p.grad is not None and (ids is None or id(p) in ids)) for p in t["params"]]
            if host != t["active_host"]:
                # (tensors of segments already stepped in this round were all active, before and now: their flags do not move)
                t["active"].copy_(torch.tensor(host, dtype=torch.int32))
                t["active_host"] = host
            if not any(g == gi for g, _ in self._done):
                self._launch(group, t, 0, t["n_chunks"])                # nothing stepped early: the one launch of before
            else:
                for si, (first, count) in enumerate(t["seg_range"]):
                    if (gi, si) not in self._done:
                        self._launch(group, t, first, count)
        self._round_open = False
        self._done.clear()
        self._gate = None
        note_weights_changed()                             # in-place through raw pointers: no _version bump
        return None

    def state_dict(self):
        """torch's format: per parameter ``step`` (its own count), ``exp_avg``, ``exp_avg_sq``."""
        for t in self._tables or []:
            skipped = t["skipped"].cpu().tolist()
            for p, sk in zip(t["params"], skipped):
                self.state[p]["step"] = torch.tensor(float(self._step - sk))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._restore = True


def build_optimizer(name: str, params, device=None, **kwargs):
    if name == "AdamW" and device is not None and torch.device(device).type == "cuda" and not kwargs.get("amsgrad"):
        return FusedAdamW(params, **kwargs)
    return getattr(torch.optim, name)(params, **kwargs)
