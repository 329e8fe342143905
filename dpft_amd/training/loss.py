"""Set-to-set loss with Hungarian assignment, MI355X-native host side.

Mirror of ``src/dprt/training/loss.py`` (focal_loss :17-60, SetCriterion :176-373, Loss :376-564)
and ``src/dprt/training/assigner.py`` (HungarianAnassigner :27-143) with identical values:
  * focal loss keeps the reference's raw-logit ``p_t`` quirk (loss.py:44), alpha 0.75, gamma 2;
  * matcher cost = -logit[gt class] + L1(center) + L1(size) + L1(angle) - GIoU3D (weights from
    ``loss_weights``, GIoU weight 1.0, assigner.py:113-132);
  * ``SetCriterion`` ignores ``train.losses`` / ``loss_inputs`` exactly like the reference (:189-203).
Differences in execution only: GIoU3D comes from the HIP kernel ``dpft_giou3d_yaw_f32`` (exact
yaw-only box geometry instead of pytorch3d.box3d_overlap, src/dprt/utils/iou.py:121-210), and the
B per-sample ``C.cpu()`` syncs (assigner.py:135) are one padded device->host copy per step;
``scipy.optimize.linear_sum_assignment`` still runs on the host (indices must be bit-exact).
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment
from torch import nn

from dpft_amd.hip import ops


def focal_loss(inputs: torch.Tensor, targets: torch.Tensor, alpha: float = 0.75, gamma: float = 2.0,
               reduction: str = "none") -> torch.Tensor:
    ce_loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = inputs * targets + (1 - inputs) * (1 - targets)          # raw logits, as in the reference
    loss = ce_loss * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    return loss


class HungarianAnassigner(nn.Module):
    def __init__(self, loss_weights: Dict[str, float] = None, giou_weight: float = 1.0, **kwargs):
        super().__init__()
        self.loss_weights = loss_weights
        self.giou_weight = giou_weight

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "HungarianAnassigner":
        return cls(loss_weights=config.get("loss_weights"))

    @torch.no_grad()
    def cost_matrices(self, outputs: Dict[str, torch.Tensor], targets: List[Dict[str, torch.Tensor]]):
        """-> padded cost (B, N, Mmax) on the device and the per-sample target counts."""
        B, N = outputs["class"].shape[:2]
        counts = [int(t["gt_class"].shape[0]) for t in targets]
        Mmax = max(max(counts), 1)
        dev = outputs["class"].device
        yaw = torch.atan2(outputs["angle"][..., 0], outputs["angle"][..., 1])
        pred7 = torch.cat((outputs["center"], outputs["size"], yaw[..., None]), -1).contiguous()
        gt7 = torch.zeros((B, Mmax, 7), dtype=pred7.dtype, device=dev)
        gt7[..., 3:6] = 1.0
        for b, t in enumerate(targets):
            m = counts[b]
            if m:
                gyaw = torch.atan2(t["gt_angle"][:, 0], t["gt_angle"][:, 1])
                gt7[b, :m] = torch.cat((t["gt_center"], t["gt_size"], gyaw[:, None]), -1)
        giou = ops.giou3d_yaw(pred7, gt7)                                   # (B,N,Mmax)
        w = self.loss_weights
        cost = torch.zeros((B, N, Mmax), dtype=pred7.dtype, device=dev)
        for b, t in enumerate(targets):
            m = counts[b]
            if not m:
                continue
            gt_ids = torch.argmax(t["gt_class"], dim=-1)
            c = w["total_class"] * (-outputs["class"][b][:, gt_ids]) \
                + w["center"] * torch.cdist(outputs["center"][b], t["gt_center"], p=1) \
                + w["size"] * torch.cdist(outputs["size"][b], t["gt_size"], p=1) \
                + w["angle"] * torch.cdist(outputs["angle"][b], t["gt_angle"], p=1) \
                + self.giou_weight * (-giou[b, :, :m])
            cost[b, :, :m] = c
        return cost, counts

    @torch.no_grad()
    def forward(self, outputs: Dict[str, torch.Tensor], targets: List[Dict[str, torch.Tensor]]):
        """Batched matching: list of (index_i, index_j) int64 device tensors, one pair per sample."""
        cost, counts = self.cost_matrices(outputs, targets)
        host = cost.cpu().numpy()                                            # the one sync of the step
        dev = cost.device
        result = []
        for b, m in enumerate(counts):
            if not m:
                result.append(None)
                continue
            i, j = linear_sum_assignment(host[b, :, :m])
            result.append((torch.as_tensor(np.ascontiguousarray(i), dtype=torch.int64).to(dev, non_blocking=True),
                           torch.as_tensor(np.ascontiguousarray(j), dtype=torch.int64).to(dev, non_blocking=True)))
        return result


class SetCriterion(nn.Module):
    """Per-sample criterion (the reference always calls it with a batch dimension of 1, loss.py:532-540)."""

    def __init__(self):
        super().__init__()
        self.losses = {"total_class": "total_focal_loss", "object_class": "object_focal_loss", "center": "l1_loss",
                       "size": "l1_loss", "angle": "l1_loss"}
        self.loss_inputs = {"total_class": ["class"], "object_class": ["class"], "center": ["center"],
                            "size": ["size"], "angle": ["angle"]}

    @staticmethod
    def total_focal_loss(inputs, targets, i, j):
        N, C = inputs.shape
        M = j.numel()
        one_hot = torch.zeros((N, C), dtype=inputs.dtype, device=inputs.device)
        one_hot[:, 0] = 1.0
        one_hot[i] = targets[:i.numel()]          # scatter_ with src=targets in assignment order (loss.py:305-306);
                                                  # more targets than queries: scatter_ reads the first len(i) rows
        loss = focal_loss(inputs, one_hot, reduction="none")
        return (loss.mean(0).sum() / M) * N

    @staticmethod
    def object_focal_loss(inputs, targets, i, j):
        N = inputs.shape[0]
        M = j.numel()
        loss = focal_loss(inputs[i], targets[j], reduction="none")
        return (loss.mean(0).sum() / M) * N

    @staticmethod
    def l1_loss(inputs, targets, i, j):
        return F.l1_loss(inputs[i], targets[j], reduction="mean")

    def forward(self, inputs: Dict[str, torch.Tensor], targets: Dict[str, torch.Tensor],
                indices: Tuple[torch.Tensor, torch.Tensor]) -> Dict[str, torch.Tensor]:
        i, j = indices
        return {name: getattr(self, fn)(torch.cat([inputs[k] for k in self.loss_inputs[name]], -1),
                                        torch.cat([targets[f"gt_{k}"] for k in self.loss_inputs[name]], -1), i, j)
                for name, fn in self.losses.items()}



def pack_targets(targets: List[Dict[str, torch.Tensor]], counts: List[int], ncls: int, dev):
    """List of per-sample label dicts -> padded device tensors for the HIP loss / metric kernels:
    gt_box (B,Mmax,8) = center | size | angle, gt_onehot (B,Mmax,C), gt_id (B,Mmax) int32, counts (B) int32."""
    from torch.nn.utils.rnn import pad_sequence
    Mmax = max(max(counts), 1)
    if dev.type == "cuda" and len(targets) <= 32:
        fused = _pack_targets_hip(targets, counts, ncls, dev, Mmax)
        if fused is not None:
            return fused
    empty8 = torch.zeros((0, 8), dtype=torch.float32, device=dev)
    emptyc = torch.zeros((0, ncls), dtype=torch.float32, device=dev)
    boxes = [torch.cat((t["gt_center"], t["gt_size"], t["gt_angle"]), -1).float() if m else empty8
             for t, m in zip(targets, counts)]
    gt_box = pad_sequence(boxes, batch_first=True).contiguous()
    gt_onehot = pad_sequence([t["gt_class"].float() if m else emptyc for t, m in zip(targets, counts)],
                             batch_first=True).contiguous()
    if gt_box.shape[1] < Mmax:                      # every sample empty
        gt_box = torch.zeros((len(targets), Mmax, 8), dtype=torch.float32, device=dev)
        gt_onehot = torch.zeros((len(targets), Mmax, ncls), dtype=torch.float32, device=dev)
    gt_id = gt_onehot.argmax(-1).to(torch.int32).contiguous()
    counts_t = torch.tensor(counts, dtype=torch.int32).to(dev, non_blocking=True)
    return gt_box, gt_onehot, gt_id, counts_t, Mmax


def _pack_targets_hip(targets, counts, ncls, dev, Mmax):
    """One launch (dpft_pack_targets_f32) when every label tensor is fp32, contiguous and on the device; else None."""
    import ctypes as C
    from dpft_amd.hip.lib import lib, stream
    B = len(targets)
    keys = ("gt_center", "gt_size", "gt_angle", "gt_class")
    widths = (3, 3, 2, ncls)
    ptrs = [(C.c_void_p * B)() for _ in keys]
    for b, (t, m) in enumerate(zip(targets, counts)):
        for arr, k, w in zip(ptrs, keys, widths):
            v = t[k]
            if m == 0:
                arr[b] = None
                continue
            if not (v.is_cuda and v.dtype == torch.float32 and v.is_contiguous() and tuple(v.shape) == (m, w)):
                return None
            arr[b] = v.data_ptr()
    gt_box = torch.empty((B, Mmax, 8), dtype=torch.float32, device=dev)
    gt_onehot = torch.empty((B, Mmax, ncls), dtype=torch.float32, device=dev)
    gt_id = torch.empty((B, Mmax), dtype=torch.int32, device=dev)
    counts_t = torch.empty((B,), dtype=torch.int32, device=dev)
    ch = (C.c_int32 * B)(*counts)
    lib.call("dpft_pack_targets_f32", ptrs[0], ptrs[1], ptrs[2], ptrs[3], ch, B, Mmax, ncls, gt_box.data_ptr(),
             gt_onehot.data_ptr(), gt_id.data_ptr(), counts_t.data_ptr(), stream())
    return gt_box, gt_onehot, gt_id, counts_t, Mmax


# ticket + partial sums of dpft_set_loss_fwd_total_f32 (zero between launches): one buffer per (device, stream) -- two launches in
# flight on different streams (train and validation criteria, several losses) must not share a ticket
_loss_scratch: Dict[tuple, torch.Tensor] = {}


def _set_loss_launch(cls, center, size, angle, gt_box, gt_onehot, match, counts, weights5, alpha, sel):
    """(losses5, total) of dpft_set_loss_fwd_total_f32 -- shared by the autograd Function and the trainer's direct chain."""
    import ctypes as C
    from dpft_amd.hip.lib import lib, stream
    B, N, ncls = cls.shape
    Mmax = gt_box.shape[1]
    losses = torch.empty(5, dtype=torch.float32, device=cls.device)
    total = torch.empty((), dtype=torch.float32, device=cls.device)
    w = (C.c_float * 5)(*weights5)
    # one launch: the five terms (per-block partial sums added in block order by the last block) and the total of the
    # configured ones -- instead of a cleared output + atomics + a dot product, and of select x5 / stack / sum in
    # autograd, whose backward alone is a dozen tiny launches between the step's two host syncs
    need = int(lib.dpft_set_loss_scratch_floats(B, N))
    skey = (cls.device, int(torch.cuda.current_stream(cls.device).cuda_stream))
    scratch = _loss_scratch.get(skey)
    if scratch is None or scratch.numel() < need:
        scratch = _loss_scratch[skey] = torch.zeros(max(need, 1024), dtype=torch.float32, device=cls.device)
    try:
        lib.call("dpft_set_loss_fwd_total_f32", cls.data_ptr(), center.data_ptr(), size.data_ptr(), angle.data_ptr(),
                 gt_box.data_ptr(), gt_onehot.data_ptr(), match.data_ptr(), counts.data_ptr(), C.byref(w), float(alpha),
                 sel.data_ptr(), scratch.data_ptr(), losses.data_ptr(), total.data_ptr(), B, N, Mmax, ncls, stream())
    except Exception:
        scratch.zero_()      # a launch that did not complete may leave its ticket behind
        raise
    return losses, total


class _SetLossFn(torch.autograd.Function):
    """dpft_set_loss_fwd/bwd_f32: the five batch-reduced, weighted criterion terms for fixed assignments."""

    @staticmethod
    def forward(ctx, cls, center, size, angle, gt_box, gt_onehot, match, counts, weights5, alpha, sel, pre=None):
        # pre = (losses5, total) already computed by dpft_assign_loss_f32 (Loss.forward_fused): no launch here
        losses, total = pre if pre is not None else \
            _set_loss_launch(cls, center, size, angle, gt_box, gt_onehot, match, counts, weights5, alpha, sel)
        ctx.save_for_backward(cls, center, size, angle, gt_box, gt_onehot, match, counts, sel)
        ctx.meta = (weights5, float(alpha))
        ctx.mark_non_differentiable(losses)
        return losses, total

    @staticmethod
    def backward(ctx, _glosses, gtotal):
        import ctypes as C
        from dpft_amd.hip.lib import lib, stream
        cls, center, size, angle, gt_box, gt_onehot, match, counts, sel = ctx.saved_tensors
        weights5, alpha = ctx.meta
        B, N, ncls = cls.shape
        Mmax = gt_box.shape[1]
        gout = (sel * gtotal).contiguous().float()          # d total / d term = 1 for the configured terms
        dcls, dcenter, dsize, dangle = (torch.empty_like(t) for t in (cls, center, size, angle))
        w = (C.c_float * 5)(*weights5)
        lib.call("dpft_set_loss_bwd_f32", cls.data_ptr(), center.data_ptr(), size.data_ptr(), angle.data_ptr(),
                 gt_box.data_ptr(), gt_onehot.data_ptr(), match.data_ptr(), counts.data_ptr(), C.byref(w), alpha,
                 gout.data_ptr(), dcls.data_ptr(), dcenter.data_ptr(), dsize.data_ptr(), dangle.data_ptr(), B, N, Mmax,
                 ncls, stream())
        return dcls, dcenter, dsize, dangle, None, None, None, None, None, None, None, None


class Loss(nn.modules.loss._Loss):
    def __init__(self, anassigner: nn.Module = None, criterion: nn.Module = None, loss_weights: Dict[str, float] = None,
                 reduction: str = "mean", **kwargs):
        super().__init__()
        if reduction not in {"none", "mean", "sum"}:
            raise ValueError(f"Invalid Value for arg 'reduction': '{reduction}'")
        if anassigner is None or criterion is None:
            raise ValueError("dpft_amd Loss: the hot path is HungarianAnassigner + SetCriterion (every reference config)")
        self.anassigner, self.criterion = anassigner, criterion
        self.loss_weights = loss_weights if loss_weights is not None else {}
        self.reduction = reduction

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "Loss":
        if "hungarian" not in config.get("anassigner", "").lower() or config.get("criterion") != "SetCriterion":
            raise ValueError("dpft_amd Loss supports anassigner=HungarianAnassigner, criterion=SetCriterion")
        return cls(anassigner=HungarianAnassigner.from_config(config), criterion=SetCriterion(),
                   loss_weights=config.get("loss_weights"), reduction=config.get("reduction", "mean"))

    use_fused = True      # CUDA: matcher cost, criterion and its gradient from 3 HIP kernels (dpft_amd/csrc/misc.hip)
    _TERMS = ("total_class", "object_class", "center", "size", "angle")

    def _fused_ok(self, inputs) -> bool:
        return (self.use_fused and inputs["class"].is_cuda and self.reduction == "mean"
                and isinstance(self.anassigner, HungarianAnassigner) and isinstance(self.criterion, SetCriterion)
                and set(self.loss_weights) <= set(self._TERMS) and inputs["class"].dtype == torch.float32)

    # Assignments on the device (csrc/lsap.hip) instead of a read-back + host assignment.  Off by default for a stand-alone Loss:
    # scipy raises ValueError for a cost matrix with NaN / inf entries AT the call, a kernel can only leave a status word; the
    # trainer switches it on and calls check_assignment_status() where it reads values back anyway (logging, epoch end).
    assign_on_device = False

    def _status_word(self, dev) -> torch.Tensor:
        """One page-locked int32 the assignment kernel writes its error code to (device-visible host memory: the host can look at it
        without a sync; it only ever changes on an error)."""
        st = self.__dict__.get("_lsap_status")
        if st is None:
            st = self.__dict__["_lsap_status"] = torch.zeros(1, dtype=torch.int32).pin_memory()
        return st

    def assignment_status(self, sync: bool = True) -> int:
        """The on-device assignment's status word (0 = fine), read and cleared.  ``check_assignment_status`` raises from it; the
        trainer MAX-reduces it over the ranks first so that all of them raise in the same place."""
        st = self.__dict__.get("_lsap_status")
        if st is None:
            return 0
        if sync:
            torch.cuda.synchronize()
        code = int(st[0])
        if code:
            st[0] = 0
        return code

    @staticmethod
    def describe_assignment_status(code: int) -> str:
        what = "contains invalid numeric entries" if code < 0x10000 else ("is infeasible" if code < 0x20000 else
                                                                          "has more targets than the packed width")
        return f"matcher: cost matrix of sample {(code & 0xffff) - (1 if code < 0x10000 else 0)} {what}"

    def check_assignment_status(self, sync: bool = True) -> None:
        """Raise the ValueError scipy would have raised in the step whose cost matrix was not finite (or infeasible)."""
        code = self.assignment_status(sync)
        if code:
            raise ValueError(self.describe_assignment_status(code))

    def _to_host(self, t: torch.Tensor) -> "np.ndarray":
        """Device tensor -> numpy through a reused pinned buffer (one async copy + one stream sync instead of a pageable
        allocation and a staged copy: this read-back is the step's host sync, the GPU idles until the host moves on)."""
        n = t.numel()
        pin = self.__dict__.get("_pin_f32")
        if pin is None or pin.numel() < n:
            pin = self.__dict__["_pin_f32"] = torch.empty(max(n, 1 << 16), dtype=torch.float32).pin_memory()
        view = pin[:n].view(t.shape)
        view.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return view.numpy()

    def forward_fused(self, inputs: Dict[str, torch.Tensor], targets: List[Dict[str, torch.Tensor]]):
        """Same values as ``forward_eager`` from three launches + the host assignment (one D2H, one H2D copy)."""
        import ctypes as C
        from dpft_amd.hip.lib import lib, stream
        cls, center = inputs["class"].contiguous(), inputs["center"].contiguous()
        size, angle = inputs["size"].contiguous(), inputs["angle"].contiguous()
        B, N, ncls = cls.shape
        dev = cls.device
        counts = [int(t["gt_class"].shape[0]) if all(v.numel() for v in t.values()) else 0 for t in targets]
        # host-known: whether this batch has any target.  Without one the loss is exactly 0 (below); with one it is positive (the
        # focal term alone), so the trainer takes its `loss > 0` decision (trainer.py:131) from this flag instead of a read-back
        self.__dict__["last_has_targets"] = max(counts) > 0
        if max(counts) == 0:
            zero = torch.zeros((), device=dev, dtype=cls.dtype, requires_grad=True)
            return zero * 1.0, {k: zero for k in self.loss_weights}
        gt_box, gt_onehot, gt_id, counts_t, Mmax = pack_targets(targets, counts, ncls, dev)
        aw = self.anassigner.loss_weights
        cw = (C.c_float * 5)(aw["total_class"], aw["center"], aw["size"], aw["angle"], self.anassigner.giou_weight)
        cost = torch.empty((B, N, Mmax), dtype=torch.float32, device=dev)
        with torch.no_grad():
            lib.call("dpft_match_cost_f32", cls.data_ptr(), center.data_ptr(), size.data_ptr(), angle.data_ptr(),
                     gt_box.data_ptr(), gt_id.data_ptr(), counts_t.data_ptr(), C.byref(cw), cost.data_ptr(), B, N, Mmax,
                     ncls, stream())
        hook = self.__dict__.get("_cost_hook")      # (bench.py: an event behind the cost kernel, the start of the loss window)
        if hook is not None:
            hook()
        # (everything the host window needs is prepared BEFORE the sync, while the GPU still runs the cost kernel: after the
        # read-back only the one C call is left)
        n_pack = B * Mmax * 2 + B
        pin = self.__dict__.get("_pin_i32")
        if pin is None or pin.numel() < n_pack:
            pin = self.__dict__["_pin_i32"] = torch.empty(max(n_pack, 4096), dtype=torch.int32).pin_memory()
        packed = pin.numpy()[:n_pack]                                # assignments | pair counts: ONE upload, from pinned memory
        weights5 = tuple(float(self.loss_weights.get(k, 0.0)) for k in self._TERMS)
        sel = self.__dict__.get("_sel")
        if sel is None or sel.device != dev:
            sel = self.__dict__["_sel"] = torch.tensor([1.0 if k in self.loss_weights else 0.0 for k in self._TERMS],
                                                       dtype=torch.float32, device=dev)
        packed_t = torch.empty(n_pack, dtype=torch.int32, device=dev)
        match_t, counts_m = packed_t[:B * Mmax * 2].view(B, Mmax, 2), packed_t[B * Mmax * 2:]
        self.__dict__["_bwd_written"] = None
        use_c = os.environ.get("DPFT_LSAP_C", "1") != "0"
        on_dev = use_c and self.assign_on_device and os.environ.get("DPFT_LSAP_DEV", "1") != "0"
        if use_c:
            cnt = np.asarray(counts, dtype=np.int32)
            losses5 = torch.empty(5, dtype=torch.float32, device=dev)
            total = torch.empty((), dtype=torch.float32, device=dev)
            need = int(lib.dpft_set_loss_scratch_floats(B, N))
            st_ = stream()
            skey = (dev, int(st_.value or 0))
            scratch = _loss_scratch.get(skey)
            if scratch is None or scratch.numel() < need:
                scratch = _loss_scratch[skey] = torch.zeros(max(need, 1024), dtype=torch.float32, device=dev)
            tg = self.__dict__.get("fused_grad_targets")            # (dcenter, dsize, dangle, dcls) or None
            ok_t = tg is not None and all(t.shape == r.shape and t.is_contiguous() and t.dtype == torch.float32
                                          for t, r in zip(tg, (center, size, angle, cls)))
            w5 = (C.c_float * 5)(*weights5)
            if on_dev:
                self.check_assignment_status(sync=False)      # (an error of an earlier step that has finished by now)
                head = (cost.data_ptr(), counts_t.data_ptr(), packed_t.data_ptr(), self._status_word(dev).data_ptr())
            else:
                head = (cnt.ctypes.data, packed.ctypes.data, packed_t.data_ptr())
            c_args = head + (
                      cls.data_ptr(), center.data_ptr(), size.data_ptr(), angle.data_ptr(),
                      gt_box.data_ptr(), gt_onehot.data_ptr(), C.cast(w5, C.c_void_p), 0.75, sel.data_ptr(),
                      scratch.data_ptr(), losses5.data_ptr(), total.data_ptr(),
                      tg[3].data_ptr() if ok_t else None, tg[0].data_ptr() if ok_t else None,
                      tg[1].data_ptr() if ok_t else None, tg[2].data_ptr() if ok_t else None,
                      B, N, Mmax, ncls, st_)
            assign_loss = lib.dpft_assign_loss_dev_f32 if on_dev else lib.dpft_assign_loss_f32
        if on_dev:
            # No host round trip at all (dpft_assign_loss_dev_f32, csrc/lsap.hip): the assignments are computed by one wavefront per
            # sample from the cost matrices where the cost kernel left them -- the same step sequence as the host code, the same
            # pairs (tests/test_gpu_lsap.py) --, criterion and gradient launches follow in the same call.  The step has no host
            # sync left; a non-finite cost matrix is reported through the status word (check_assignment_status) instead of here.
            try:
                rc = assign_loss(*c_args)
            except Exception:
                scratch.zero_()
                raise
            if rc != 0:
                scratch.zero_()
                raise ValueError("assignment / loss failed: " + lib.dpft_last_error().decode())
            if ok_t:
                self.__dict__["_bwd_written"] = tuple(tg)
            losses5, total = _SetLossFn.apply(cls, center, size, angle, gt_box, gt_onehot, match_t, counts_m, weights5, 0.75, sel,
                                              (losses5, total))
            self.__dict__["_last"] = (cls, center, size, angle, gt_box, gt_onehot, match_t, counts_m, weights5, 0.75, sel, total)
            return total, {k: losses5[self._TERMS.index(k)] for k in self.loss_weights}
        host = self._to_host(cost)                                         # the host path's sync (assign_on_device has none)
        if use_c:
            # The whole host window in ONE C call (dpft_assign_loss_f32, csrc/cabi.cpp): the batch's assignments (scipy's algorithm
            # restated in C, same pairs in the same order -- tests/test_host.py), their upload from the pinned buffer (rewritten only
            # after the next step's sync), the criterion launch and -- when the trainer has named the buffers the decoder's backward
            # graph reads (fused_grad_targets) -- the criterion's gradient launch.  This window is the one place of the step where
            # the GPU waits for the host.
            try:
                rc = assign_loss(host.ctypes.data, *c_args)
            except Exception:
                scratch.zero_()
                raise
            if rc != 0:
                scratch.zero_()      # a launch that did not complete may leave its ticket behind
                raise ValueError("assignment / loss failed: " + lib.dpft_last_error().decode())      # (scipy raises ValueError as well)
            if ok_t:
                self.__dict__["_bwd_written"] = tuple(tg)
            losses5, total = _SetLossFn.apply(cls, center, size, angle, gt_box, gt_onehot, match_t, counts_m, weights5, 0.75, sel,
                                              (losses5, total))
        else:      # A/B switch: scipy per sample, separate launches
            packed[:] = -1
            match = packed[:B * Mmax * 2].reshape(B, Mmax, 2)
            for b, m in enumerate(counts):
                if m:
                    i, j = linear_sum_assignment(host[b, :, :m])
                    match[b, :len(i), 0], match[b, :len(i), 1] = i, j
                    counts[b] = len(i)           # min(N, m) assigned pairs
            packed[B * Mmax * 2:] = counts
            packed_t.copy_(pin[:n_pack], non_blocking=True)
            losses5, total = _SetLossFn.apply(cls, center, size, angle, gt_box, gt_onehot, match_t, counts_m, weights5, 0.75, sel)
        self.__dict__["_last"] = (cls, center, size, angle, gt_box, gt_onehot, match_t, counts_m, weights5, 0.75, sel, total)
        batch_losses = {k: losses5[self._TERMS.index(k)] for k in self.loss_weights}        # views, for logging
        return total, batch_losses

    def backward_into(self, total: torch.Tensor, dcenter, dsize, dangle, dcls) -> bool:
        """d total / d (center, size, angle, class) of the LAST fused forward, written into the given buffers by the one
        backward launch (what ``total.backward()`` would hand to the producers of the outputs, without autograd)."""
        import ctypes as C
        from dpft_amd.hip.lib import lib, stream
        last = self.__dict__.pop("_last", None)
        if last is None or last[-1] is not total:
            return False
        cls, center, size, angle, gt_box, gt_onehot, match, counts, weights5, alpha, sel, _ = last
        written = self.__dict__.pop("_bwd_written", None)
        if written is not None and all(a is b for a, b in zip(written, (dcenter, dsize, dangle, dcls))):
            return True      # dpft_assign_loss_f32 launched the gradient kernel into exactly these buffers
        B, N, ncls = cls.shape
        w = (C.c_float * 5)(*weights5)
        for t, ref in ((dcenter, center), (dsize, size), (dangle, angle), (dcls, cls)):
            assert t.shape == ref.shape and t.is_contiguous() and t.dtype == torch.float32
        lib.call("dpft_set_loss_bwd_f32", cls.data_ptr(), center.data_ptr(), size.data_ptr(), angle.data_ptr(),
                 gt_box.data_ptr(), gt_onehot.data_ptr(), match.data_ptr(), counts.data_ptr(), C.byref(w), alpha,
                 sel.data_ptr(), dcls.data_ptr(), dcenter.data_ptr(), dsize.data_ptr(), dangle.data_ptr(), B, N,
                 gt_box.shape[1], ncls, stream())
        return True

    def forward(self, inputs: Dict[str, torch.Tensor], targets: List[Dict[str, torch.Tensor]]):
        self.__dict__.pop("_last", None)
        self.__dict__["last_has_targets"] = None      # (None = unknown: the eager path)
        if self._fused_ok(inputs):
            return self.forward_fused(inputs, targets)
        return self.forward_eager(inputs, targets)

    def forward_eager(self, inputs: Dict[str, torch.Tensor], targets: List[Dict[str, torch.Tensor]]):
        """-> (total_loss, {name: batch-reduced loss}) exactly like loss.py:486-564."""
        dev, dtype = inputs["class"].device, inputs["class"].dtype
        matches = self.anassigner(inputs, targets)
        per_sample = []
        for b, (tgt, match) in enumerate(zip(targets, matches)):
            if match is None or not all(t.numel() for t in tgt.values()):
                per_sample.append({k: torch.zeros((), device=dev, dtype=dtype, requires_grad=True)
                                   for k in self.loss_weights})
                continue
            inp = {k: v[b] for k, v in inputs.items()}
            losses = self.criterion(inp, tgt, match)
            per_sample.append({k: losses[k] * w for k, w in self.loss_weights.items()})
        batch_losses = {k: torch.stack([p[k] for p in per_sample]) for k in self.loss_weights}
        if self.reduction != "none":
            batch_losses = {k: getattr(torch, self.reduction)(v) for k, v in batch_losses.items()}
        total = torch.stack(tuple(batch_losses.values())).sum(dim=-1 if self.reduction != "none" else 0)
        return total, batch_losses


def build_loss(*args, **kwargs):
    return Loss.from_config(*args, **kwargs)
