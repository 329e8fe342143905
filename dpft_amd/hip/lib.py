"""ctypes binding of libdpft_hip.so -- one thin Python function per C-ABI entry point
(include/dpft_hip.h).  No torch types cross the boundary: tensors are passed as raw device
pointers, the stream as the raw hipStream_t of torch's current stream.

There is deliberately NO fallback: a missing library or a failing call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libdpft_hip.so")
MAX_LEVELS = 8


class HipLibraryError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "H", "W", "C", "K", "kh", "kw", "stride", "pad", "OH", "OW", "act16")] + \
               [("a_planes", C.c_void_p), ("w_planes", C.c_void_p)]


class Pyramid(C.Structure):
    _fields_ = [("level", C.c_void_p * MAX_LEVELS), ("grad", C.c_void_p * MAX_LEVELS),
                ("H", C.c_int32 * MAX_LEVELS), ("W", C.c_int32 * MAX_LEVELS), ("L", C.c_int32),
                ("grad_replicas", C.c_int32 * MAX_LEVELS)]


class ResnetDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("in_channels", C.c_int32),
                ("depths", C.c_int32 * 4), ("n_layers", C.c_int32), ("eps", C.c_float), ("momentum", C.c_float),
                ("act16", C.c_int32)]


class ResnetTables(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_void_p)) for n in
                ("conv_w", "conv_dw", "bn_gamma", "bn_beta", "bn_rm", "bn_rv", "bn_dgamma", "bn_dbeta")]


class DecoderView(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "norm1_w", "norm1_b",
        "off_w", "off_b", "att_w", "att_b", "val_w", "val_b", "outp_w", "outp_b", "norm2_w", "norm2_b",
        "ffn1_w", "ffn1_b", "ffn2_w", "ffn2_b", "norm3_w", "norm3_b")]


class SaParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("in_w", "in_b", "out_w", "out_b", "n1_w", "n1_b")]


class HeadTrain(C.Structure):
    _fields_ = [("y3", C.c_void_p), ("packed", C.c_void_p), ("red_w", C.c_void_p), ("head_w", (C.c_void_p * 3) * 4),
                ("prev_center", C.c_void_p), ("T", C.c_void_p * 4), ("P", C.c_void_p * 4), ("shape", C.c_void_p * 4),
                ("p_rows", C.c_int32 * 4), ("has_t", C.c_int32 * 4), ("num_classes", C.c_int32),
                ("x", C.c_void_p), ("center", C.c_void_p), ("size", C.c_void_p), ("angle", C.c_void_p),
                ("cls", C.c_void_p), ("refs", C.c_void_p),
                ("dx", C.c_void_p), ("dcenter", C.c_void_p), ("dsize", C.c_void_p), ("dangle", C.c_void_p),
                ("dcls", C.c_void_p), ("drefs", C.c_void_p),
                ("dy3", C.c_void_p), ("dcenter_prev", C.c_void_p), ("rows", C.c_void_p), ("shape_stride", C.c_int32)]


class DecoderFwd(C.Structure):
    _fields_ = [("B", C.c_int32), ("Q", C.c_int32), ("V", C.c_int32), ("iters", C.c_int32), ("num_classes", C.c_int32),
                ("n_points", C.c_int32 * 4), ("packed_views", C.c_void_p), ("packed_heads", C.c_void_p),
                ("pyr", C.c_void_p), ("query0", C.c_void_p), ("pos", C.c_void_p), ("center0", C.c_void_p),
                ("T", C.c_void_p * 4), ("P", C.c_void_p * 4), ("shape", C.c_void_p * 4),
                ("p_rows", C.c_int32 * 4), ("has_t", C.c_int32 * 4), ("work", C.c_void_p),
                ("center", C.c_void_p), ("size", C.c_void_p), ("angle", C.c_void_p), ("cls", C.c_void_p),
                ("attn0", C.c_void_p), ("shape_stride", C.c_int32)]


class MemOp(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("bytes", C.c_uint64)]


class SumSrc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("n_lead", C.c_int32)]


class OuterSpec(C.Structure):
    _fields_ = [("col_a", C.c_int32), ("n_a", C.c_int32), ("col_b", C.c_int32), ("n_b", C.c_int32), ("out_off", C.c_int64)]


_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_DESC = C.POINTER(ConvDesc)
_PYR = C.POINTER(Pyramid)

# name -> (restype, argtypes); must list every symbol include/dpft_hip.h declares
SIGNATURES = {
    "dpft_version": (_I, []),
    "dpft_last_error": (C.c_char_p, []),
    "dpft_conv2d_workspace_bytes": (_L, [_DESC]),
    "dpft_conv2d_workspace_header_bytes": (_L, []),
    "dpft_conv2d_workspace_init": (_I, [_P, _P]),
    "dpft_split_planes_f32": (_I, [_P, _P, _L, _P]),
    "dpft_conv2d_stats_tiles": (_I, [_DESC, C.POINTER(_I)]),
    "dpft_conv2d_stats_tiles_pro": (_I, [_DESC, _I, C.POINTER(_I)]),
    "dpft_conv_set_compute": (_I, [_I]),
    "dpft_conv_get_compute": (_I, []),
    "dpft_conv_set_split": (_I, [_I]),
    "dpft_conv_get_split": (_I, []),
    "dpft_conv2d_nhwc_fwd_f32": (_I, [_DESC, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "dpft_conv2d_nhwc_fwd_bnact_f32": (_I, [_DESC, _P, _P, _P, _I, _P, _P, _P, _P]),
    "dpft_conv2d_nhwc_dgrad_f32": (_I, [_DESC, _P, _P, _P, _I, _P, _P]),
    "dpft_conv2d_nhwc_dgrad_bn_reduce_f32": (_I, [_DESC, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "dpft_conv2d_nhwc_wgrad_f32": (_I, [_DESC, _P, _P, _P, _I, _P, _P, _P]),
    "dpft_conv2d_nhwc_wgrad_bias_f32": (_I, [_P, _P, _P, _P, _P, _P, _P]),
    "dpft_weight_transpose_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "dpft_weight_transpose_batch_f32": (_I, [_I, _P, _P, _P, _P, _P, _P]),
    "dpft_bias_grad_f32": (_I, [_P, _P, _L, _I, _P]),
    "dpft_bn_stats_f32": (_I, [_P, _P, _L, _I, _I, _P]),
    "dpft_bn_finalize_f32": (_I, [_P, _I, _I, _L, _I, _P, _P, _F, _F, _P, _P, _P, _P]),
    "dpft_bn_eval_params_f32": (_I, [_P, _P, _P, _P, _F, _I, _P, _P]),
    "dpft_bn_act_f32": (_I, [_P, _P, _P, _P, _I, _P, _L, _I, _P]),
    "dpft_bn_relu_maxpool_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dpft_bn_relu_maxpool_bwd_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dpft_bn_bwd_reduce_f32": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "dpft_bn_bwd_apply_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "dpft_relu_bwd_f32": (_I, [_P, _P, _P, _L, _P]),
    "dpft_add_inplace_f32": (_I, [_P, _P, _L, _P]),
    "dpft_fpn_topdown_add_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dpft_fpn_topdown_add_bwd_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dpft_add_pos_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dpft_fpn_lateral_f32": (_I, [_DESC, _P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "dpft_fpn_output_f32": (_I, [_DESC, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dpft_msda_fwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dpft_msda_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dpft_xattn_fwd_f32": (_I, [_PYR, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dpft_xattn_bwd_f32": (_I, [_PYR, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dpft_giou3d_yaw_f32": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "dpft_decoder_packed_view_floats": (_L, []),
    "dpft_decoder_packed_head_floats": (_L, []),
    "dpft_decoder_packed_infer_floats": (_L, [_I]),
    "dpft_decoder_pack_infer_f32": (_I, [C.POINTER(DecoderView), _I, _I, _P, _P, _I, _I, _P, _P, _I, _P, _P]),
    "dpft_decoder_pack_view_f32": (_I, [C.POINTER(DecoderView), _I, _I, _P, _P]),
    "dpft_decoder_pack_views_f32": (_I, [_P, _I, _P, _P, _P, _P]),
    "dpft_decoder_pack_head_f32": (_I, [_P, C.POINTER(C.c_void_p * 12), _I, _I, _P, _P]),
    "dpft_decoder_work_floats": (_L, [_I, _I, _I]),
    "dpft_debug_decoder_stamps": (_I, [_P]),
    "dpft_selfattn_train_fwd_f32": (_I, [_P, _I, _P, _L, _P, _F, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P]),
    "dpft_selfattn_train_bwd_f32": (_I, [_P, _I, _P, _L, _P, _F, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "dpft_selfattn_train_scratch_floats": (_L, [_I, _I, _I]),
    "dpft_xattn_ffn_train_row_floats": (_L, []),
    "dpft_head_train_row_floats": (_L, []),
    "dpft_head_train_fwd_f32": (_I, [C.POINTER(HeadTrain), _I, _I, _I, _P]),
    "dpft_head_train_bwd_f32": (_I, [C.POINTER(HeadTrain), _I, _I, _I, _P]),
    "dpft_xattn_ffn_train_saved_floats": (_L, []),
    "dpft_xattn_ffn_train_scratch_floats": (_L, []),
    "dpft_xattn_ffn_train_fwd_f32": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _F, _P, _I, _P, _P, _I, _I, _P]),
    "dpft_xattn_ffn_train_bwd_f32": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _F, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "dpft_decoder_forward_f32": (_I, [C.POINTER(DecoderFwd), _P]),
    "dpft_decoder_attn0_f32": (_I, [_P, _P, _I, _I, _P, _P]),
    "dpft_pack_targets_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "dpft_match_cost_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, C.POINTER(_F * 5), _P, _I, _I, _I, _I, _P]),
    "dpft_set_loss_fwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(_F * 5), _F, _P, _I, _I, _I, _I, _P]),
    "dpft_set_loss_scratch_floats": (_L, [_I, _I]),
    "dpft_set_loss_fwd_total_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(_F * 5), _F, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dpft_set_loss_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(_F * 5), _F, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dpft_resize_bilinear_nhwc_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dpft_resize_bilinear_nhwc_u8": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dpft_scale_clip_f32": (_I, [_P, _P, _L, _F, _F, _F, _F, _P]),
    "dpft_detection_metrics_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _I, _P, _P, _I, _I, _I, _I, _P]),
    "dpft_radar_projection_scratch_floats": (_L, [_I, _I, _I, _I]),
    "dpft_export_select_f32": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "dpft_radar_projection_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dpft_profile_start": (_I, []),
    "dpft_profile_stop": (_I, []),
    "dpft_profile_serialize": (_I, [_I]),
    "dpft_profile_overhead_ms": (_F, []),
    "dpft_profile_get": (_I, [_I, C.POINTER(_I), C.POINTER(C.c_double), C.POINTER(_F), C.POINTER(_I * 7)]),
    "dpft_profile_get_family": (_I, [_I, C.POINTER(_I)]),
    "dpft_rows_outer_f32": (_I, [_P, _I, _I, _I, _P, _I, _P, _L, _P]),
    "dpft_memops": (_I, [_I, _P, _P]),
    "dpft_lsap_batch_f32": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "dpft_assign_loss_f32": (_I, [_P] * 11 + [_F] + [_P] * 8 + [_I] * 4 + [_P]),
    "dpft_lsap_batch_dev_f32": (_I, [_P] * 5 + [_I] * 3 + [_P]),
    "dpft_assign_loss_dev_f32": (_I, [_P] * 11 + [_F] + [_P] * 8 + [_I] * 4 + [_P]),
    "dpft_add_many_f32": (_I, [_I, _P, _P]),
    "dpft_i64_add_many": (_I, [_I, _P, _L, _P]),
    "dpft_seed_advance": (_I, [_P, _P, _L, _P]),
    "dpft_sum_leading_f32": (_I, [_I, _P, _L, _P, _I, _P]),
    "dpft_adamw_f32": (_I, [_P, _I, _P, _P, _F, _F, _F, _F, _F, _I, _P, _P]),
    "dpft_resnet_plan_create": (_L, [C.POINTER(ResnetDesc)]),
    "dpft_resnet_plan_destroy": (None, [_L]),
    "dpft_resnet_plan_query": (_L, [_L, _I, _I]),
    "dpft_resnet_forward": (_I, [_L, _P, C.POINTER(ResnetTables), _P, _I, _P]),
    "dpft_resnet_backward_stage": (_I, [_L, _I, _P, C.POINTER(ResnetTables), _P, _P, _I, _P]),
    "dpft_resnet_plan_set_side_stream": (_I, [_L, _P]),
    "dpft_resnet_plan_set_graph": (_I, [_L, _I]),
    "dpft_stream_set": (_I, [_P, _I, C.POINTER(C.c_void_p)]),
}


class _Lib:
    """Lazy loader so that importing the package (host logic, state-dict plumbing) works on a
    machine where the library has not been built; any *use* without it raises."""

    def __init__(self):
        self._dll = None

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise HipLibraryError(
                    f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "or `make -C dpft_amd/csrc`. dpft_amd has no CPU/PyTorch fallback.")
            dll = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(dll, name)
                fn.restype, fn.argtypes = res, args
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        return getattr(self.load(), name)

    def call(self, name: str, *args):
        rc = getattr(self.load(), name)(*args)
        if rc != 0:
            msg = self._dll.dpft_last_error().decode("utf-8", "replace")
            raise HipLibraryError(f"{name} failed (code {rc}): {msg}")


lib = _Lib()


def ptr(t: Optional[torch.Tensor]):
    """Raw device pointer of a contiguous fp32/int64 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipLibraryError("dpft_amd ops need CUDA (ROCm) tensors; there is no CPU path")
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """The current HIP stream of the current device as a C pointer (asked ~100 times per training step: the raw query, not a
    torch.cuda.Stream object per call -- 0.3 vs 8 us)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Kernels that update parameters through raw pointers (dpft_adamw_f32) do not touch torch's per-tensor version
# counters; anything that caches a function of the weights (the inference decoder's packed blobs) keys on this too.
_weights_generation = 0


def note_weights_changed() -> None:
    global _weights_generation
    _weights_generation += 1


def weights_generation() -> int:
    return _weights_generation


def make_desc(B, H, W, Cin, K, kh, kw, stride, pad) -> ConvDesc:
    OH = (H + 2 * pad - kh) // stride + 1
    OW = (W + 2 * pad - kw) // stride + 1
    return ConvDesc(B, H, W, Cin, K, kh, kw, stride, pad, OH, OW)


def make_pyramid(levels: Sequence[torch.Tensor], grads: Optional[Sequence[torch.Tensor]] = None) -> Pyramid:
    """grads[i] with one more leading dimension than levels[i] is a replicated buffer (R,B,H,W,C)."""
    p = Pyramid()
    p.L = len(levels)
    for i, l in enumerate(levels):
        assert l.is_contiguous() and l.dtype == torch.float32
        p.level[i] = l.data_ptr()
        p.H[i], p.W[i] = l.shape[1], l.shape[2]
        p.grad[i] = grads[i].data_ptr() if grads is not None else None
        p.grad_replicas[i] = grads[i].shape[0] if grads is not None and grads[i].dim() == l.dim() + 1 else 0
    return p


_stream_sets = {}


def stream_set(device, n: int = 3):
    """``n`` torch streams on hardware queues distinct from the current (main) stream's and from each other (as far as the
    device has queues) -- dpft_stream_set; cached per device.  Used for the view encoders' streams and the camera
    encoder's weight-gradient stream (dpft_amd/models/dprt.py)."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), n)
    got = _stream_sets.get(key)
    if got is None:
        arr = (C.c_void_p * n)()
        with torch.cuda.device(key[0]):
            distinct = int(lib.load().dpft_stream_set(stream(), n, arr))
            if distinct < 0:
                raise HipLibraryError("dpft_stream_set failed: " + lib.dpft_last_error().decode("utf-8", "replace"))
            got = _stream_sets[key] = ([torch.cuda.ExternalStream(arr[i], device=key[0]) for i in range(n)], distinct)
    return got
