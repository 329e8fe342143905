"""Drop-in for the reference's only native seam: ``import MultiScaleDeformableAttention as MSDA``
(/root/reference/src/dprt/models/layers/ms_deform_attn.py:24), called as ``MSDA.ms_deform_attn_forward`` (:32-39) and
``MSDA.ms_deform_attn_backward`` (:58-66) -- upstream that is the Deformable-DETR CUDA extension.

Put this directory on PYTHONPATH in place of the CUDA extension and the reference file runs unchanged on MI355X: the two
functions bind ``dpft_msda_fwd_f32`` / ``dpft_msda_bwd_f32`` of libdpft_hip.so (include/dpft_hip.h) through ctypes --
plain pointers and sizes, the caller's current stream, no torch types in the C signatures.  The library is looked up in
$DPFT_HIP_LIB, next to the dpft_amd package of this checkout, then on the loader path; there is NO fallback: a missing
library or a failed launch raises.

Argument meaning and error behaviour follow the extension: fp32 CUDA(HIP) tensors, value (N, S, M, D), spatial_shapes
(L, 2) int64 rows (H, W), level_start_index (L,) int64, sampling_loc (N, Lq, M, L, P, 2) in [0, 1] (x, y), attn_weight
(N, Lq, M, L, P); forward -> (N, Lq, M * D); backward -> (grad_value, grad_sampling_loc, grad_attn_weight).  Unlike
upstream there is no ``im2col_step`` divisibility requirement (the argument is accepted and ignored).
"""
import ctypes
import os

import torch

_P, _I = ctypes.c_void_p, ctypes.c_int32


def _load():
    here = os.path.dirname(os.path.abspath(__file__))
    cands = [os.environ.get("DPFT_HIP_LIB"), os.path.join(here, "..", "dpft_amd", "libdpft_hip.so"), "libdpft_hip.so"]
    err = None
    for c in cands:
        if not c:
            continue
        try:
            return ctypes.CDLL(c)
        except OSError as e:
            err = e
    raise ImportError(f"MultiScaleDeformableAttention: libdpft_hip.so not found ({err}); build it with "
                      "`make -C dpft_amd/csrc ARCH=gfx950` or point DPFT_HIP_LIB at it")


_lib = _load()
_lib.dpft_msda_fwd_f32.argtypes = [_P] * 6 + [_I] * 7 + [_P]
_lib.dpft_msda_fwd_f32.restype = _I
_lib.dpft_msda_bwd_f32.argtypes = [_P] * 9 + [_I] * 7 + [_P]
_lib.dpft_msda_bwd_f32.restype = _I
_lib.dpft_last_error.restype = ctypes.c_char_p


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check(rc, what):
    if rc:
        raise RuntimeError(f"{what}: {_lib.dpft_last_error().decode('utf-8', 'replace')} (rc={rc})")


def _operands(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    for name, t in (("value", value), ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)):
        if not (t.is_cuda and t.dtype == torch.float32):
            raise RuntimeError(f"MultiScaleDeformableAttention: {name} must be a float32 GPU tensor (got {t.dtype} on {t.device})")
    if value.dim() != 4 or sampling_loc.dim() != 6 or attn_weight.dim() != 5:
        raise RuntimeError("MultiScaleDeformableAttention: value (N,S,M,D), sampling_loc (N,Lq,M,L,P,2), attn_weight (N,Lq,M,L,P)")
    return (value.contiguous(), spatial_shapes.to(device=value.device, dtype=torch.int64).contiguous(),
            level_start_index.to(device=value.device, dtype=torch.int64).contiguous(), sampling_loc.contiguous(),
            attn_weight.contiguous())


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    value, spatial_shapes, level_start_index, sampling_loc, attn_weight = _operands(
        value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    out = value.new_empty(N, Lq, M * D)
    _check(_lib.dpft_msda_fwd_f32(_p(value), _p(spatial_shapes), _p(level_start_index), _p(sampling_loc), _p(attn_weight),
                                  _p(out), N, S, M, D, Lq, L, P, _stream()), "ms_deform_attn_forward")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    value, spatial_shapes, level_start_index, sampling_loc, attn_weight = _operands(
        value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    grad_output = grad_output.contiguous()
    gv, gl, ga = torch.zeros_like(value), torch.empty_like(sampling_loc), torch.empty_like(attn_weight)
    _check(_lib.dpft_msda_bwd_f32(_p(value), _p(spatial_shapes), _p(level_start_index), _p(sampling_loc), _p(attn_weight),
                                  _p(grad_output), _p(gv), _p(gl), _p(ga), N, S, M, D, Lq, L, P, _stream()),
           "ms_deform_attn_backward")
    return gv, gl, ga
