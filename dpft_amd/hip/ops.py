"""Tensor-level wrappers over the C-ABI: allocate outputs with torch, pass raw pointers.

These are *not* autograd functions; the hand-scheduled forward/backward pipelines in
``dpft_amd.models`` compose them and expose autograd at module granularity.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from dpft_amd.hip.lib import ConvDesc, lib, make_desc, make_pyramid, ptr, stream

_ws_cache = {}

def conv_set_compute(mode: str) -> None:
    """"fp32" (the reference's arithmetic, default) or "bf16" (bf16 operands / fp32 accumulation in the forward and
    data-gradient GEMMs of the C % 64 == 0 convs; BASELINE.json configs[4])."""
    lib.call("dpft_conv_set_compute", {"fp32": 0, "bf16": 1, "bf16x3": 2}[mode])
    _conv_cache.clear()      # tile shapes (statistics tiles, workspace) depend on the mode


def conv_get_compute() -> str:
    return ("fp32", "bf16", "bf16x3")[int(lib.dpft_conv_get_compute())]


def conv_set_split(on: bool) -> None:
    """fp32 mode: big multi-tap GEMMs as 3 x bf16 split products on the bf16 matrix cores (default on; conv_x3.hip).
    Changes tile shapes: cached problems are dropped."""
    lib.call("dpft_conv_set_split", int(bool(on)))
    _conv_cache.clear()


def conv_get_split() -> bool:
    return bool(lib.dpft_conv_get_split())


def profile_start():
    lib.call("dpft_profile_start")


CONV_FAMILIES = ("f32", "x3", "bf16", "vector")      # dpft_profile_get_family: the pipe the library launched the conv on


def profile_collect():
    """-> list of (kind, flops, seconds, shape7, family) for every conv launch since profile_start() (syncs).
    family: 'f32' fp32 MFMA | 'x3' three-term bf16 split on the bf16 MFMA | 'bf16' bf16 operands | 'vector' no matrix core
    -- written by the library's dispatch code at the launch, not derived from the shape."""
    n = int(lib.dpft_profile_stop())
    torch.cuda.synchronize()
    out = []
    kind, flops, ms, shape, fam = C.c_int32(), C.c_double(), C.c_float(), (C.c_int32 * 7)(), C.c_int32()
    names = ("fwd", "dgrad", "wgrad")
    for i in range(n):
        lib.call("dpft_profile_get", i, C.byref(kind), C.byref(flops), C.byref(ms), C.byref(shape))
        lib.call("dpft_profile_get_family", i, C.byref(fam))
        out.append((names[kind.value], flops.value, ms.value * 1e-3, tuple(shape), CONV_FAMILIES[fam.value]))
    return out


def workspace(nbytes: int, device) -> Optional[torch.Tensor]:
    """One grow-only scratch buffer per device (split-K partials).  Stream-ordered reuse is safe
    because every consumer of the scratch is enqueued on the same stream before the next producer."""
    if nbytes <= 0:
        return None
    # one scratch per (device, stream): the view encoders run concurrently on separate streams
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        # zeros: the buffer starts with the split-K ticket header (include/dpft_hip.h: must be zero at first use; the
        # convolutions leave it zero)
        buf = torch.zeros(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


MEMOPS_MAX = 16
SUM_SRCS_MAX = 16


def memops(pairs) -> None:
    """[(dst, src | None), ...] contiguous same-dtype device tensors: all copies (src None = zero fill) in ceil(n / 16)
    launches on the current stream (dpft_memops) -- the glue's replacement for per-tensor Tensor.copy_ / zero_ calls."""
    from dpft_amd.hip.lib import MemOp
    todo = []
    for dst, src in pairs:
        if dst.numel() == 0:
            continue
        nbytes = dst.numel() * dst.element_size()
        if not dst.is_contiguous() or nbytes % 4 or (src is not None and (not src.is_contiguous() or src.dtype != dst.dtype or
                                                                         src.numel() != dst.numel() or src.device != dst.device)):
            dst.copy_(src) if src is not None else dst.zero_()      # layouts the kernel does not take (never on the step's path)
            continue
        todo.append((dst.data_ptr(), 0 if src is None else src.data_ptr(), nbytes))
    for i in range(0, len(todo), MEMOPS_MAX):
        chunk = todo[i:i + MEMOPS_MAX]
        arr = (MemOp * len(chunk))(*[MemOp(d, s or None, n) for d, s, n in chunk])
        lib.call("dpft_memops", len(chunk), C.cast(arr, C.c_void_p), stream())


def sum_leading(srcs, inner_shape, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """out (+)= sum over every source's leading axes: each source is a contiguous fp32 tensor whose trailing dimensions are
    ``inner_shape``; one launch (dpft_sum_leading_f32), fixed order.  Up to 16 sources."""
    from dpft_amd.hip.lib import SumSrc
    inner = 1
    for d in inner_shape:
        inner *= int(d)
    srcs = [t if t.is_contiguous() else t.contiguous() for t in srcs]
    if out is None:
        out = torch.empty(tuple(inner_shape), dtype=torch.float32, device=srcs[0].device)
        accumulate = False
    arr = (SumSrc * len(srcs))(*[SumSrc(t.data_ptr(), t.numel() // inner) for t in srcs])
    lib.call("dpft_sum_leading_f32", len(srcs), C.cast(arr, C.c_void_p), inner, ptr(out), int(accumulate), stream())
    return out


# Gradient sinks: a producer of an activation that wants its gradient at a FIXED address (the backbone's launch plan replays
# captured backward stages that read their output gradients from static buffers) registers that buffer here; the consumer's
# backward (the FPN's lateral data gradients) then writes the gradient straight into it instead of a fresh tensor that the
# producer would have to copy (12 device copies, 225 MB per step on kradar.json).  Keyed by the activation's address.
_grad_sinks = {}


def register_grad_sink(activation: torch.Tensor, sink: torch.Tensor) -> None:
    _grad_sinks[activation.data_ptr()] = sink


def drop_grad_sink(activation: torch.Tensor) -> None:
    _grad_sinks.pop(activation.data_ptr(), None)


def grad_sink(activation: torch.Tensor) -> Optional[torch.Tensor]:
    s = _grad_sinks.get(activation.data_ptr())
    if s is None or s.shape != activation.shape or s.device != activation.device or s.dtype != activation.dtype:
        return None
    return s


class Conv:
    """Geometry + cached descriptor/workspace size of one convolution problem."""
    __slots__ = ("desc", "ws_bytes", "B", "H", "W", "C", "K", "kh", "kw", "stride", "pad", "OH", "OW")

    def __init__(self, B, H, W, Cin, K, kh, kw, stride, pad):
        self.desc = make_desc(B, H, W, Cin, K, kh, kw, stride, pad)
        self.B, self.H, self.W, self.C, self.K = B, H, W, Cin, K
        self.kh, self.kw, self.stride, self.pad = kh, kw, stride, pad
        self.OH, self.OW = self.desc.OH, self.desc.OW
        self.ws_bytes = int(lib.dpft_conv2d_workspace_bytes(C.byref(self.desc)))
        if self.ws_bytes < 0:
            raise RuntimeError("bad conv descriptor: " + lib.dpft_last_error().decode())

    # row tiles of the forward kernel's statistics epilogue: asked at every use -- the answer follows the compute mode
    # (conv_set_compute / conv_set_split), and a problem object may outlive a mode change
    @property
    def tiles(self):
        return int(lib.dpft_conv2d_stats_tiles(C.byref(self.desc), None))

    @property
    def tile_rows(self):
        tr = C.c_int32(0)
        lib.dpft_conv2d_stats_tiles(C.byref(self.desc), C.byref(tr))
        return tr.value

    @property
    def M(self):
        return self.B * self.OH * self.OW


_conv_cache = {}


def conv_problem(B, H, W, Cin, K, kh, kw, stride, pad) -> Conv:
    key = (B, H, W, Cin, K, kh, kw, stride, pad)
    c = _conv_cache.get(key)
    if c is None:
        c = _conv_cache[key] = Conv(*key)
    return c


def split_planes(t: torch.Tensor) -> torch.Tensor:
    """fp32 tensor -> its three bf16 planes, shape (3, *t.shape): t == planes.float().sum(0) exactly
    (dpft_split_planes_f32; the operand format of the split convolution kernels, conv_x3.hip)."""
    t = t.contiguous()
    out = torch.empty((3,) + tuple(t.shape), dtype=torch.bfloat16, device=t.device)
    lib.call("dpft_split_planes_f32", ptr(t), ptr(out), t.numel(), stream())
    return out


class _with_planes:
    """Puts (a_planes, w_planes) into a cached problem's descriptor for the duration of one call."""

    def __init__(self, cv, planes):
        self.cv, self.planes = cv, planes

    def __enter__(self):
        if self.planes is not None:
            self.cv.desc.a_planes, self.cv.desc.w_planes = ptr(self.planes[0]), ptr(self.planes[1])

    def __exit__(self, *exc):
        self.cv.desc.a_planes = self.cv.desc.w_planes = None


def conv_fwd(cv: Conv, x, w, bias=None, pro=None, want_stats=False, out=None, planes=None):
    """x (B,H,W,C) contiguous; w physical [K][kh][kw][C]. pro = (bn_block[4][C], relu) or None.
    Returns y (B,OH,OW,K) and the per-tile stats tensor (or None).  ``out``: write y into this buffer.
    ``planes`` = (split_planes(x), split_planes(w)): the GEMM reads these instead (bf16 matrix cores, fp32-grade result)."""
    y = out if out is not None else torch.empty((cv.B, cv.OH, cv.OW, cv.K), dtype=torch.float32, device=x.device)
    if out is not None and (tuple(out.shape) != (cv.B, cv.OH, cv.OW, cv.K) or not out.is_contiguous() or out.dtype != torch.float32):
        raise ValueError("conv_fwd: the output buffer does not match the problem")
    stats = torch.empty((cv.tiles, 2, cv.K), dtype=torch.float32, device=x.device) if want_stats else None
    ws = workspace(cv.ws_bytes, x.device)
    pb, prelu = (pro[0], int(pro[1])) if pro is not None else (None, 0)
    with _with_planes(cv, planes):
        lib.call("dpft_conv2d_nhwc_fwd_f32", C.byref(cv.desc), ptr(x), ptr(w), ptr(bias), ptr(pb), prelu,
                 ptr(y), ptr(stats), ptr(ws), stream())
    return y, stats


def fpn_lateral(cv: Conv, x, w, bias, top=None, out=None):
    """lat = conv1x1(x) + bias [+ nearest_upsample(top)] -- the neck's lateral with the top-down add in its epilogue where a
    thin-channel kernel takes the shape (dpft_fpn_lateral_f32; otherwise conv + fpn_topdown_add, same arithmetic)."""
    y = out if out is not None else torch.empty((cv.B, cv.OH, cv.OW, cv.K), dtype=torch.float32, device=x.device)
    ws = workspace(cv.ws_bytes, x.device)
    TH, TW = (top.shape[1], top.shape[2]) if top is not None else (0, 0)
    lib.call("dpft_fpn_lateral_f32", C.byref(cv.desc), ptr(x), ptr(w), ptr(bias), ptr(top), TH, TW, ptr(y), ptr(ws), stream())
    return y


def fpn_output(cv: Conv, lat, w, bias, pos=None, out=None):
    """out = conv3x3(lat) + bias [+ pos_x[W]; + pos_y[H]] -- the neck's output conv with the positional embedding in its epilogue
    (dpft_fpn_output_f32).  ``pos`` = (pos_x (W,K), pos_y (H,K)) or None."""
    y = out if out is not None else torch.empty((cv.B, cv.OH, cv.OW, cv.K), dtype=torch.float32, device=lat.device)
    if out is not None and (tuple(out.shape) != (cv.B, cv.OH, cv.OW, cv.K) or not out.is_contiguous() or out.dtype != torch.float32):
        raise ValueError("fpn_output: the output buffer does not match the problem")
    ws = workspace(cv.ws_bytes, lat.device)
    px, py = pos if pos is not None else (None, None)
    lib.call("dpft_fpn_output_f32", C.byref(cv.desc), ptr(lat), ptr(w), ptr(bias), ptr(px), ptr(py), ptr(y), ptr(ws), stream())
    return y


def conv_fwd_bnact(cv: Conv, x, w, out_bn, relu=True, residual=None):
    """Inference conv + BatchNorm (+ residual) (+ ReLU): [relu](bn(conv(x, w)) [+ residual]); out_bn = BN block (4,K) of
    the output channels (bn_eval_params)."""
    y = torch.empty((cv.B, cv.OH, cv.OW, cv.K), dtype=torch.float32, device=x.device)
    ws = workspace(cv.ws_bytes, x.device)
    lib.call("dpft_conv2d_nhwc_fwd_bnact_f32", C.byref(cv.desc), ptr(x), ptr(w), ptr(out_bn), int(relu), ptr(residual),
             ptr(y), ptr(ws), stream())
    return y


def conv_dgrad(cv: Conv, dy, w_t, out=None, accumulate=False, planes=None):
    """dx (B,H,W,C); w_t physical [C][kh][kw][K].  ``planes`` = (split_planes(dy), split_planes(w_t))."""
    if out is None:
        out = torch.empty((cv.B, cv.H, cv.W, cv.C), dtype=torch.float32, device=dy.device)
        accumulate = False
    ws = workspace(cv.ws_bytes, dy.device)
    with _with_planes(cv, planes):
        lib.call("dpft_conv2d_nhwc_dgrad_f32", C.byref(cv.desc), ptr(dy), ptr(w_t), ptr(out), int(accumulate), ptr(ws),
                 stream())
    return out


def conv_dgrad_bn_reduce(cv: Conv, dy, w_t, bn_y, bn_block, sums, bn_mask8=None, residual=None, out=None, accumulate=False,
                         planes=None):
    """The launch plan's fused data gradient (dpft_conv2d_nhwc_dgrad_bn_reduce_f32): dx, and -- when the launch could carry
    it (returned flag) -- the BatchNorm-backward sums of the layer whose dout dx is added into ``sums`` [2][C].
    ``bn_mask8`` None = the ReLU sits directly behind that BatchNorm (mask = bn(bn_y) > 0).
    ``residual`` = (res_src, block_out, res_mask8 or None): dx += res_src under the block output's ReLU mask."""
    if out is None:
        out = torch.empty((cv.B, cv.H, cv.W, cv.C), dtype=torch.float32, device=dy.device)
        accumulate = False
    ws = workspace(cv.ws_bytes, dy.device)
    applied = C.c_int32(0)
    rs, ro, rm = residual if residual is not None else (None, None, None)
    with _with_planes(cv, planes):
        lib.call("dpft_conv2d_nhwc_dgrad_bn_reduce_f32", C.byref(cv.desc), ptr(dy), ptr(w_t), ptr(out), int(accumulate),
                 ptr(rs), ptr(ro), ptr(rm), ptr(bn_y), ptr(bn_block), ptr(bn_mask8), int(bn_mask8 is None), ptr(sums),
                 C.addressof(applied), ptr(ws), stream())
    return out, bool(applied.value)


def conv_wgrad(cv: Conv, x, dy, pro=None, out=None):
    """dw physical [K][kh][kw][C] (returned as a (K,kh,kw,C) tensor).  ``out``: write into this buffer (same physical
    layout, e.g. a DP bucket view) instead of a fresh tensor."""
    dw = out if out is not None else torch.empty((cv.K, cv.kh, cv.kw, cv.C), dtype=torch.float32, device=x.device)
    ws = workspace(cv.ws_bytes, x.device)
    pb, prelu = (pro[0], int(pro[1])) if pro is not None else (None, 0)
    lib.call("dpft_conv2d_nhwc_wgrad_f32", C.byref(cv.desc), ptr(x), ptr(dy), ptr(pb), prelu, ptr(dw),
             ptr(ws), stream())
    return dw


def conv_wgrad_bias(cv: Conv, x, dy, out=None, bias_out=None):
    """(dw, db) of a conv with bias in one call (dpft_conv2d_nhwc_wgrad_bias_f32); ``out`` / ``bias_out`` as in conv_wgrad /
    bias_grad."""
    dw = out if out is not None else torch.empty((cv.K, cv.kh, cv.kw, cv.C), dtype=torch.float32, device=x.device)
    db = bias_out if bias_out is not None else torch.empty(cv.K, dtype=torch.float32, device=x.device)
    ws = workspace(cv.ws_bytes, x.device)
    lib.call("dpft_conv2d_nhwc_wgrad_bias_f32", C.byref(cv.desc), ptr(x), ptr(dy), ptr(dw), ptr(db), ptr(ws), stream())
    return dw, db


def weight_transpose(w_khwc: torch.Tensor) -> torch.Tensor:
    """[K][kh][kw][C] -> [C][kh][kw][K]"""
    K, kh, kw, Cin = w_khwc.shape
    wt = torch.empty((Cin, kh, kw, K), dtype=torch.float32, device=w_khwc.device)
    lib.call("dpft_weight_transpose_f32", ptr(w_khwc), ptr(wt), K, kh * kw, Cin, stream())
    return wt


def weight_transpose_many(ws) -> list:
    """``weight_transpose`` of up to 80 [K][kh][kw][C] weights in one launch (one flat buffer behind the results)."""
    ws = list(ws)
    n = len(ws)
    sizes = [w.numel() for w in ws]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=ws[0].device)
    outs, off = [], 0
    for w, sz in zip(ws, sizes):
        K, kh, kw, Cin = w.shape
        outs.append(flat[off:off + sz].view(Cin, kh, kw, K))
        off += sz
    src = (C.c_void_p * n)(*[w.data_ptr() for w in ws])
    dst = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    Ks = (C.c_int32 * n)(*[w.shape[0] for w in ws])
    taps = (C.c_int32 * n)(*[w.shape[1] * w.shape[2] for w in ws])
    Cs = (C.c_int32 * n)(*[w.shape[3] for w in ws])
    lib.call("dpft_weight_transpose_batch_f32", n, C.cast(src, C.c_void_p), C.cast(dst, C.c_void_p), C.cast(Ks, C.c_void_p),
             C.cast(taps, C.c_void_p), C.cast(Cs, C.c_void_p), stream())
    return outs


def bias_grad(dy: torch.Tensor, out=None) -> torch.Tensor:
    K = dy.shape[-1]
    db = out if out is not None else torch.empty((K,), dtype=torch.float32, device=dy.device)
    lib.call("dpft_bias_grad_f32", ptr(dy), ptr(db), dy.numel() // K, K, stream())
    return db


def bn_stats(y: torch.Tensor, tile_rows: int = 128) -> torch.Tensor:
    K = y.shape[-1]
    M = y.numel() // K
    tiles = (M + tile_rows - 1) // tile_rows
    stats = torch.empty((tiles, 2, K), dtype=torch.float32, device=y.device)
    lib.call("dpft_bn_stats_f32", ptr(y), ptr(stats), M, K, tile_rows, stream())
    return stats


def bn_finalize(stats, tile_rows, M, gamma, beta, eps, momentum, running_mean=None, running_var=None):
    """-> BN block (4,K) = (mean, gamma*invstd, beta, invstd); updates the running buffers in place."""
    K = gamma.numel()
    bnp = torch.empty((4, K), dtype=torch.float32, device=gamma.device)
    lib.call("dpft_bn_finalize_f32", ptr(stats), stats.shape[0], tile_rows, M, K, ptr(gamma), ptr(beta),
             float(eps), float(momentum), ptr(running_mean), ptr(running_var), ptr(bnp), stream())
    return bnp


def bn_eval_params(gamma, beta, running_mean, running_var, eps):
    K = gamma.numel()
    bnp = torch.empty((4, K), dtype=torch.float32, device=gamma.device)
    lib.call("dpft_bn_eval_params_f32", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps),
             K, ptr(bnp), stream())
    return bnp


def bn_act(y, bnp, res=None, res_bnp=None, relu=True):
    out = torch.empty_like(y)
    K = y.shape[-1]
    lib.call("dpft_bn_act_f32", ptr(y), ptr(bnp), ptr(res), ptr(res_bnp), int(relu), ptr(out), y.numel() // K, K,
             stream())
    return out


def bn_relu_maxpool(y, bnp):
    B, H, W, K = y.shape
    PH, PW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    out = torch.empty((B, PH, PW, K), dtype=torch.float32, device=y.device)
    lib.call("dpft_bn_relu_maxpool_f32", ptr(y), ptr(bnp), ptr(out), B, H, W, K, PH, PW, stream())
    return out


def bn_relu_maxpool_bwd(y, bnp, dout):
    B, H, W, K = y.shape
    PH, PW = dout.shape[1], dout.shape[2]
    dz = torch.empty_like(y)
    lib.call("dpft_bn_relu_maxpool_bwd_f32", ptr(y), ptr(bnp), ptr(dout), ptr(dz), B, H, W, K, PH, PW, stream())
    return dz


def bn_bwd(y, dout, bnp, gamma, out=None, mask_bnp=None):
    """Full BatchNorm backward (two passes).  mask_bnp recomputes a fused ReLU mask from a BN block.
    Returns dy, dgamma, dbeta."""
    K = y.shape[-1]
    M = y.numel() // K
    sums = torch.empty((2, K), dtype=torch.float32, device=y.device)
    lib.call("dpft_bn_bwd_reduce_f32", ptr(y), ptr(dout), ptr(out), ptr(mask_bnp), ptr(bnp), ptr(sums), M, K,
             stream())
    dy = torch.empty_like(y)
    dgb = torch.empty((2, K), dtype=torch.float32, device=y.device)
    lib.call("dpft_bn_bwd_apply_f32", ptr(y), ptr(dout), ptr(out), ptr(mask_bnp), ptr(bnp), ptr(gamma), ptr(sums),
             ptr(dy), ptr(dgb[0]), ptr(dgb[1]), M, K, stream())
    return dy, dgb[0], dgb[1]


def relu_bwd(dout, out):
    dz = torch.empty_like(dout)
    lib.call("dpft_relu_bwd_f32", ptr(dout), ptr(out), ptr(dz), dout.numel(), stream())
    return dz


def add_(a, b):
    lib.call("dpft_add_inplace_f32", ptr(a), ptr(b), a.numel(), stream())
    return a


def fpn_topdown_add_(lat, top):
    B, H, W, K = lat.shape
    lib.call("dpft_fpn_topdown_add_f32", ptr(lat), ptr(top), B, H, W, top.shape[1], top.shape[2], K, stream())
    return lat


def fpn_topdown_add_bwd_(dlat, dtop):
    B, H, W, K = dlat.shape
    lib.call("dpft_fpn_topdown_add_bwd_f32", ptr(dlat), ptr(dtop), B, H, W, dtop.shape[1], dtop.shape[2], K,
             stream())
    return dtop


def add_pos_(x, pos_x, pos_y):
    B, H, W, K = x.shape
    lib.call("dpft_add_pos_f32", ptr(x), ptr(pos_x), ptr(pos_y), B, H, W, K, stream())
    return x


def msda_fwd(value, shapes, lsi, loc, attn):
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = torch.empty((N, Lq, M * D), dtype=torch.float32, device=value.device)
    lib.call("dpft_msda_fwd_f32", ptr(value), ptr(shapes), ptr(lsi), ptr(loc), ptr(attn), ptr(out), N, S, M, D, Lq,
             L, P, stream())
    return out


def msda_bwd(value, shapes, lsi, loc, attn, grad_out):
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    gv = torch.zeros_like(value)
    gl = torch.empty_like(loc)
    ga = torch.empty_like(attn)
    lib.call("dpft_msda_bwd_f32", ptr(value), ptr(shapes), ptr(lsi), ptr(loc), ptr(attn), ptr(grad_out), ptr(gv),
             ptr(gl), ptr(ga), N, S, M, D, Lq, L, P, stream())
    return gv, gl, ga


def xattn_fwd(levels: Sequence[torch.Tensor], ref, off, attn, Wv, bv, n_heads: int, n_points: int):
    B, Q, _ = ref.shape
    Cm = levels[0].shape[-1]
    D = Cm // n_heads
    pyr = make_pyramid(levels)
    out = torch.empty((B, Q, Cm), dtype=torch.float32, device=ref.device)
    samp = torch.empty((B, Q, n_heads, Cm), dtype=torch.float32, device=ref.device)
    mass = torch.empty((B, Q, n_heads), dtype=torch.float32, device=ref.device)
    lib.call("dpft_xattn_fwd_f32", C.byref(pyr), ptr(ref), ptr(off), ptr(attn), ptr(Wv), ptr(bv), ptr(out),
             ptr(samp), ptr(mass), B, Q, n_heads, D, n_points, stream())
    return out, samp, mass


def xattn_bwd(levels, level_grads, ref, off, attn, Wv, bv, grad_out, n_heads: int, n_points: int):
    B, Q, _ = ref.shape
    D = levels[0].shape[-1] // n_heads
    pyr = make_pyramid(levels, level_grads)
    goff = torch.empty_like(off)
    gattn = torch.empty_like(attn)
    gref = torch.empty_like(ref)
    lib.call("dpft_xattn_bwd_f32", C.byref(pyr), ptr(ref), ptr(off), ptr(attn), ptr(Wv), ptr(bv), ptr(grad_out),
             ptr(goff), ptr(gattn), ptr(gref), B, Q, n_heads, D, n_points, stream())
    return goff, gattn, gref


def giou3d_yaw(pred7: torch.Tensor, gt7: torch.Tensor) -> torch.Tensor:
    """pred (B,N,7), gt (B,M,7) rows (x,y,z,l,w,h,yaw) -> (B,N,M)."""
    B, N, _ = pred7.shape
    Mg = gt7.shape[1]
    out = torch.empty((B, N, Mg), dtype=torch.float32, device=pred7.device)
    lib.call("dpft_giou3d_yaw_f32", ptr(pred7), ptr(gt7), ptr(out), B, N, Mg, stream())
    return out
