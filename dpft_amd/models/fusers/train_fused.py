"""Fused HIP training blocks of the fusion decoder (dpft_amd/csrc/decoder_train.hip) as autograd Functions.

``SelfAttnBlocksFn``: the self-attention block of every view of one MPFusion layer
(src/dprt/models/fusers/mpfusion.py:122-148) in one forward launch and two backward launches.  Dropout masks are
regenerated in the backward from a device-side seed, so the op is hipGraph-capturable: the seed lives in a
persistent int64 tensor that the captured ``advance_seed`` kernels bump on every replay.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List

import torch

from dpft_amd.hip import ops
from dpft_amd.hip.lib import DecoderView, HeadTrain, Pyramid, SaParams, lib, make_pyramid, stream

_SA_SIZES = (768, 48, 256, 16, 16, 16)        # in_proj_weight, in_proj_bias, out_proj.weight, .bias, norm1.weight, .bias
_seed_state: Dict[torch.device, torch.Tensor] = {}


def advance_seed(device: torch.device) -> torch.Tensor:
    """Snapshot of the dropout seed for this decoder forward; the persistent state moves on (captured in graphs)."""
    st = _seed_state.get(device)
    if st is None:
        st = _seed_state[device] = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(device)
    snap = torch.empty_like(st)
    lib.call("dpft_seed_advance", st.data_ptr(), snap.data_ptr(), 0x9E3779B97F4A7C15 & (2 ** 62 - 1), stream())      # snap = state; state += c
    return snap


class PosState:
    """The (Q,16) query-position table of one decoder forward: read by both blocks of every layer (8 uses on kradar.json).
    The blocks take the detached table plus the hub's token and leave their un-reduced gradient buffers here; the hub's
    backward, which autograd runs after the last of them, sums everything in ONE launch (ops.sum_leading, fixed order) --
    instead of a reduction per block and the engine's chain of adds."""

    def __init__(self, pos: torch.Tensor):
        self.pos = pos.detach().contiguous()
        self.parts: List[torch.Tensor] = []


class _PosHubFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state: PosState, pos):
        ctx.state = state
        ctx.set_materialize_grads(False)      # the token carries ordering only
        return pos.new_empty(())

    @staticmethod
    def backward(ctx, gtoken):
        parts, ctx.state.parts = ctx.state.parts, []
        if not parts:
            return None, None
        g = None
        for i in range(0, len(parts), ops.SUM_SRCS_MAX):
            g = ops.sum_leading(parts[i:i + ops.SUM_SRCS_MAX], ctx.state.pos.shape, out=g, accumulate=g is not None)
        return None, g


def make_pos_hub(pos: torch.Tensor):
    """-> (state, token) for ``self_attn_blocks`` / ``xattn_ffn_blocks`` in place of the table itself."""
    state = PosState(pos)
    return state, _PosHubFn.apply(state, pos)


def _pos_args(pos):
    """table | hub -> (table the kernels read, hub state | None, differentiable tensor threaded through autograd)"""
    if isinstance(pos, tuple):
        return pos[0].pos, pos[0], pos[1]
    return pos, None, pos


def sa_supported(ml) -> bool:
    a = ml.self_attn
    return (ml.d_model == 16 and ml.n_heads == 8 and ml.norm and a.batch_first and a._qkv_same_embed_dim
            and a.in_proj_bias is not None and a.bias_k is None and not a.add_zero_attn)


def sa_params(ml) -> List[torch.Tensor]:
    a = ml.self_attn
    return [a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, ml.norm1.weight, ml.norm1.bias]


def _structs(tensors: List[torch.Tensor], V: int):
    arr = (SaParams * V)()
    for v in range(V):
        arr[v] = SaParams(*[t.data_ptr() for t in tensors[6 * v:6 * v + 6]])
    return arr


class SelfAttnBlocksFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos_dep, seed, salt: int, p_drop: float, pos, hub, batch, *params):
        # x (B,Q,16), or the (Q,16) table of a first layer, broadcast over `batch` elements (row stride 0);
        # pos_dep: the table itself or a hub token (what autograd differentiates), pos: the table the kernels read
        V = len(params) // 6
        if x.stride()[-2:] != (16, 1):
            x = x.contiguous()
        pos = pos.contiguous()
        params = [p if p.is_contiguous() else p.contiguous() for p in params]
        B, Q = (int(batch), x.shape[0]) if x.dim() == 2 else x.shape[:2]
        xs0 = 0 if (x.dim() == 2 or B == 1) else x.stride(0)
        dev = x.device
        y1 = torch.empty((V, B, Q, 16), dtype=torch.float32, device=dev)
        attn, zhat = torch.empty_like(y1), torch.empty_like(y1)
        lse = torch.empty((V, B, Q, 8), dtype=torch.float32, device=dev)
        rstd = torch.empty((V, B, Q), dtype=torch.float32, device=dev)
        arr = _structs(params, V)
        lib.call("dpft_selfattn_train_fwd_f32", C.cast(arr, C.c_void_p), V, x.data_ptr(), xs0,
                 pos.data_ptr(), float(p_drop), seed.data_ptr(), int(salt), y1.data_ptr(), lse.data_ptr(),
                 attn.data_ptr(), zhat.data_ptr(), rstd.data_ptr(), B, Q, stream())
        ctx.save_for_backward(x, pos, seed, lse, attn, zhat, rstd, *params)
        ctx.meta = (V, int(salt), float(p_drop), B, Q, xs0)
        ctx.hub = hub
        return y1

    @staticmethod
    def backward(ctx, dy1):
        x, pos, seed, lse, attn, zhat, rstd, *params = ctx.saved_tensors
        V, salt, p_drop, B, Q, xs0 = ctx.meta
        dev = x.device
        dy1 = dy1.contiguous()
        per_view = sum(_SA_SIZES)
        flat = torch.empty(V * per_view, dtype=torch.float32, device=dev)
        ops.memops([(flat, None)])
        grads, garr, off = [], (SaParams * V)(), 0
        for v in range(V):
            ptrs = []
            for n, p in zip(_SA_SIZES, params[6 * v:6 * v + 6]):
                grads.append(flat[off:off + n].view(p.shape))
                ptrs.append(flat.data_ptr() + off * 4)
                off += n
            garr[v] = SaParams(*ptrs)
        dx = torch.empty((V, B, Q, 16), dtype=torch.float32, device=dev)
        dxp = torch.empty_like(dx)
        scratch = torch.empty(int(lib.dpft_selfattn_train_scratch_floats(B, Q, V)), dtype=torch.float32, device=dev)
        arr = _structs(params, V)
        lib.call("dpft_selfattn_train_bwd_f32", C.cast(arr, C.c_void_p), V, x.data_ptr(), xs0,
                 pos.data_ptr(), p_drop, seed.data_ptr(), salt, dy1.data_ptr(), lse.data_ptr(), attn.data_ptr(),
                 zhat.data_ptr(), rstd.data_ptr(), C.cast(garr, C.c_void_p), dx.data_ptr(), dxp.data_ptr(),
                 scratch.data_ptr(), B, Q, stream())
        gx = ops.sum_leading([dx], x.shape)                       # over the views (and the batch for a broadcast table)
        if ctx.hub is not None:
            ctx.hub.parts.append(dxp)                             # summed with the other blocks' shares by the hub
            gpos = None
        else:
            gpos = ops.sum_leading([dxp], pos.shape)
        return (gx, gpos, None, None, None, None, None, None, *grads)


def self_attn_blocks(layers, x, pos, seed, salt: int, p_drop: float, batch=None):
    """y1 (V,B,Q,16) of the V MLFusion layers' self-attention blocks; x (B,Q,16) -- or (Q,16), broadcast over ``batch``
    elements --, pos (Q,16) or a hub (``make_pos_hub``)."""
    params = [t for ml in layers for t in sa_params(ml)]
    table, hub, dep = _pos_args(pos)
    return SelfAttnBlocksFn.apply(x, dep, seed, salt, p_drop, table, hub, batch, *params)


def _rows_outer(rows: torch.Tensor, specs, out_floats: int) -> torch.Tensor:
    """out[g][off + a*n_b + b] = sum_r rows[g][r][col_a+a] * rows[g][r][col_b+b] for every (col_a, n_a, col_b, n_b, off)
    of ``specs`` in ONE launch (dpft_rows_outer_f32; col_b < 0 = column sums).  rows (G,R,W) -> out (G,out_floats)."""
    from dpft_amd.hip.lib import OuterSpec
    G, R, W = rows.shape
    arr = (OuterSpec * len(specs))(*[OuterSpec(*sp) for sp in specs])
    out = torch.empty((G, out_floats), dtype=torch.float32, device=rows.device)
    args = (rows.data_ptr(), G, R, W, C.cast(arr, C.c_void_p), len(specs), out.data_ptr(), out_floats)
    if FORK_WGRADS and torch.cuda.is_current_stream_capturing():
        # Inside a graph capture the weight gradients leave the critical path: nothing in the rest of the decoder's backward
        # reads them (they go to the parameters), so the launch becomes a parallel branch of the graph that
        # ``join_forked`` closes before the capture's consumer of the gradients (GraphedFuser).  `rows` / `out` stay
        # referenced until then: the capture's allocator must not hand their memory to a later tensor of the main branch.
        cur = torch.cuda.current_stream(rows.device)
        side = _fork_streams.get(rows.device)
        if side is None:
            side = _fork_streams[rows.device] = torch.cuda.Stream(rows.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            lib.call("dpft_rows_outer_f32", *args, stream())
        _forked.append((side, rows, out))
    else:
        lib.call("dpft_rows_outer_f32", *args, stream())
    return out


# Measured (round 4, three alternating 100-step pairs on one box): the decoder's backward graph takes 1.52 ms with the eight
# forked launches against 1.39 ms in line -- every fork / join edge of a hipGraph becomes a cross-queue barrier that costs more
# than the 17 us launch it takes off the chain.  Off; kept as the measured negative.
FORK_WGRADS = os.environ.get("DPFT_DEC_FORK_WGRADS", "0") != "0"
_fork_streams: Dict[torch.device, "torch.cuda.Stream"] = {}
_forked: list = []


def join_forked() -> None:
    """The current stream waits for the forked weight-gradient launches of a capture in progress (see ``_rows_outer``)."""
    for side, *_ in _forked:
        torch.cuda.current_stream(side.device).wait_stream(side)
    _forked.clear()


# ---------------------------------------------------------------------------------------------------------
# deformable cross-attention + FFN block of all views (decoder_train_x.hip)
# ---------------------------------------------------------------------------------------------------------
# Small-map pyramid gradients through scatter records + LDS images (xf_scatter_small_kernel) instead of per-sample atomics.
# Built, parity-tested (tests/test_gpu_kernels.py::test_xattn_ffn_backward_small_map_scatter_equals_atomics) and measured in
# round 4: the decoder's backward takes 2140 us with it against 1769 us without (bench.py roofline_decoder_train) -- the
# atomics on maps that live in the L2 / Infinity Cache were never the expensive ones (the cold 64-byte read-modify-writes
# into the camera's two large levels are), and the record round trip costs more than they did.  Off by default.
XF_SCATTER = os.environ.get("DPFT_XF_SCATTER", "0") != "0"
_XR = dict(DLIN=0, DF=480, DPRE=496, DOUT=528, G3=544, B3=560, G2=576, B2=592, DBV=608, DVEC=624, QP=640, HD=656,
           Y2=688, VEC=704, SAMP=720, FLOATS=848)


def xf_supported(ml) -> bool:
    a = ml.ms_deform_attn
    return (sa_supported(ml) and ml.d_ffn == 32 and ml.activation == "Mish" and a.n_heads == 8
            and a.n_points <= 4 and a.n_levels * a.n_points <= 20 and a.n_levels <= 8)


def view_params(ml) -> List[torch.Tensor]:
    """The 22 tensors of dpft_decoder_view, in its field order."""
    a = ml.ms_deform_attn
    return [ml.self_attn.in_proj_weight, ml.self_attn.in_proj_bias, ml.self_attn.out_proj.weight,
            ml.self_attn.out_proj.bias, ml.norm1.weight, ml.norm1.bias,
            a.sampling_offsets.weight, a.sampling_offsets.bias, a.attention_weights.weight, a.attention_weights.bias,
            a.value_proj.weight, a.value_proj.bias, a.output_proj.weight, a.output_proj.bias,
            ml.norm2.weight, ml.norm2.bias, ml.ffn1.weight, ml.ffn1.bias, ml.ffn2.weight, ml.ffn2.bias,
            ml.norm3.weight, ml.norm3.bias]


def _pack_views(params: List[torch.Tensor], V: int, n_levels, n_points, dev):
    nv = int(lib.dpft_decoder_packed_view_floats())
    packed = torch.empty(V * nv, dtype=torch.float32, device=dev)
    views = (DecoderView * V)()
    for v in range(V):
        views[v] = DecoderView(*[t.data_ptr() for t in params[22 * v:22 * v + 22]])
    lib.call("dpft_decoder_pack_views_f32", C.cast(views, C.c_void_p), V, C.cast((C.c_int32 * V)(*n_levels), C.c_void_p),
             C.cast((C.c_int32 * V)(*n_points), C.c_void_p), packed.data_ptr(), stream())      # one launch for the V views
    return packed, views


class XattnFfnBlocksFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, states, seed, salt: int, p_drop: float, n_points, y1, pos_dep, refs, pos, hub, *rest):
        V = len(states)
        tokens, params = rest[:V], list(rest[V:])
        params = [p if p.is_contiguous() else p.contiguous() for p in params]
        y1, pos, refs = y1.contiguous(), pos.contiguous(), refs.contiguous()
        _, B, Q, _ = y1.shape
        dev = y1.device
        n_levels = [len(s.levels) for s in states]
        packed, views = _pack_views(params, V, n_levels, n_points, dev)
        pyrs = (Pyramid * V)()
        for v in range(V):
            pyrs[v] = make_pyramid(states[v].levels)
        npts = (C.c_int32 * V)(*n_points)
        y3 = torch.empty_like(y1)
        # the one expensive intermediate of a row (gathered features + masses, 544 B) is kept for the backward
        saved = torch.empty((V, B * Q, int(lib.dpft_xattn_ffn_train_saved_floats())), dtype=torch.float32, device=dev)
        lib.call("dpft_xattn_ffn_train_fwd_f32", C.cast(pyrs, C.c_void_p), C.cast(views, C.c_void_p), packed.data_ptr(),
                 V, C.cast(npts, C.c_void_p), y1.data_ptr(), pos.data_ptr(), refs.data_ptr(), float(p_drop),
                 seed.data_ptr(), int(salt), y3.data_ptr(), saved.data_ptr(), B, Q, stream())
        ctx.save_for_backward(y1, pos, refs, seed, packed, saved, *params)
        ctx.states, ctx.meta = states, (V, int(salt), float(p_drop), list(n_points), n_levels)
        ctx.hub = hub
        return y3

    @staticmethod
    def backward(ctx, dy3):
        y1, pos, refs, seed, packed, saved, *params = ctx.saved_tensors
        V, salt, p_drop, n_points, n_levels = ctx.meta
        states = ctx.states
        _, B, Q, _ = y1.shape
        dev = y1.device
        dy3 = dy3.contiguous()
        R, W = B * Q, _XR["FLOATS"]
        rows = torch.empty((V, R, W), dtype=torch.float32, device=dev)
        dy1, dqp = torch.empty_like(y1), torch.empty_like(y1)
        dref = torch.empty_like(refs)
        views = (DecoderView * V)()
        pyrs = (Pyramid * V)()
        for v in range(V):
            views[v] = DecoderView(*[t.data_ptr() for t in params[22 * v:22 * v + 22]])
            # small maps (incl. the tiny ones that used to need gradient replicas) are scattered through LDS images by
            # the backward's second launch: plain buffers, nothing to fold afterwards (DPFT_XF_SCATTER=0: atomics + replicas)
            pyrs[v] = make_pyramid(states[v].levels, states[v].grad_buffers() if XF_SCATTER
                                   else states[v].replicated_grad_buffers())
        npts = (C.c_int32 * V)(*n_points)
        scratch = torch.empty((V, R, int(lib.dpft_xattn_ffn_train_scratch_floats())), dtype=torch.float32,
                              device=dev) if XF_SCATTER else None
        lib.call("dpft_xattn_ffn_train_bwd_f32", C.cast(pyrs, C.c_void_p), C.cast(views, C.c_void_p), packed.data_ptr(),
                 V, C.cast(npts, C.c_void_p), y1.data_ptr(), pos.data_ptr(), refs.data_ptr(), p_drop, seed.data_ptr(),
                 salt, saved.data_ptr(), dy3.data_ptr(), dy1.data_ptr(), dqp.data_ptr(), dref.data_ptr(), rows.data_ptr(),
                 scratch.data_ptr() if scratch is not None else None, B, Q, stream())
        X = _XR
        # every weight gradient = a product of two column blocks of `rows`, summed over the rows: one launch
        specs, off = [], 0
        def add(ca, na, cb, nb):
            nonlocal off
            specs.append((ca, na, cb, nb, off))
            off += na * (1 if cb < 0 else nb)
            return off - na * (1 if cb < 0 else nb)
        o_col = add(0, X["QP"], -1, 1)                                            # (640) vector gradients
        o_oa = add(X["DLIN"], X["DF"] - X["DLIN"], X["QP"], 16)                   # (480,16)
        o_f2 = add(X["DF"], X["DPRE"] - X["DF"], X["HD"], 32)                     # (16,32)
        o_f1 = add(X["DPRE"], X["DOUT"] - X["DPRE"], X["Y2"], 16)                 # (32,16)
        o_op = add(X["DOUT"], X["G3"] - X["DOUT"], X["VEC"], 16)                  # (16,16)
        o_vw = [add(X["DVEC"] + 2 * m, 2, X["SAMP"] + 16 * m, 16) for m in range(8)][0]       # (8,2,16)
        res = _rows_outer(rows, specs, off)
        col = res[:, o_col:o_col + X["QP"]]
        g_oa = res[:, o_oa:o_oa + 480 * 16].view(V, 480, 16)
        g_f2 = res[:, o_f2:o_f2 + 512].view(V, 16, 32)
        g_f1 = res[:, o_f1:o_f1 + 512].view(V, 32, 16)
        g_op = res[:, o_op:o_op + 256].view(V, 16, 16)
        g_vw = res[:, o_vw:o_vw + 256].view(V, 16, 16)
        grads = []
        for v in range(V):
            n_off = 8 * n_levels[v] * n_points[v] * 2
            n_att = n_off // 2
            grads += [None] * 6
            grads += [g_oa[v, :n_off], col[v, :n_off], g_oa[v, n_off:n_off + n_att], col[v, n_off:n_off + n_att],
                      g_vw[v], col[v, X["DBV"]:X["DVEC"]], g_op[v], col[v, X["DOUT"]:X["G3"]],
                      col[v, X["G2"]:X["B2"]], col[v, X["B2"]:X["DBV"]],
                      g_f1[v], col[v, X["DPRE"]:X["DOUT"]], g_f2[v], col[v, X["DF"]:X["DPRE"]],
                      col[v, X["G3"]:X["B3"]], col[v, X["B3"]:X["G2"]]]
        gtok = [None] * V                       # tokens carry ordering only (_PyramidHub ignores their gradient)
        if ctx.hub is not None:
            ctx.hub.parts.append(dqp)
            gpos = None
        else:
            gpos = ops.sum_leading([dqp], pos.shape)
        return (None, None, None, None, None, dy1, gpos, dref, None, None, *gtok, *grads)


def xattn_ffn_blocks(layers, pyramids, y1, pos, refs, seed, salt: int, p_drop: float):
    """y3 (V,B,Q,16); pyramids = [(PyramidState, token)] per view, refs (V,B,Q,2)."""
    states = [p[0] for p in pyramids]
    tokens = [p[1] for p in pyramids]
    params = [t for ml in layers for t in view_params(ml)]
    n_points = [ml.ms_deform_attn.n_points for ml in layers]
    table, hub, dep = _pos_args(pos)
    return XattnFfnBlocksFn.apply(states, seed, salt, p_drop, n_points, y1, dep, refs, table, hub, *tokens, *params)


# ---------------------------------------------------------------------------------------------------------
# view reduction + detection head + next reference points (decoder_train_h.hip)
# ---------------------------------------------------------------------------------------------------------
_HR = dict(DX=0, Y3C=16, D1=80, D2=144, DO=208, H1=272, H2=336, X=400, FLOATS=416)
_BRANCHES = ("center", "size", "angle", "class")


def head_supported(layer, head) -> bool:
    from dpft_amd.models.heads.detection import LinearDetectionHead
    return (layer.reduction == "linear" and layer.d_model == 16 and isinstance(head, LinearDetectionHead)
            and head.num_reg_layers == 3 and head.num_cls_layers == 3 and not head.bias and head.in_channels == 16
            and head.num_classes <= 16 and (head.dropout == 0.0 or not head.training))


def head_params(layer, head) -> List[torch.Tensor]:
    """reduction weight + the 12 head weights (branch-major, layers .0 .3 .6)."""
    out = [layer.reduction_layer.weight]
    for name in _BRANCHES:
        seq = head.layers[name + "_head"]
        out += [seq[0].weight, seq[3].weight, seq[6].weight]
    return out


class _Proj:
    """Non-differentiable projection inputs of the reference points (per view: T, P, shape, flag)."""

    def __init__(self, projection, shape, flags):
        self.T = [t.contiguous().float() for t, _ in projection]
        self.P = [p.contiguous().float() for _, p in projection]
        # (H, W) rows as int64: read in place from the dataset's (B, >= 2) int64 rows when every view has the same row stride
        # (dpft_head_train.shape_stride), converted / compacted otherwise
        strides = {s.stride(0) for s in shape if s.dtype == torch.int64 and s.dim() == 2 and s.stride(1) == 1 and s.shape[1] >= 2}
        if len(strides) == 1 and all(s.dtype == torch.int64 and s.dim() == 2 and s.stride(1) == 1 for s in shape):
            self.shape, self.shape_stride = list(shape), strides.pop()
        else:
            self.shape, self.shape_stride = [s[:, :2].to(torch.int64).contiguous() for s in shape], 2
        self.flags = [int(f) for f in flags]

    def fill(self, h: HeadTrain):
        for v in range(len(self.T)):
            h.T[v], h.P[v], h.shape[v] = self.T[v].data_ptr(), self.P[v].data_ptr(), self.shape[v].data_ptr()
            h.p_rows[v], h.has_t[v] = self.P[v].shape[1], self.flags[v]
        h.shape_stride = self.shape_stride


def reference_points(proj: _Proj, center: torch.Tensor) -> torch.Tensor:
    """(V,B,Q,2) reference points of a center that carries no gradient (the querent output)."""
    V = len(proj.T)
    center = center[..., :3].detach().contiguous().float()
    B, Q, _ = center.shape
    refs = torch.empty((V, B, Q, 2), dtype=torch.float32, device=center.device)
    h = HeadTrain()
    proj.fill(h)
    h.prev_center, h.refs, h.num_classes = center.data_ptr(), refs.data_ptr(), 1
    lib.call("dpft_head_train_fwd_f32", C.byref(h), B, Q, V, stream())
    return refs


def _fill_head_weights(h: HeadTrain, weights, packed):
    h.packed, h.red_w = packed.data_ptr(), weights[0].data_ptr()
    for g in range(4):
        for k in range(3):
            h.head_w[g][k] = weights[1 + g * 3 + k].data_ptr()


class HeadBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, proj: _Proj, want_refs: bool, ncls: int, y3, prev_center, *weights):
        ctx.set_materialize_grads(False)
        V, B, Q, _ = y3.shape
        dev = y3.device
        y3 = y3.contiguous()
        prev_center = prev_center[..., :3].contiguous()
        weights = [w if w.is_contiguous() else w.contiguous() for w in weights]
        packed = torch.empty(int(lib.dpft_decoder_packed_head_floats()), dtype=torch.float32, device=dev)
        hw = (C.c_void_p * 12)(*[w.data_ptr() for w in weights[1:]])
        lib.call("dpft_decoder_pack_head_f32", weights[0].data_ptr(), C.byref(hw), V, ncls, packed.data_ptr(), stream())
        x = torch.empty((B, Q, 16), dtype=torch.float32, device=dev)
        center, size = torch.empty((B, Q, 3), dtype=torch.float32, device=dev), torch.empty((B, Q, 3), dtype=torch.float32, device=dev)
        angle, cls = torch.empty((B, Q, 2), dtype=torch.float32, device=dev), torch.empty((B, Q, ncls), dtype=torch.float32, device=dev)
        refs = torch.empty((V, B, Q, 2) if want_refs else (0,), dtype=torch.float32, device=dev)
        h = HeadTrain()
        proj.fill(h)
        _fill_head_weights(h, weights, packed)
        h.y3, h.prev_center, h.num_classes = y3.data_ptr(), prev_center.data_ptr(), ncls
        h.x, h.center, h.size, h.angle, h.cls = (t.data_ptr() for t in (x, center, size, angle, cls))
        h.refs = refs.data_ptr() if want_refs else None
        lib.call("dpft_head_train_fwd_f32", C.byref(h), B, Q, V, stream())
        ctx.save_for_backward(y3, prev_center, packed, *weights)
        ctx.proj, ctx.meta = proj, (ncls, want_refs)
        return x, center, size, angle, cls, refs

    @staticmethod
    def backward(ctx, dx, dcenter, dsize, dangle, dcls, drefs):
        y3, prev_center, packed, *weights = ctx.saved_tensors
        ncls, want_refs = ctx.meta
        V, B, Q, _ = y3.shape
        dev = y3.device
        R, W = B * Q, _HR["FLOATS"]
        rows = torch.empty((R, W), dtype=torch.float32, device=dev)
        dy3 = torch.empty_like(y3)
        dcp = torch.empty((B, Q, 3), dtype=torch.float32, device=dev)
        h = HeadTrain()
        ctx.proj.fill(h)
        _fill_head_weights(h, weights, packed)
        h.y3, h.prev_center, h.num_classes = y3.data_ptr(), prev_center.data_ptr(), ncls
        keep = [None if t is None else t.contiguous() for t in (dx, dcenter, dsize, dangle, dcls, drefs if want_refs else None)]
        h.dx, h.dcenter, h.dsize, h.dangle, h.dcls, h.drefs = (None if t is None else t.data_ptr() for t in keep)
        h.dy3, h.dcenter_prev, h.rows = dy3.data_ptr(), dcp.data_ptr(), rows.data_ptr()
        lib.call("dpft_head_train_bwd_f32", C.byref(h), B, Q, V, stream())
        X = _HR
        specs, off = [(X["DX"], 16, X["Y3C"], 16 * V, 0)], 16 * 16 * V                              # g_red (16, 16V)
        o0 = off
        specs.append((X["D1"], 64, X["X"], 16, off)); off += 64 * 16                              # "rgo,rk->gok"
        o3 = off
        for g in range(4):
            specs.append((X["D2"] + 16 * g, 16, X["H1"] + 16 * g, 16, off)); off += 256           # "rgo,rgk->gok"
        o6 = off
        for g in range(4):
            specs.append((X["DO"] + 16 * g, 16, X["H2"] + 16 * g, 16, off)); off += 256
        res = _rows_outer(rows.view(1, R, W), specs, off)[0]
        g_red = res[:16 * 16 * V].view(16, 16 * V)
        g0, g3, g6 = res[o0:o3].view(4, 16, 16), res[o3:o6].view(4, 16, 16), res[o6:off].view(4, 16, 16)
        grads = [g_red]
        for g, nout in enumerate((3, 3, 2, ncls)):
            grads += [g0[g], g3[g], g6[g, :nout]]
        return (None, None, None, dy3, dcp, *grads)


def head_block(layer, head, proj: _Proj, y3, prev_center, want_refs: bool):
    """-> (x, out dict, refs (V,B,Q,2) or None)."""
    x, center, size, angle, cls, refs = HeadBlockFn.apply(proj, want_refs, head.num_classes, y3, prev_center,
                                                         *head_params(layer, head))
    from collections import OrderedDict
    out = OrderedDict([("center", center), ("size", size), ("angle", angle), ("class", cls)])
    return x, out, (refs if want_refs else None)
