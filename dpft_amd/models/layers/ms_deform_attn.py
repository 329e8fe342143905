"""Multi-scale deformable attention module, MI355X-native.

Mirror of ``src/dprt/models/layers/ms_deform_attn.py`` (MSDeformAttnFunction :27-68, MSDeformAttn
:71-217): same parameters / init (``sampling_offsets``, ``attention_weights``, ``value_proj``,
``output_proj``), same ``forward`` signature, plus the fused hot path ``forward_levels`` that reads
the NHWC FPN levels in place ("sample-then-project", include/dpft_hip.h dpft_xattn_*), which is
what ``MLFusion.forward_cross_attn`` uses.
"""
from __future__ import annotations

import math
import warnings
from typing import List, Sequence

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.init import xavier_uniform_

from dpft_amd.hip import ops


class MSDeformAttnFunction(Function):
    """Operator-level drop-in for the MSDA extension (ms_deform_attn.py:27-68) on the C-ABI."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        value, sampling_locations = value.contiguous(), sampling_locations.contiguous()
        attention_weights = attention_weights.contiguous()
        shapes = value_spatial_shapes.to(torch.int64).contiguous()
        lsi = value_level_start_index.to(torch.int64).contiguous()
        out = ops.msda_fwd(value, shapes, lsi, sampling_locations, attention_weights)
        ctx.save_for_backward(value, shapes, lsi, sampling_locations, attention_weights)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, attn = ctx.saved_tensors
        gv, gl, ga = ops.msda_bwd(value, shapes, lsi, loc, attn, grad_output.contiguous())
        return gv, None, None, gl, ga, None


class PyramidState:
    """One view's FPN pyramid for one forward pass: detached level tensors shared by every
    cross-attention call of the pass + lazily zero-initialised gradient buffers that all of those
    calls accumulate into (fp32 atomics), handed to autograd exactly once by ``_PyramidHub``."""

    REPLICAS = 32            # gradient replicas of tiny levels (see dpft_pyramid.grad_replicas)
    TINY_PIXELS = 256        # levels with H*W <= this are replicated

    def __init__(self, levels: Sequence[torch.Tensor]):
        self.levels = [l.detach().contiguous() for l in levels]
        self.grads = None
        self.rep = None

    def _alloc(self, with_replicas: bool):
        """All gradient buffers of the view from ONE zero-filled allocation (one fill launch instead of one per level and
        replica stack); chunks start on 64-element boundaries."""
        shapes = [tuple(l.shape) for l in self.levels]
        reps = [(self.REPLICAS,) + sh if with_replicas and sh[1] * sh[2] <= self.TINY_PIXELS else None for sh in shapes]
        sizes = [int(torch.Size(sh).numel()) for sh in shapes] + [int(torch.Size(r).numel()) if r else 0 for r in reps]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 63) // 64 * 64
        l0 = self.levels[0]
        flat = torch.empty(total, dtype=l0.dtype, device=l0.device)
        ops.memops([(flat, None)])      # (these buffers only exist on the device path: the HIP kernels add into them)
        n = len(shapes)
        self.grads = [flat[offs[i]:offs[i] + sizes[i]].view(shapes[i]) for i in range(n)]
        if with_replicas:
            self.rep = [flat[offs[n + i]:offs[n + i] + sizes[n + i]].view(reps[i]) if reps[i] else None for i in range(n)]

    def grad_buffers(self) -> List[torch.Tensor]:
        if self.grads is None:
            self._alloc(False)
        return self.grads

    def replicated_grad_buffers(self) -> List[torch.Tensor]:
        """Like grad_buffers(), but tiny levels come as (R,B,H,W,C) replica stacks that `finish()` folds back."""
        if self.grads is None:
            self._alloc(True)
        elif self.rep is None:           # plain buffers exist already (mixed use): add the replica stacks
            self.rep = [torch.zeros((self.REPLICAS,) + tuple(l.shape), dtype=l.dtype, device=l.device)
                        if l.shape[1] * l.shape[2] <= self.TINY_PIXELS else None for l in self.levels]
        return [r if r is not None else t for r, t in zip(self.rep, self.grads)]

    def finish(self):
        if self.rep is not None and self.grads is not None:
            for g, r in zip(self.grads, self.rep):
                if r is not None:
                    ops.sum_leading([r], g.shape, out=g, accumulate=True)      # g += sum of the replicas, one launch
        self.rep = None


class _PyramidHub(Function):
    @staticmethod
    def forward(ctx, state: PyramidState, *levels):
        ctx.state = state
        ctx.n = len(levels)
        ctx.set_materialize_grads(False)      # the token carries ordering only: its consumers return no gradient for it
        return levels[0].new_empty(())      # (never read)

    @staticmethod
    def backward(ctx, gtoken):
        ctx.state.finish()
        g = ctx.state.grads
        ctx.state.grads = None
        return (None, *(g if g is not None else [None] * ctx.n))


def make_pyramid_state(levels: Sequence[torch.Tensor]):
    """-> (state, token).  ``token`` threads the autograd dependency from every cross-attention
    call back to the level tensors."""
    state = PyramidState(levels)
    token = _PyramidHub.apply(state, *levels)
    return state, token


class _XAttnFn(Function):
    @staticmethod
    def forward(ctx, state: PyramidState, token, ref, off, attn, Wv, bv, n_heads: int, n_points: int):
        ref, off, attn = ref.contiguous(), off.contiguous(), attn.contiguous()
        Wv, bv = Wv.contiguous(), bv.contiguous()
        out, samp, mass = ops.xattn_fwd(state.levels, ref, off, attn, Wv, bv, n_heads, n_points)
        ctx.state, ctx.n_heads, ctx.n_points = state, n_heads, n_points
        ctx.save_for_backward(ref, off, attn, Wv, bv, samp, mass)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        ref, off, attn, Wv, bv, samp, mass = ctx.saved_tensors
        state = ctx.state
        gout = gout.contiguous()
        goff, gattn, gref = ops.xattn_bwd(state.levels, state.grad_buffers(), ref, off, attn, Wv, bv, gout,
                                          ctx.n_heads, ctx.n_points)
        B, Q, C = gout.shape
        M = ctx.n_heads
        g4 = gout.view(B, Q, M, C // M)
        gWv = torch.einsum("bqmd,bqmc->mdc", g4, samp).reshape(C, C)
        gbv = torch.einsum("bqmd,bqm->md", g4, mass).reshape(C)
        return None, None, gref, goff, gattn, gWv, gbv, None, None      # token: ordering only (_PyramidHub)


class MSDeformAttn(nn.Module):
    """Multi-scale deformable attention with the parameter set, initial values and call signatures of the reference's
    module (src/dprt/models/layers/ms_deform_attn.py:75-217) on the HIP operators.  Parameter names / shapes are the
    state-dict contract; the initial values are pinned by tests/golden/msda_init.npz (the reference's seeded init)."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        head_dim, rest = divmod(d_model, n_heads)
        if rest:
            raise ValueError(f"d_model must be divisible by n_heads, but got {d_model} and {n_heads}")
        if head_dim & (head_dim - 1):
            warnings.warn("d_model // n_heads should be a power of 2")
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        slots = n_heads * n_levels * n_points
        # (creation order = the order the reference draws from the RNG: a seeded module gets the reference's values)
        self.sampling_offsets = nn.Linear(d_model, 2 * slots)
        self.attention_weights = nn.Linear(d_model, slots)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    @torch.no_grad()
    def _reset_parameters(self):
        """Initial state (ms_deform_attn.py:117-136): offsets and attention logits start from zero weights, so every query
        first looks at a fixed star -- head m along direction 2 pi m / n_heads, scaled to the unit square's border, point
        p at p + 1 steps -- with uniform weights; the two projections are Xavier-uniform with zero bias."""
        angle = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        ray = torch.stack((angle.cos(), angle.sin()), dim=-1)
        ray = ray / ray.abs().amax(dim=-1, keepdim=True)                                # onto the border of [-1, 1]^2
        steps = torch.arange(1, self.n_points + 1, dtype=torch.float32)
        star = ray[:, None, None, :] * steps[None, None, :, None]                       # (heads, 1, points, 2)
        star = star.expand(self.n_heads, self.n_levels, self.n_points, 2)
        self.sampling_offsets.weight.zero_()
        self.sampling_offsets.bias = nn.Parameter(star.reshape(-1).clone())
        self.attention_weights.weight.zero_()
        self.attention_weights.bias.zero_()
        for proj in (self.value_proj, self.output_proj):      # (this order: value_proj draws first)
            xavier_uniform_(proj.weight)
            proj.bias.zero_()

    def _offsets_and_weights(self, query):
        N, Len_q, _ = query.shape
        # The bias is folded into the GEMM (ones column) instead of nn.Linear's addmm: the (B*Q x 320) column
        # reduction that otherwise produces this bias' gradient returned garbage under hipGraph replay
        # (ROCm 7.0 / torch 2.10; weight gradients and every other bias were unaffected).
        w_aug = torch.cat((self.sampling_offsets.weight, self.sampling_offsets.bias[:, None]), dim=1)
        q_aug = torch.cat((query, torch.ones_like(query[..., :1])), dim=-1)
        off = F.linear(q_aug, w_aug).view(N, Len_q, self.n_heads, self.n_levels, self.n_points, 2)
        aw = self.attention_weights(query).view(N, Len_q, self.n_heads, self.n_levels * self.n_points)
        aw = F.softmax(aw, -1).view(N, Len_q, self.n_heads, self.n_levels, self.n_points)
        return off, aw

    def forward_levels(self, query, reference_points, state: PyramidState, token):
        """Fused hot path.  query (N,Lq,C) [with pos]; reference_points (N,Lq,2) (the same 2-D point is
        used for every level, mpfusion.py:190); state/token from ``make_pyramid_state``."""
        assert len(state.levels) == self.n_levels
        off, aw = self._offsets_and_weights(query)
        out = _XAttnFn.apply(state, token, reference_points, off, aw, self.value_proj.weight,
                             self.value_proj.bias, self.n_heads, self.n_points)
        return self.output_proj(out)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        """The reference's call signature (ms_deform_attn.py:138-217) on the operator-level C-ABI (dpft_msda_fwd / bwd):
        query (N, Lq, C); reference_points (N, Lq, L, 2) in [0, 1] or (N, Lq, L, 4) boxes; input_flatten (N, sum HW, C);
        input_spatial_shapes (L, 2) rows (H, W); input_level_start_index (L,); input_padding_mask (N, sum HW) True = pad."""
        N, Lq, _ = query.shape
        Lin = input_flatten.shape[1]
        hw = input_spatial_shapes.to(torch.long)
        if int(hw.prod(dim=1).sum()) != Lin:
            raise ValueError("input_spatial_shapes do not add up to the flattened input length")
        if not (reference_points.shape[2] == hw.shape[0] == input_level_start_index.shape[0] == self.n_levels):
            raise ValueError("reference_points / spatial shapes / level start indices disagree with n_levels")
        value = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            value = torch.where(input_padding_mask[..., None], value.new_zeros(()), value)
        value = value.view(N, Lin, self.n_heads, self.d_model // self.n_heads)
        off, aw = self._offsets_and_weights(query)
        anchor = reference_points[:, :, None, :, None, :]                     # broadcast over heads and points
        kind = reference_points.shape[-1]
        if kind == 2:          # a point per level: offsets are in pixels of that level -> normalise by (W, H)
            wh = input_spatial_shapes.flip(-1).to(off.dtype)
            loc = anchor + off / wh[None, None, None, :, None, :]
        elif kind == 4:        # a box per level (cx, cy, w, h): offsets in units of half the box, spread over the points
            loc = anchor[..., :2] + off * (anchor[..., 2:] * (0.5 / self.n_points))
        else:
            raise ValueError(f"Last dim of reference_points must be 2 or 4, but get {kind} instead.")
        out = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index, loc, aw, self.im2col_step)
        return self.output_proj(out)
