"""Fused HIP training blocks of the fusion decoder (dpft_amd/csrc/decoder_train.hip) as autograd Functions.

``SelfAttnBlocksFn``: the self-attention block of every view of one MPFusion layer
(src/dprt/models/fusers/mpfusion.py:122-148) in one forward launch and two backward launches.  Dropout masks are
regenerated in the backward from a device-side seed, so the op is hipGraph-capturable: the seed lives in a
persistent int64 tensor that the captured ``advance_seed`` kernels bump on every replay.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import torch

from dpft_amd.hip.lib import SaParams, lib, stream

_SA_SIZES = (768, 48, 256, 16, 16, 16)        # in_proj_weight, in_proj_bias, out_proj.weight, .bias, norm1.weight, .bias
_seed_state: Dict[torch.device, torch.Tensor] = {}


def advance_seed(device: torch.device) -> torch.Tensor:
    """Snapshot of the dropout seed for this decoder forward; the persistent state moves on (captured in graphs)."""
    st = _seed_state.get(device)
    if st is None:
        st = _seed_state[device] = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(device)
    snap = st.clone()
    st.add_(0x9E3779B97F4A7C15 & (2 ** 62 - 1))
    return snap


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
    def forward(ctx, x, pos, seed, salt: int, p_drop: float, *params):
        V = len(params) // 6
        if x.stride()[1:] != (16, 1):
            x = x.contiguous()
        pos = pos.contiguous()
        params = [p if p.is_contiguous() else p.contiguous() for p in params]
        B, Q, _ = x.shape
        dev = x.device
        y1 = torch.empty((V, B, Q, 16), dtype=torch.float32, device=dev)
        attn, zhat = torch.empty_like(y1), torch.empty_like(y1)
        lse = torch.empty((V, B, Q, 8), dtype=torch.float32, device=dev)
        rstd = torch.empty((V, B, Q), dtype=torch.float32, device=dev)
        arr = _structs(params, V)
        lib.call("dpft_selfattn_train_fwd_f32", C.cast(arr, C.c_void_p), V, x.data_ptr(), x.stride(0) if B > 1 else 0,
                 pos.data_ptr(), float(p_drop), seed.data_ptr(), int(salt), y1.data_ptr(), lse.data_ptr(),
                 attn.data_ptr(), zhat.data_ptr(), rstd.data_ptr(), B, Q, stream())
        ctx.save_for_backward(x, pos, seed, lse, attn, zhat, rstd, *params)
        ctx.meta = (V, int(salt), float(p_drop))
        return y1

    @staticmethod
    def backward(ctx, dy1):
        x, pos, seed, lse, attn, zhat, rstd, *params = ctx.saved_tensors
        V, salt, p_drop = ctx.meta
        B, Q, _ = x.shape
        dev = x.device
        dy1 = dy1.contiguous()
        per_view = sum(_SA_SIZES)
        flat = torch.zeros(V * per_view, dtype=torch.float32, device=dev)
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
        lib.call("dpft_selfattn_train_bwd_f32", C.cast(arr, C.c_void_p), V, x.data_ptr(), x.stride(0) if B > 1 else 0,
                 pos.data_ptr(), p_drop, seed.data_ptr(), salt, dy1.data_ptr(), lse.data_ptr(), attn.data_ptr(),
                 zhat.data_ptr(), rstd.data_ptr(), C.cast(garr, C.c_void_p), dx.data_ptr(), dxp.data_ptr(),
                 scratch.data_ptr(), B, Q, stream())
        return (dx.sum(0), dxp.sum((0, 1)), None, None, None, *grads)


def self_attn_blocks(layers, x, pos, seed, salt: int, p_drop: float):
    """y1 (V,B,Q,16) of the V MLFusion layers' self-attention blocks; x (B,Q,16), pos (Q,16)."""
    params = [t for ml in layers for t in sa_params(ml)]
    return SelfAttnBlocksFn.apply(x, pos, seed, salt, p_drop, *params)
