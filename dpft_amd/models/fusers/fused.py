"""Host side of the fused inference decoder (dpft_amd/csrc/decoder.hip): builds the C-ABI parameter
structs from an ``IMPFusion`` module and runs 2*i_iter + 1 kernels instead of ~700 eager ops."""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from dpft_amd.hip.lib import DecoderFwd, DecoderView, Pyramid, lib, make_pyramid, stream, weights_generation


def supported(fuser: nn.Module) -> bool:
    from dpft_amd.models.heads.detection import LinearDetectionHead
    try:
        ok = (fuser.d_model == 16 and fuser.d_ffn == 32 and fuser.norm and fuser.reduction == "linear"
              and fuser.activation == "Mish" and 1 <= fuser.m_views <= 4
              and all(h == 8 for h in fuser.n_heads)
              and all(l * p <= 20 and l <= 5 for l, p in zip(fuser.n_levels, fuser.n_points)))
        for h in fuser.heads:
            ok = ok and isinstance(h, LinearDetectionHead) and h.num_reg_layers == 3 and h.num_cls_layers == 3 \
                and not h.bias and h.num_classes <= 16
        return bool(ok)
    except AttributeError:
        return False


def _view_struct(ml: nn.Module) -> Tuple[DecoderView, list]:
    a = ml.ms_deform_attn
    tensors = [ml.self_attn.in_proj_weight, ml.self_attn.in_proj_bias, ml.self_attn.out_proj.weight,
               ml.self_attn.out_proj.bias, ml.norm1.weight, ml.norm1.bias,
               a.sampling_offsets.weight, a.sampling_offsets.bias, a.attention_weights.weight, a.attention_weights.bias,
               a.value_proj.weight, a.value_proj.bias, a.output_proj.weight, a.output_proj.bias,
               ml.norm2.weight, ml.norm2.bias, ml.ffn1.weight, ml.ffn1.bias, ml.ffn2.weight, ml.ffn2.bias,
               ml.norm3.weight, ml.norm3.bias]
    for t in tensors:
        assert t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda
    return DecoderView(*[t.data_ptr() for t in tensors]), tensors


class FusedDecoder:
    def __init__(self, fuser: nn.Module):
        self.fuser = fuser
        self._key = None

    def _build(self):
        """(Re)pack the parameters into the kernels' transposed blobs whenever a weight changed."""
        f = self.fuser
        plist = self.__dict__.get("_plist")
        if plist is None:               # walking the module tree costs ~1 ms per forward; the Parameter objects are fixed
            plist = self._plist = list(f.parameters())
        key = (weights_generation(),) + tuple((p.data_ptr(), p._version) for p in plist)
        if key == self._key:
            return
        V, I = f.m_views, f.i_iter
        dev = f.query.device
        nv, nh = int(lib.dpft_decoder_packed_infer_floats(f.n_queries)), int(lib.dpft_decoder_packed_head_floats())
        self.packed_views = torch.empty(I * V * nv, dtype=torch.float32, device=dev)
        self.packed_heads = torch.empty(I * nh, dtype=torch.float32, device=dev)
        d = DecoderFwd()
        pos = f.query_embedding.weight
        assert pos.is_contiguous() and pos.dtype == torch.float32
        layers = list(f.mpfusion.values())
        for it, layer in enumerate(layers):
            # hand-over to the next layer's self-attention: its in_proj composed with this layer's view reduction
            nxt = None
            if it + 1 < len(layers):
                nxt = (C.c_void_p * V)(*[ml.self_attn.in_proj_weight.data_ptr()
                                         for ml in layers[it + 1].ml_fusion_layers.values()])
            red = layer.reduction_layer.weight
            assert red.is_contiguous() and red.dtype == torch.float32
            for v, ml in enumerate(layer.ml_fusion_layers.values()):
                view, _keep = _view_struct(ml)
                lib.call("dpft_decoder_pack_infer_f32", C.byref(view), f.n_levels[v], f.n_points[v], red.data_ptr(),
                         None if nxt is None else C.cast(nxt, C.c_void_p), v, V, pos.data_ptr(),
                         f.query.data_ptr() if it == 0 else None, f.n_queries,
                         self.packed_views.data_ptr() + (it * V + v) * nv * 4, stream())
            head = f.heads[it]
            hw = (C.c_void_p * 12)()
            for bi, name in enumerate(("center", "size", "angle", "class")):
                seq = head.layers[name + "_head"]
                for li, idx in enumerate((0, 3, 6)):
                    w = seq[idx].weight
                    assert w.is_contiguous() and w.dtype == torch.float32
                    hw[bi * 3 + li] = w.data_ptr()
            red = layer.reduction_layer.weight
            assert red.is_contiguous() and red.dtype == torch.float32
            lib.call("dpft_decoder_pack_head_f32", red.data_ptr(), C.byref(hw), V, head.num_classes,
                     self.packed_heads.data_ptr() + it * nh * 4, stream())
        # iteration 0's self-attention output depends on the weights alone (its input is the learned query table): made
        # here, once per weight version, instead of by the first launch of every forward (DPFT_DEC_ATTN0=0: A/B switch)
        self.attn0 = None
        if os.environ.get("DPFT_DEC_ATTN0", "1") != "0":
            self.attn0 = torch.empty(V * f.n_queries * 16, dtype=torch.float32, device=dev)
            lib.call("dpft_decoder_attn0_f32", self.packed_views.data_ptr(), pos.data_ptr(), f.n_queries, V,
                     self.attn0.data_ptr(), stream())
        d.attn0 = None if self.attn0 is None else self.attn0.data_ptr()
        d.V, d.iters, d.Q, d.num_classes = V, I, f.n_queries, f.heads[0].num_classes
        for v in range(V):
            d.n_points[v] = f.n_points[v]
        d.packed_views, d.packed_heads = self.packed_views.data_ptr(), self.packed_heads.data_ptr()
        d.query0, d.pos = f.query.data_ptr(), f.query_embedding.weight.data_ptr()
        self.desc = d
        self._key = key

    @torch.no_grad()
    def prepare(self, batch: List[Dict[str, torch.Tensor]], shape: List[torch.Tensor],
                projection: List[Tuple[torch.Tensor, torch.Tensor]], out: Dict[str, torch.Tensor],
                flags: Optional[List[bool]] = None):
        """Fill the C-ABI descriptor for these inputs (allocates outputs + scratch); `launch()` then runs it."""
        f = self.fuser
        self._build()
        d = self.desc
        V, Q = f.m_views, f.n_queries
        center0 = out["center"][..., :3].contiguous().float()
        B, dev = center0.shape[0], center0.device
        keep = [[l if l.is_contiguous() else l.contiguous() for l in levels.values()] for levels in batch]
        pyrs = (Pyramid * V)()
        for v in range(V):
            pyrs[v] = make_pyramid(keep[v])
        strides = {s.stride(0) for s in shape if s.dtype == torch.int64 and s.dim() == 2 and s.stride(1) == 1 and s.shape[1] >= 2}
        if len(strides) == 1 and all(s.dtype == torch.int64 and s.dim() == 2 and s.stride(1) == 1 for s in shape):
            shapes, d.shape_stride = list(shape), strides.pop()      # the dataset's rows, read in place
        else:
            shapes, d.shape_stride = [s[:, :2].to(torch.int64).contiguous() for s in shape], 2
        Ts = [t.contiguous().float() for t, _ in projection]
        Ps = [p.contiguous().float() for _, p in projection]
        d.B = B
        d.pyr = C.cast(pyrs, C.c_void_p)
        d.center0 = center0.data_ptr()
        for v in range(V):
            d.T[v], d.P[v], d.shape[v] = Ts[v].data_ptr(), Ps[v].data_ptr(), shapes[v].data_ptr()
            d.p_rows[v], d.has_t[v] = Ps[v].shape[1], (-1 if flags is None else int(flags[v]))      # -1: decided on the device
        work = torch.empty(int(lib.dpft_decoder_work_floats(B, Q, V)), dtype=torch.float32, device=dev)
        ncls = d.num_classes
        res = (torch.empty((B, Q, 3), dtype=torch.float32, device=dev), torch.empty((B, Q, 3), dtype=torch.float32, device=dev),
               torch.empty((B, Q, 2), dtype=torch.float32, device=dev), torch.empty((B, Q, ncls), dtype=torch.float32, device=dev))
        d.work = work.data_ptr()
        d.center, d.size, d.angle, d.cls = (t.data_ptr() for t in res)
        self._live = (keep, pyrs, shapes, Ts, Ps, center0, work)      # referenced by the descriptor
        self._res = res

    def launch(self):
        """One C-ABI call: i_iter x (attention scores | cross attention + FFN | view reduction + heads + next refs)."""
        lib.call("dpft_decoder_forward_f32", C.byref(self.desc), stream())
        res = self._res
        return OrderedDict([("center", res[0]), ("size", res[1]), ("angle", res[2]), ("class", res[3])])

    def __call__(self, batch, shape, projection, out, flags):
        self.prepare(batch, shape, projection, out, flags)
        return self.launch()
