"""Iterative multi-perspective fusion decoder, MI355X-native.

Mirror of ``src/dprt/models/fusers/mpfusion.py``: MLFusion (:16-263), MPFusion (:266-514, 'linear'
/ 'mean' / 'max' reductions), IMPFusion (:517-745) with identical parameter names.  Differences in
*execution only*:
  * cross attention reads the NHWC pyramid levels in place through the fused HIP kernel
    (no flatten/cat at :179, no dense value_proj, no ``value`` tensor);
  * the host decision ``transformation.any()`` (:647) is taken once per forward for all views
    (one sync) instead of once per view per iteration;
  * reference points are computed out-of-place (same values).
"""
from __future__ import annotations

from copy import deepcopy
from functools import partial
from typing import Any, Callable, Dict, List, Tuple, Union

import torch
from torch import nn

from dpft_amd.models.layers.ms_deform_attn import MSDeformAttn, make_pyramid_state
from dpft_amd.models.utils.transformations import cart2spher


class MLFusion(nn.Module):
    def __init__(self, d_model: int = 256, d_ffn: int = 1024, n_levels: int = 1, n_heads: int = 1,
                 n_points: int = 1, ffn_layer: str = "Linear", activation: str = "ReLU", dropout: float = 0.0,
                 norm: bool = False, **kwargs):
        super().__init__()
        if ffn_layer != "Linear":
            raise ValueError("dpft_amd MLFusion: only ffn_layer='Linear' is on the hot path")
        self.d_model, self.d_ffn, self.n_levels, self.n_heads, self.n_points = d_model, d_ffn, n_levels, n_heads, n_points
        self.ffn_layer, self.activation, self.dropout, self.norm = ffn_layer, activation, dropout, norm
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout, batch_first=True)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.flatten1 = nn.Flatten(start_dim=1, end_dim=2)
        self.ms_deform_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.ffn1 = nn.Linear(d_model, d_ffn)
        self.activation1 = getattr(nn, activation)()
        self.dropout3 = nn.Dropout(dropout)
        self.ffn2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "MLFusion":
        return cls(**config)

    @staticmethod
    def with_pos_embed(tensor, pos=None):
        return tensor if pos is None else tensor + pos

    def forward_self_attn(self, query, query_positions=None):
        q = k = self.with_pos_embed(query, query_positions)
        out = self.self_attn(query=q, key=k, value=query, need_weights=False)[0]
        out = query + self.dropout1(out)
        return self.norm1(out) if self.norm else out

    def forward_cross_attn(self, query, pyramid, reference_points, query_positions=None):
        """pyramid = (PyramidState, token) of this view (see IMPFusion.forward)."""
        state, token = pyramid
        out = self.ms_deform_attn.forward_levels(self.with_pos_embed(query, query_positions), reference_points,
                                                 state, token)
        out = query + self.dropout2(out)
        return self.norm2(out) if self.norm else out

    def forward_ffn(self, query):
        out = self.ffn2(self.dropout3(self.activation1(self.ffn1(query))))
        out = query + self.dropout4(out)
        return self.norm3(out) if self.norm else out

    def forward(self, query, pyramid, reference_points, query_positions=None):
        out = self.forward_self_attn(query=query, query_positions=query_positions)
        out = self.forward_cross_attn(query=out, pyramid=pyramid, reference_points=reference_points,
                                      query_positions=query_positions)
        return self.forward_ffn(query=out)


class MPFusion(nn.Module):
    def __init__(self, m_views: int, d_model: int = 256, d_ffn: int = 1024, n_levels: List[int] = None,
                 n_heads: List[int] = None, n_points: List[int] = None, ffn_layer: str = "Linear",
                 activation: str = "ReLU", dropout: float = 0.0, norm: bool = False, reduction: str = "mean",
                 **kwargs):
        super().__init__()
        if reduction not in {"mean", "max", "linear"}:
            raise ValueError(f"dpft_amd MPFusion supports reduction 'mean', 'max' or 'linear', got {reduction!r}")
        self.m_views, self.d_model, self.d_ffn = m_views, d_model, d_ffn
        self.n_levels = n_levels if n_levels is not None else [1] * m_views
        self.n_heads = n_heads if n_heads is not None else [1] * m_views
        self.n_points = n_points if n_points is not None else [1] * m_views
        self.ffn_layer, self.activation, self.dropout, self.norm, self.reduction = \
            ffn_layer, activation, dropout, norm, reduction
        self.ml_fusion_layers = nn.ModuleDict({
            "ms_deform_attn" + str(v): MLFusion(d_model, d_ffn, l, h, p, ffn_layer, activation, dropout, norm)
            for v, l, h, p in zip(range(m_views), self.n_levels, self.n_heads, self.n_points)})
        if reduction == "linear":
            self.reduction_layer = nn.Linear(m_views * d_model, d_model, bias=False)
        elif reduction == "mean":
            self.reduction_layer = partial(torch.mean, dim=-1)
        else:
            self.reduction_layer = partial(torch.max, dim=-1)

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "MPFusion":
        return cls(**config)

    def reduce(self, query, queries, query_positions):
        if self.reduction in {"mean", "max"}:
            return self.reduction_layer(queries)
        B, N = query.shape[:2]
        return self.reduction_layer(queries.reshape(B, N, self.d_model * self.m_views))   # :436-438

    use_fused_train = True      # CUDA: self-attention blocks of all views from the fused HIP kernels (train_fused.py)

    def fused_blocks_supported(self) -> bool:
        from dpft_amd.models.fusers import train_fused as _tf
        return self.use_fused_train and all(_tf.xf_supported(ml) for ml in self.ml_fusion_layers.values())

    def forward_fused_blocks(self, query, batch_views, refs, pos2d, seed, salt: int, batch=None):
        """(V,B,Q,16) outputs of every view's MLFusion from the fused HIP training kernels; refs (V,B,Q,2).  ``query`` (B,Q,16),
        or (Q,16) broadcast over ``batch`` elements; ``pos2d`` the (Q,16) table or a hub from ``train_fused.make_pos_hub``."""
        from dpft_amd.models.fusers import train_fused as _tf
        layers = list(self.ml_fusion_layers.values())
        p_drop = self.dropout if self.training else 0.0
        y1 = _tf.self_attn_blocks(layers, query, pos2d, seed, salt, p_drop, batch=batch)
        return _tf.xattn_ffn_blocks(layers, batch_views, y1, pos2d, refs, seed, salt, p_drop)

    def forward(self, query, batch, reference_points, query_positions, pos2d=None, seed=None, salt: int = 0):
        layers = list(self.ml_fusion_layers.values())
        if self.use_fused_train and query.is_cuda and pos2d is not None and seed is not None:
            from dpft_amd.models.fusers import train_fused as _tf
            if all(_tf.sa_supported(ml) for ml in layers):
                p_drop = self.dropout if self.training else 0.0
                y1 = _tf.self_attn_blocks(layers, query, pos2d, seed, salt, p_drop)
                if all(_tf.xf_supported(ml) for ml in layers):
                    y3 = _tf.xattn_ffn_blocks(layers, batch, y1, pos2d, torch.stack(reference_points), seed, salt, p_drop)
                    queries = y3.permute(1, 2, 3, 0)           # (B,N,C,V): channel-major / view-minor
                else:
                    outs = [ml.forward_ffn(ml.forward_cross_attn(y1[v], pyr, ref, query_positions))
                            for v, (ml, pyr, ref) in enumerate(zip(layers, batch, reference_points))]
                    queries = torch.stack(outs, dim=-1)
                return self.reduce(query, queries, query_positions)
        outs = [layer(query, pyr, ref, query_positions)
                for layer, pyr, ref in zip(layers, batch, reference_points)]
        queries = torch.stack(outs, dim=-1)            # (B,N,C,V): channel-major / view-minor (:496-509)
        return self.reduce(query, queries, query_positions)


class IMPFusion(nn.Module):
    def __init__(self, i_iter: int = 1, m_views: int = 1, d_model: int = 256, d_ffn: int = 1024,
                 n_queries: int = 100, n_levels: List[int] = None, n_heads: List[int] = None,
                 n_points: List[int] = None, q_init: str = "uniform_", ffn_layer: str = "Linear",
                 activation: str = "ReLU", dropout: float = 0.0, norm: bool = False, reduction: str = "mean",
                 head: nn.Module = None, **kwargs):
        super().__init__()
        self.i_iter, self.m_views, self.d_model, self.d_ffn, self.n_queries = i_iter, m_views, d_model, d_ffn, n_queries
        self.n_levels = n_levels if n_levels is not None else [1] * m_views
        self.n_heads = n_heads if n_heads is not None else [1] * m_views
        self.n_points = n_points if n_points is not None else [1] * m_views
        self.ffn_layer, self.activation, self.dropout, self.norm, self.reduction = \
            ffn_layer, activation, dropout, norm, reduction
        self.q_init = getattr(nn.init, q_init)
        if head is None:
            head = nn.Identity()
        self.mpfusion = nn.ModuleDict({
            "fusion" + str(i): MPFusion(m_views, d_model, d_ffn, self.n_levels, self.n_heads, self.n_points,
                                        ffn_layer, activation, dropout, norm, reduction)
            for i in range(i_iter)})
        self.heads = nn.ModuleList([deepcopy(head) for _ in range(i_iter)])
        self.query_embedding = nn.Embedding(n_queries, d_model)
        self.query = nn.Parameter(torch.empty((n_queries, d_model)))
        self.reset_parameters()

    @classmethod
    def from_config(cls, config: Dict[str, Any], **kwargs) -> "IMPFusion":
        return cls(**config, **kwargs)

    use_fused_inference = True       # eval + no_grad forward runs the fused HIP decoder when the config allows
    use_fused_train = True           # CUDA: layers run from the fused HIP training kernels when the config allows

    def reset_parameters(self) -> None:
        self.q_init(self.query)

    def __getstate__(self):          # the fused inference decoder holds packed device blobs + a native descriptor:
        st = self.__dict__.copy()    # rebuilt on the next eval forward (torch.save(model), deepcopy)
        st.pop("_fused_decoder", None)
        return st

    @staticmethod
    def get_reference_points(query, transformation, projection, shape, has_transformation: bool = None):
        """mpfusion.py:617-696 -> (B,N,2) ordered (u = x/W, v = y/H), clipped to [0,1]."""
        if has_transformation is None:
            has_transformation = bool(transformation.any())
        pts = query[..., :3]
        if has_transformation:
            hom = torch.cat((pts, torch.ones_like(pts[..., :1])), dim=-1)
            p = torch.einsum("bij,bkj->bki", transformation, hom)
            r, phi, roh = cart2spher(p[..., 0], p[..., 1], p[..., 2], degrees=True)
            pts = torch.stack((r, phi, roh), dim=-1)
        hom = torch.cat((pts[..., :3], torch.ones_like(pts[..., :1])), dim=-1)
        p = torch.einsum("bij,bkj->bki", projection, hom)
        w = p[..., 2]
        mask = w != 0
        safe = torch.where(mask, w, torch.ones_like(w))
        u = torch.where(mask, p[..., 0] / safe, p[..., 0])
        v = torch.where(mask, p[..., 1] / safe, p[..., 1])
        u = (u - 0) / (shape[:, 1].unsqueeze(1) - 0) * (1 - 0) + 0
        v = (v - 0) / (shape[:, 0].unsqueeze(1) - 0) * (1 - 0) + 0
        return torch.clip(torch.stack((u, v), dim=-1), min=0.0, max=1.0)

    @staticmethod
    def transformation_flags(projection: List[Tuple[torch.Tensor, torch.Tensor]]) -> List[bool]:
        """One host decision (one sync) for all views; the reference evaluates ``transformation.any()``
        per view per iteration (mpfusion.py:647)."""
        return [bool(f) for f in torch.stack([t.any() for t, _ in projection]).tolist()]

    def forward(self, batch: List[Dict[str, torch.Tensor]], shape: List[torch.Tensor],
                projection: List[Tuple[torch.Tensor, torch.Tensor]], out: Dict[str, torch.Tensor],
                has_transformation: List[bool] = None):
        B = out["center"].shape[0]
        if not self.training and not torch.is_grad_enabled() and self.use_fused_inference \
                and out["center"].is_cuda:
            fused = self.__dict__.get("_fused_decoder")
            if fused is None:
                from dpft_amd.models.fusers import fused as _f
                fused = self.__dict__["_fused_decoder"] = _f.FusedDecoder(self) if _f.supported(self) else False
            if fused:
                # has_transformation None: `transformation.any()` is evaluated on the device (a host read-back here
                # would stall the host until every encoder has finished: 4 ms of the forward latency)
                return fused(batch, shape, projection, out, has_transformation)
        flags = has_transformation if has_transformation is not None else self.transformation_flags(projection)
        pyramids = [make_pyramid_state(list(levels.values())) for levels in batch]
        seed = None
        if self.query.is_cuda:
            from dpft_amd.models.fusers import train_fused as _tf
            seed = _tf.advance_seed(self.query.device)
            layers = list(self.mpfusion.values())
            if self.use_fused_train and all(l.fused_blocks_supported() and _tf.head_supported(l, h)
                                            for l, h in zip(layers, self.heads)):
                # every layer = 3 fused forward launches (self attention | cross attention + FFN | reduction +
                # heads + next reference points) with hand-written backward kernels (train_fused.py)
                proj = _tf._Proj(projection, shape, flags)
                # query_embedding.weight is read by both blocks of every layer: its gradient is collected by one hub
                # (one summation launch in the backward instead of a reduction per block + autograd's add chain); the
                # learned query table goes in as it is -- the first layer broadcasts it over the batch itself
                pos_hub = _tf.make_pos_hub(self.query_embedding.weight)
                refs = _tf.reference_points(proj, out["center"])
                query = self.query
                for it, (layer, head) in enumerate(zip(layers, self.heads)):
                    y3 = layer.forward_fused_blocks(query, pyramids, refs, pos_hub, seed, it, batch=B)
                    query, out, refs = _tf.head_block(layer, head, proj, y3, out["center"], it + 1 < len(layers))
                return out
        query = self.query.unsqueeze(0).repeat(B, 1, 1)
        query_pos = self.query_embedding.weight.unsqueeze(0).repeat(B, 1, 1)
        for it, (layer, head) in enumerate(zip(self.mpfusion.values(), self.heads)):
            reference_points = [
                self.get_reference_points(out["center"][..., :3], p[0], p[1], s, f)
                for p, s, f in zip(projection, shape, flags)]
            query = layer(query, pyramids, reference_points, query_pos, pos2d=self.query_embedding.weight,
                          seed=seed, salt=it)
            out = head(query, out)
        return out


def build_mpfusion(*args, **kwargs):
    return IMPFusion.from_config(*args, **kwargs)
