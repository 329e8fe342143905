"""FLOPS / MACS / Parameters of one forward pass -- the three scalars ``Evaluator.evaluate_complexity`` logs
(src/dprt/evaluation/evaluator.py:70-94).  The reference asks deepspeed's ``get_model_profile`` (absent here, and its
module hooks would see nothing: the encoders and the decoder run from launch plans, not from ``nn.Module.forward``
calls), so the multiply-accumulates are counted where they are issued:

* convolutions (encoders, FPN): the library's own launch log of one eval forward (``dpft_profile_start`` /
  ``dpft_profile_get``: algorithmic 2 * M * K * kh * kw * C per launch), i.e. exactly the convs that ran;
* decoder: every ``nn.Linear`` under ``model.fuser`` once per forward on its rows (B * n_queries; the value projections
  on their view's pyramid tokens), the self-attention products (QK^T and AV) and in-projections, the deformable
  sampling (4 bilinear taps + the attention-weighted sum per sampled channel).

MACS counts multiply-accumulates of these matrix / sampling products; FLOPS = 2 * MACS.  Element-wise work (BatchNorm,
activations, softmax, LayerNorm), which deepspeed adds to FLOPS only, is not counted -- < 1 % of the total here.
"""
from __future__ import annotations

from typing import Dict

import torch
from torch import nn


@torch.no_grad()
def model_complexity(model: nn.Module, data: Dict[str, torch.Tensor]) -> Dict[str, float]:
    from dpft_amd.hip import ops
    from dpft_amd.models.layers.ms_deform_attn import MSDeformAttn
    was_training = model.training
    model.eval()
    model(data)                                   # plans / fused decoder built, nothing of the set-up inside the log
    torch.cuda.synchronize()
    ops.profile_start()
    try:
        model(data)
    finally:
        recs = ops.profile_collect()
    conv_macs = sum(r[1] for r in recs) / 2.0
    feats = model._encode_views(data)
    tokens = [sum(int(l.shape[1] * l.shape[2]) for l in feats[i].values()) for i in model.inputs]
    model.train(was_training)
    fuser = model.fuser
    B = int(next(iter(data.values())).shape[0])
    Q = int(fuser.n_queries)
    dec = 0.0
    for name, m in fuser.named_modules():
        if isinstance(m, nn.Linear):
            rows = B * Q
            if name.endswith("value_proj"):       # ...ml_fusion_layers.ms_deform_attn<view>.ms_deform_attn.value_proj
                view = int(name.split("ml_fusion_layers.ms_deform_attn")[1].split(".")[0])
                rows = B * tokens[view]
            dec += rows * m.in_features * m.out_features
        elif isinstance(m, nn.MultiheadAttention):
            E = m.embed_dim
            dec += B * Q * 3 * E * E              # in-projection (out_proj is an nn.Linear: counted above)
            dec += 2 * B * Q * Q * E              # QK^T and AV over all heads
        elif isinstance(m, MSDeformAttn):
            samples = B * Q * m.n_heads * m.n_levels * m.n_points
            dec += samples * (m.d_model // m.n_heads) * 5      # 4 bilinear taps + the weighted sum, per sampled channel
    macs = conv_macs + dec
    return {"FLOPS": 2.0 * macs, "MACS": macs, "Parameters": float(sum(p.numel() for p in model.parameters())),
            "MACS_conv": conv_macs, "MACS_decoder": dec}
