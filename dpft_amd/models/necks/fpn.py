"""Feature Pyramid Network neck, MI355X-native.

Mirror of ``src/dprt/models/necks/fpn.py`` (FPN :11-83) around torchvision's
``FeaturePyramidNetwork(in_channels_list, out_channels, norm_layer=None)``: state-dict names
``fpn.inner_blocks.{i}.0.{weight,bias}`` (1x1) and ``fpn.layer_blocks.{i}.0.{weight,bias}`` (3x3,
pad 1); kaiming_uniform_(a=1) weights, zero bias; NHWC in / NHWC out.  One autograd node with a
hand-scheduled backward over the HIP conv / top-down kernels.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Any, Dict, List, Optional

import torch
from torch import nn

from dpft_amd.hip import ops
from dpft_amd.models.backbones.resnet import khwc


def _fpn_conv(cin, cout, k, pad) -> nn.Sequential:
    conv = nn.Conv2d(cin, cout, k, padding=pad, bias=True)
    nn.init.kaiming_uniform_(conv.weight, a=1)
    nn.init.constant_(conv.bias, 0)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    return nn.Sequential(conv)          # Conv2dNormActivation without norm/activation => key ".0."


class FeaturePyramidNetwork(nn.Module):
    def __init__(self, in_channels_list: List[int], out_channels: int):
        super().__init__()
        self.inner_blocks = nn.ModuleList([_fpn_conv(c, out_channels, 1, 0) for c in in_channels_list])
        self.layer_blocks = nn.ModuleList([_fpn_conv(out_channels, out_channels, 3, 1) for _ in in_channels_list])
        self.out_channels = out_channels


class _FPNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, fpn: FeaturePyramidNetwork, n: int, out_buffers, pos, *tensors: torch.Tensor):
        # ``pos`` (or None): per level (pos_x (W,K), pos_y (H,K)) -- the positional embedding that follows the neck
        # (embeddings/sinusoidal.py), added in the epilogue of the level's output conv; its backward is the identity
        xs = [t.contiguous() for t in tensors[:n]]
        K = fpn.out_channels
        lasts, outs, ci, cl = [None] * n, [None] * n, [None] * n, [None] * n
        for i in range(n - 1, -1, -1):
            B, H, W, C = xs[i].shape
            ci[i] = ops.conv_problem(B, H, W, C, K, 1, 1, 1, 0)
            conv = fpn.inner_blocks[i][0]
            lat = ops.fpn_lateral(ci[i], xs[i], khwc(conv.weight), conv.bias, top=lasts[i + 1] if i < n - 1 else None)
            lasts[i] = lat
            cl[i] = ops.conv_problem(B, H, W, K, K, 3, 3, 1, 1)
            conv = fpn.layer_blocks[i][0]
            outs[i] = ops.fpn_output(cl[i], lat, khwc(conv.weight), conv.bias, pos=None if pos is None else pos[i],
                                     out=None if out_buffers is None else out_buffers[i])
        ctx.fpn, ctx.n, ctx.xs, ctx.lasts, ctx.ci, ctx.cl = fpn, n, xs, lasts, ci, cl
        ctx.params = tensors[n:]
        ctx.x_needs = [t.requires_grad for t in tensors[:n]]
        # DP reducer attached by the trainer: weight / bias gradients are written straight into its bucket views
        # (each FPN parameter receives exactly one gradient per step) instead of going through AccumulateGrad
        direct = owner.grad_direct if owner is not None else None
        if direct is not None and any(direct.grad_buffer(p) is None for p in fpn.parameters()):
            direct.clear(list(fpn.parameters()))      # accumulated by autograd this step (see GradBucketReducer.set_overwritten)
            direct = None
        ctx.direct = direct
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        fpn, n, xs, lasts, ci, cl, direct = ctx.fpn, ctx.n, ctx.xs, ctx.lasts, ctx.ci, ctx.cl, ctx.direct
        grads: Dict[nn.Parameter, torch.Tensor] = {}

        def wgrad(cv, x, dy, conv):
            if direct is not None:          # bucket view of a conv weight: physically [K][kh][kw][C]
                ops.conv_wgrad_bias(cv, x, dy, out=direct.grad_buffer(conv.weight).permute(0, 2, 3, 1),
                                    bias_out=direct.grad_buffer(conv.bias))
            else:
                dw, db = ops.conv_wgrad_bias(cv, x, dy)
                grads[conv.weight] = dw.permute(0, 3, 1, 2)
                grads[conv.bias] = db

        # the data gradients' transposed weights of all convs of this pyramid in one launch (they were 2 n launches)
        need = [fpn.layer_blocks[i][0] for i in range(n) if douts[i] is not None] + \
               [fpn.inner_blocks[i][0] for i in range(n) if ctx.x_needs[i]]
        wts = dict(zip(need, ops.weight_transpose_many([khwc(c.weight) for c in need]))) if need else {}
        dxs: List[Optional[torch.Tensor]] = [None] * n
        g_prev = None
        for i in range(n):
            conv = fpn.layer_blocks[i][0]
            if douts[i] is None:
                g = torch.zeros_like(lasts[i])
                if direct is not None:      # this output was not used: its conv receives no gradient -- the bucket is not
                    direct.grad_buffer(conv.weight).zero_()      # cleared per step for overwritten parameters
                    direct.grad_buffer(conv.bias).zero_()
            else:
                do = douts[i].contiguous()
                wgrad(cl[i], lasts[i], do, conv)
                g = ops.conv_dgrad(cl[i], do, wts[conv])
            if g_prev is not None:
                ops.fpn_topdown_add_bwd_(g_prev, g)       # grad(last_i) += upsample_bwd(grad(last_{i-1}))
            conv = fpn.inner_blocks[i][0]
            wgrad(ci[i], xs[i], g, conv)
            if ctx.x_needs[i]:
                dxs[i] = ops.conv_dgrad(ci[i], g, wts[conv], out=ops.grad_sink(xs[i]))      # in place where the producer asked for it
            g_prev = g
        if direct is not None:
            direct.mark_ready_many(list(fpn.parameters()))
        return (None, None, None, None, None, *dxs, *[grads.get(p) for p in ctx.params])


class FPN(nn.Module):
    def __init__(self, in_channels_list: List[int], out_channels: int, norm_layer=None,
                 channel_last: bool = True, **kwargs):
        super().__init__()
        if norm_layer is not None:
            raise ValueError("dpft_amd FPN: norm_layer is not supported (no reference config uses it)")
        self.in_channels_list = in_channels_list
        self.out_channels = out_channels
        self.channel_last = channel_last
        self.fpn = FeaturePyramidNetwork(in_channels_list, out_channels)
        self.grad_direct = None     # optional DP reducer (grad_buffer / mark_ready_many), installed by the trainer

    def overwritten_parameters(self):
        """Parameters whose gradients ``_FPNFn.backward`` writes (not adds) into an attached reducer's bucket views."""
        return list(self.fpn.parameters())

    def __getstate__(self):          # the reducer belongs to a trainer, not to the module (torch.save(model), deepcopy)
        st = self.__dict__.copy()
        st["grad_direct"] = None
        return st

    @classmethod
    def from_config(cls, config: Dict[str, Any]):
        return cls(config["in_channels_list"], config["out_channels"], config.get("norm_layer"))

    def forward(self, batch: Dict[str, torch.Tensor], out_buffers: Optional[List[torch.Tensor]] = None,
                pos: Optional[List[Any]] = None) -> Dict[str, torch.Tensor]:
        """``out_buffers`` (NHWC, one per level): the pyramid is written there (e.g. the static input buffers of a captured
        decoder graph: no copy between the neck and the graph).  ``pos``: per level the (pos_x, pos_y) tables of the positional
        embedding that follows the neck -- added by the output convs themselves (the caller then skips the embedding)."""
        keys = list(batch.keys())
        xs = list(batch.values())
        if not self.channel_last:
            if pos is not None:
                raise ValueError("FPN: the fused positional embedding needs channel-last tensors")
            xs = [x.movedim(1, -1) for x in xs]
            out_buffers = None
        outs = _FPNFn.apply(self, self.fpn, len(xs), out_buffers, pos, *xs, *self.fpn.parameters())
        if not self.channel_last:
            outs = [o.movedim(-1, 1) for o in outs]
        return OrderedDict(zip(keys, outs))


def build_fpn(name: str, *args, **kwargs):
    if "fpn" in name.lower():
        return FPN.from_config(*args, **kwargs)
