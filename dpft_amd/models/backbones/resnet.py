"""ResNet-50/101/152 backbone with intermediate returns, MI355X-native.

Mirror of ``src/dprt/models/backbones/resnet.py`` (BackboneBase :13-107, Backbone :110-176): same
constructor arguments, same ``state_dict`` names as the torchvision body behind
``IntermediateLayerGetter`` (``adjustment_layer.weight``, ``body.conv1.weight``,
``body.layer{1..4}.{i}.{conv,bn}{1,2,3}``, ``.downsample.{0,1}``), NHWC in / NHWC out.

Execution is a hand-scheduled pipeline of the HIP kernels in ``libdpft_hip.so``:
  conv (MFMA implicit GEMM, BN statistics fused in its epilogue)
  -> bn_finalize -> next conv with BN-apply+ReLU fused in its operand prologue
  -> ... -> bn_act (BN3 + residual/downsample-BN + ReLU).
The whole body is ONE autograd node whose backward is the hand-scheduled reverse pipeline
(bn_bwd two-pass, dgrad, wgrad with the same fused prologue).  nn.Conv2d / nn.BatchNorm2d are
used as parameter containers only (never called), which keeps names, shapes and default
initialisation identical to torchvision's.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Any, Dict, List, Optional

import torch
from torch import nn

from dpft_amd.hip import ops

DEPTHS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3), "resnet152": (3, 8, 36, 3)}


def khwc(w: torch.Tensor) -> torch.Tensor:
    """(O,I,kh,kw) parameter -> physical [O][kh][kw][I] view (copy only if the layout is wrong)."""
    v = w.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def _conv(cin, cout, k, stride=1, pad=0, bias=False) -> nn.Conv2d:
    m = nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=bias)
    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")   # torchvision ResNet init
    m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return m


class Bottleneck(nn.Module):
    """Parameter container with torchvision's Bottleneck (v1.5) names."""
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int, downsample: bool):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, stride=stride, pad=1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.stride = stride
        if downsample:
            self.downsample = nn.Sequential(_conv(inplanes, planes * 4, 1, stride=stride),
                                            nn.BatchNorm2d(planes * 4))
        else:
            self.downsample = None


class ResNetBody(nn.Module):
    """conv1/bn1/(relu,maxpool)/layer1..4 -- what IntermediateLayerGetter keeps (resnet.py:54-55)."""

    def __init__(self, depths, n_layers: int = 4):
        super().__init__()
        self.n_layers = n_layers
        self.conv1 = _conv(3, 64, 7, stride=2, pad=3)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for li, (n, planes) in enumerate(zip(depths[:n_layers], (64, 128, 256, 512))):
            blocks = []
            for b in range(n):
                stride = 2 if (li > 0 and b == 0) else 1
                blocks.append(Bottleneck(inplanes, planes, stride, downsample=(b == 0)))
                inplanes = planes * 4
            setattr(self, f"layer{li + 1}", nn.Sequential(*blocks))


# ------------------------------------------------------------------------------------------------
# hand-scheduled forward / backward
# ------------------------------------------------------------------------------------------------
def _bn_forward(bn: nn.BatchNorm2d, stats, cv: ops.Conv, train: bool) -> torch.Tensor:
    """-> BN block (4,K): mean, gamma*invstd, beta, invstd (batch statistics in train mode)."""
    if train:
        return ops.bn_finalize(stats, cv.tile_rows, cv.M, bn.weight, bn.bias, bn.eps, bn.momentum,
                               bn.running_mean, bn.running_var)
    return ops.bn_eval_params(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)


def _block_forward(blk: Bottleneck, x: torch.Tensor, train: bool, rec: Optional[dict]):
    B, H, W, Cin = x.shape
    planes = blk.conv1.out_channels
    c1 = ops.conv_problem(B, H, W, Cin, planes, 1, 1, 1, 0)
    y1, s1 = ops.conv_fwd(c1, x, khwc(blk.conv1.weight), want_stats=train)
    b1 = _bn_forward(blk.bn1, s1, c1, train)
    c2 = ops.conv_problem(B, H, W, planes, planes, 3, 3, blk.stride, 1)
    y2, s2 = ops.conv_fwd(c2, y1, khwc(blk.conv2.weight), pro=(b1, True), want_stats=train)
    b2 = _bn_forward(blk.bn2, s2, c2, train)
    c3 = ops.conv_problem(B, c2.OH, c2.OW, planes, planes * 4, 1, 1, 1, 0)
    y3, s3 = ops.conv_fwd(c3, y2, khwc(blk.conv3.weight), pro=(b2, True), want_stats=train)
    b3 = _bn_forward(blk.bn3, s3, c3, train)
    if blk.downsample is not None:
        cd = ops.conv_problem(B, H, W, Cin, planes * 4, 1, 1, blk.stride, 0)
        yd, sd = ops.conv_fwd(cd, x, khwc(blk.downsample[0].weight), want_stats=train)
        bd = _bn_forward(blk.downsample[1], sd, cd, train)
        out = ops.bn_act(y3, b3, res=yd, res_bnp=bd, relu=True)
    else:
        cd = yd = bd = None
        out = ops.bn_act(y3, b3, res=x, relu=True)
    if rec is not None:
        rec.update(x=x, y1=y1, y2=y2, y3=y3, yd=yd, out=out, b1=b1, b2=b2, b3=b3, bd=bd, c1=c1, c2=c2, c3=c3, cd=cd)
    return out


def _block_backward(blk: Bottleneck, rec: dict, dout: torch.Tensor, grads: Dict[nn.Parameter, torch.Tensor]):
    """dout: gradient wrt the block output (post-ReLU).  Returns the gradient wrt the block input."""
    x, y1, y2, y3, yd, out = rec["x"], rec["y1"], rec["y2"], rec["y3"], rec["yd"], rec["out"]
    b1, b2, b3, bd = rec["b1"], rec["b2"], rec["b3"], rec["bd"]
    c1, c2, c3, cd = rec["c1"], rec["c2"], rec["c3"], rec["cd"]
    # bn3 (+ residual ReLU mask from `out`)
    dy3, dg, db = ops.bn_bwd(y3, dout, b3, blk.bn3.weight, out=out)
    grads[blk.bn3.weight], grads[blk.bn3.bias] = dg, db
    # conv3: input operand = relu(bn2(y2)) recomputed in the prologue
    grads[blk.conv3.weight] = ops.conv_wgrad(c3, y2, dy3, pro=(b2, True)).permute(0, 3, 1, 2)
    da2 = ops.conv_dgrad(c3, dy3, ops.weight_transpose(khwc(blk.conv3.weight)))
    del dy3
    dy2, dg, db = ops.bn_bwd(y2, da2, b2, blk.bn2.weight, mask_bnp=b2)
    grads[blk.bn2.weight], grads[blk.bn2.bias] = dg, db
    del da2
    grads[blk.conv2.weight] = ops.conv_wgrad(c2, y1, dy2, pro=(b1, True)).permute(0, 3, 1, 2)
    da1 = ops.conv_dgrad(c2, dy2, ops.weight_transpose(khwc(blk.conv2.weight)))
    del dy2
    dy1, dg, db = ops.bn_bwd(y1, da1, b1, blk.bn1.weight, mask_bnp=b1)
    grads[blk.bn1.weight], grads[blk.bn1.bias] = dg, db
    del da1
    grads[blk.conv1.weight] = ops.conv_wgrad(c1, x, dy1).permute(0, 3, 1, 2)
    if blk.downsample is not None:
        dyd, dg, db = ops.bn_bwd(yd, dout, bd, blk.downsample[1].weight, out=out)
        grads[blk.downsample[1].weight], grads[blk.downsample[1].bias] = dg, db
        grads[blk.downsample[0].weight] = ops.conv_wgrad(cd, x, dyd).permute(0, 3, 1, 2)
        dx = ops.conv_dgrad(cd, dyd, ops.weight_transpose(khwc(blk.downsample[0].weight)))
    else:
        dx = ops.relu_bwd(dout, out)                      # identity branch: dz = dout * (out > 0)
    ops.conv_dgrad(c1, dy1, ops.weight_transpose(khwc(blk.conv1.weight)), out=dx, accumulate=True)
    return dx


class _BodyFn(torch.autograd.Function):
    """x (B,H,W,3) NHWC -> (c1, c2, c3, c4) NHWC.  params are passed so autograd routes their grads."""

    @staticmethod
    def forward(ctx, owner: "BackboneBase", need_grad: bool, x: torch.Tensor, *params: torch.Tensor):
        body: ResNetBody = owner.body
        train = owner.training
        if need_grad and not train:
            raise NotImplementedError("dpft_amd: backward through eval-mode BatchNorm is not implemented")
        recs: List[dict] = [] if need_grad else None
        x = x.contiguous()
        B, H, W, Cin = x.shape
        stem = {}
        if owner.adjustment_layer is not None:
            ca = ops.conv_problem(B, H, W, Cin, 3, 1, 1, 1, 0)
            xa, _ = ops.conv_fwd(ca, x, khwc(owner.adjustment_layer.weight))
            stem.update(ca=ca, x_raw=x)
        else:
            xa = x
        c0 = ops.conv_problem(B, H, W, 3, 64, 7, 7, 2, 3)
        y0, s0 = ops.conv_fwd(c0, xa, khwc(body.conv1.weight), want_stats=train)
        b0 = _bn_forward(body.bn1, s0, c0, train)
        cur = ops.bn_relu_maxpool(y0, b0)
        stem.update(c0=c0, xa=xa, y0=y0, b0=b0)
        outs = []
        for li in range(body.n_layers):
            for blk in getattr(body, f"layer{li + 1}"):
                rec = {} if need_grad else None
                cur = _block_forward(blk, cur, train, rec)
                if need_grad:
                    rec["blk"] = blk
                    rec["layer"] = li
                    recs.append(rec)
            outs.append(cur.detach())       # alias: keeps ctx free of references to its own outputs
        if train:
            nbt = [m.num_batches_tracked for m in body.modules() if isinstance(m, nn.BatchNorm2d)]
            torch._foreach_add_(nbt, 1)
        ctx.owner, ctx.recs, ctx.stem, ctx.params = owner, recs, stem, params
        ctx.x_requires_grad = x.requires_grad
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        owner, recs, stem, params = ctx.owner, ctx.recs, ctx.stem, ctx.params
        body: ResNetBody = owner.body
        grads: Dict[nn.Parameter, torch.Tensor] = {}
        douts = list(douts) + [None] * (4 - len(douts))
        g = None
        last_layer = body.n_layers - 1
        for rec in reversed(recs):
            li = rec["layer"]
            if li != last_layer or g is None:
                # crossing a stage boundary: add the external gradient of that stage's output
                ext = douts[li]
                if g is None:
                    g = ext.contiguous() if ext is not None else torch.zeros_like(rec["out"])
                elif ext is not None:
                    ops.add_(g, ext.contiguous())
                last_layer = li
            g = _block_backward(rec["blk"], rec, g, grads)
            rec.clear()
            sink = owner.grad_sink
            if sink is not None:          # hand this block's gradients to the DP reducer right away
                for p_ in list(grads):
                    if sink(p_, grads[p_]):
                        del grads[p_]
        # stem: maxpool + relu + bn1 + conv1 (+ adjustment conv)
        b0, y0, c0, xa = stem["b0"], stem["y0"], stem["c0"], stem["xa"]
        dz0 = ops.bn_relu_maxpool_bwd(y0, b0, g)
        dy0, dg, db = ops.bn_bwd(y0, dz0, b0, body.bn1.weight)
        grads[body.bn1.weight], grads[body.bn1.bias] = dg, db
        grads[body.conv1.weight] = ops.conv_wgrad(c0, xa, dy0).permute(0, 3, 1, 2)
        dx = None
        if owner.adjustment_layer is not None:
            dxa = ops.conv_dgrad(c0, dy0, ops.weight_transpose(khwc(body.conv1.weight)))
            grads[owner.adjustment_layer.weight] = ops.conv_wgrad(stem["ca"], stem["x_raw"], dxa).permute(0, 3, 1, 2)
        out = [None, None, dx]
        for p in params:
            out.append(grads.get(p))
        return tuple(out)


class BackboneBase(nn.Module):
    def __init__(self, depths, in_channels: int = 3, multi_scale: int = 1, channel_last: bool = True,
                 weights: "OrderedDict[str, Any]" = None, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.multi_scale = multi_scale
        self.channel_last = channel_last
        self.grad_sink = None       # optional callable(param, grad) -> bool, installed by the DP trainer
        # resnet.py:47-52 -- 1x1 conv (no bias) to 3 channels when the input is not RGB
        if in_channels == 3:
            self.adjustment_layer = None
        else:
            self.adjustment_layer = nn.Conv2d(in_channels, 3, kernel_size=(1, 1), stride=1, padding=0, bias=False)
            self.adjustment_layer.weight.data = self.adjustment_layer.weight.data.contiguous(
                memory_format=torch.channels_last)
        self.body = ResNetBody(depths, n_layers=max(1, min(4, multi_scale)))
        if weights:
            self.load_state_dict(weights)

    def forward(self, batch: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        """(B,H,W,C) [channel_last] or (B,C,H,W) -> {'1': layer1, ...} in the input's channel format."""
        if not self.channel_last:
            batch = batch.movedim(1, -1)
        params = [p for p in self.parameters()]
        # grad mode is off inside Function.forward, so the decision is taken here
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        outs = _BodyFn.apply(self, need_grad, batch, *params)
        out = OrderedDict((str(i + 1), o) for i, o in enumerate(outs))
        if not self.channel_last:
            out = OrderedDict((k, v.movedim(-1, 1)) for k, v in out.items())
        return out


class Backbone(BackboneBase):
    def __init__(self, name: str, weights: str = "", norm_layer: str = None, in_channels: int = 3,
                 multi_scale: int = 1, **kwargs):
        if name.lower() not in DEPTHS:
            raise ValueError(f"dpft_amd supports {sorted(DEPTHS)} backbones, got {name!r}")
        if norm_layer not in (None, "BatchNorm2d"):
            raise ValueError(f"dpft_amd backbones use BatchNorm2d, got norm_layer={norm_layer!r}")
        state = None
        if weights:
            # resnet.py:151-165: official weight enums need a download (impossible offline) -> only a
            # state-dict file path is accepted here.
            try:
                state = torch.load(weights, map_location="cpu")
            except (FileNotFoundError, IsADirectoryError) as e:
                raise ValueError(
                    f"backbone weights {weights!r}: torchvision weight enums (e.g. IMAGENET1K_V2) cannot be "
                    "downloaded offline; pass '' (random init) or a state-dict path") from e
        super().__init__(DEPTHS[name.lower()], in_channels, multi_scale, weights=state)

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "Backbone":
        return cls(**config)


def build_resnet(*args, **kwargs):
    return Backbone.from_config(*args, **kwargs)
