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

import ctypes as C
import weakref

from dpft_amd.hip import ops
from dpft_amd.hip.lib import HipLibraryError, ResnetDesc, ResnetTables, lib, ptr, stream, weights_generation

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
# native launch plan (dpft_amd/csrc/resnet_plan.hip): one C-ABI call per forward / per backward stage
# ------------------------------------------------------------------------------------------------
class _Plan:
    """Cached per (input shape): the C plan handle + arena geometry."""

    def __init__(self, owner: "BackboneBase", B: int, H: int, W: int, act16: bool = False):
        body = owner.body
        d = ResnetDesc()
        d.B, d.H, d.W, d.in_channels = B, H, W, owner.in_channels
        d.act16 = self.act16 = int(act16)      # 0 fp32 storage | 1 bf16 activations | 2 bf16 activations and weights
        for i, n in enumerate(owner.depths):
            d.depths[i] = n
        d.n_layers = body.n_layers
        d.eps, d.momentum = body.bn1.eps, body.bn1.momentum
        self.handle = lib.dpft_resnet_plan_create(C.byref(d))
        if not self.handle:
            raise RuntimeError("resnet plan: " + lib.dpft_last_error().decode())
        q = lambda what, idx=0: int(lib.dpft_resnet_plan_query(self.handle, what, idx))
        self.arena_bytes, self.n_conv, self.n_bn = q(0), q(1), q(2)
        self.outs = [(q(3, li), tuple(q(4, li * 4 + k) for k in range(4))) for li in range(body.n_layers)]
        # hipGraph replay (dpft_resnet_plan_set_graph): the C side keys its graphs on the pointer arguments, so this side
        # keeps them still -- one arena for the plan's lifetime, static copies of the input and of the external gradients
        self.graphed = False
        self.lease = None            # weakref to the _ArenaLease of the forward whose saved activations live in self.arena
        self.arena = None
        self.x_static = None
        self.dout_static = {}
        self.tables_cache = None

    def __del__(self):
        try:
            lib.dpft_resnet_plan_destroy(self.handle)
        except Exception:
            pass


class _ArenaLease:
    """Held by the autograd context of the ONE forward whose saved activations live in a plan's persistent arena / static
    input copy (hipGraph replay keeps those addresses still).  While it is alive -- until that forward's backward has run
    or its graph was dropped -- another grad-enabled forward of the same plan must not overwrite them (two batches per
    loss, teacher / student passes, checkpoint recompute): it takes a fresh arena and eager launches instead."""
    __slots__ = ("__weakref__",)


def _ordered_modules(owner: "BackboneBase"):
    """conv / bn modules in the plan's table order (include/dpft_hip.h dpft_resnet_tables); the module tree of a backbone
    is fixed after construction, so the lists are built once."""
    cached = owner.__dict__.get("_ordered")
    if cached is not None:
        return cached
    body = owner.body
    convs, bns = [], []
    if owner.adjustment_layer is not None:
        convs.append(owner.adjustment_layer)
    convs.append(body.conv1)
    bns.append(body.bn1)
    for li in range(body.n_layers):
        for blk in getattr(body, f"layer{li + 1}"):
            convs += [blk.conv1, blk.conv2, blk.conv3]
            bns += [blk.bn1, blk.bn2, blk.bn3]
            if blk.downsample is not None:
                convs.append(blk.downsample[0])
                bns.append(blk.downsample[1])
    owner.__dict__["_ordered"] = (convs, bns)
    return convs, bns


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


class _BodyFn(torch.autograd.Function):
    """x (B,H,W,C) NHWC -> stage outputs NHWC (views of the plan arena).  params are passed so that
    autograd routes their gradients; the computation is the native plan."""

    @staticmethod
    def forward(ctx, owner: "BackboneBase", need_grad: bool, x: torch.Tensor, *params: torch.Tensor):
        train = owner.training
        # plan mode: 0 eval | 1 train | 2 frozen BatchNorm = a gradient through an eval-mode body (running statistics in the
        # forward, no batch-statistics terms in the backward, running buffers untouched) -- what torchvision's
        # FrozenBatchNorm2d / an eval()-ed BatchNorm2d under autograd computes (resnet.py:169-176)
        mode = 1 if train else (2 if need_grad else 0)
        x = x.contiguous()
        if not x.is_cuda or x.dtype != torch.float32:
            raise HipLibraryError("dpft_amd ops need CUDA (ROCm) fp32 tensors; there is no CPU path")
        B, H, W, Cin = x.shape
        plan = owner._plan(B, H, W)
        convs, bns = _ordered_modules(owner)
        assert len(convs) == plan.n_conv and len(bns) == plan.n_bn
        lease = None
        if need_grad and owner.use_plan_graphs(plan) and (plan.lease is None or plan.lease() is None):
            lease = _ArenaLease()
            plan.lease = weakref.ref(lease)
            if plan.arena is None:
                plan.arena = torch.empty(plan.arena_bytes, dtype=torch.uint8, device=x.device)
            if plan.x_static is None or plan.x_static.shape != x.shape:
                plan.x_static = torch.empty_like(x)
            if x.data_ptr() != plan.x_static.data_ptr():
                ops.memops([(plan.x_static, x)])
                x = plan.x_static
            arena = plan.arena
        else:
            arena = torch.empty(plan.arena_bytes, dtype=torch.uint8, device=x.device)
        # inference: the pointer tables only depend on where the parameters live -- rebuilt when a tensor moved
        # (building them costs ~0.4 ms of host time per backbone, in front of the forward's first kernel)
        sig = (weights_generation(), tuple(c.weight.data_ptr() for c in convs),
               tuple(m.running_var.data_ptr() for m in bns), tuple(m.running_mean.data_ptr() for m in bns),
               tuple(m.weight.data_ptr() for m in bns), tuple(m.bias.data_ptr() for m in bns))
        cached = owner.__dict__.get("_infer_tables") if not need_grad else None
        # training with direct-to-bucket gradients: the tables also name the reducer's bucket views -- the same every step for the
        # life of the reducer (its arena): rebuilt when a parameter, a buffer or the arena moved (~1.1 ms of host time per
        # backbone and step otherwise, on a step whose launch work takes the host 15+ ms)
        tsig = None
        if need_grad and owner.grad_direct is not None:
            # (pointers only: the tables alias the parameters, an optimizer step does not invalidate them)
            tsig = (sig[1:], id(owner.grad_direct), owner.grad_direct.arena.data_ptr(), len(params))
            tc = owner.__dict__.get("_train_tables")
            if tc is not None and tc[0] == tsig:
                prev = owner.__dict__.get("_direct_lease")
                if prev is not None and prev() is not None:
                    raise RuntimeError("dpft_amd backbone: a second grad-enabled forward while the previous one's backward "
                                       "is outstanding is not supported with direct-to-bucket gradients (grad_direct)")
                cached = tc
        if cached is not None and not need_grad and cached[0] == sig:
            _, t, keep, weights = cached
            conv_g = bn_g = bn_b = flat = direct = None
        elif cached is not None and need_grad:
            _, t, keep, weights, conv_g, bn_g, bn_b = cached
            flat, direct = None, owner.grad_direct
        else:
            weights = [khwc(c.weight) for c in convs]                       # physical [K][kh][kw][C]
            # gradient buffers: straight into the DP buckets when a reducer is attached, else one flat buffer
            direct = owner.grad_direct if need_grad else None
            if direct is not None:
                # direct-to-bucket gradients are WRITTEN, not added: a second grad-enabled forward before the first one's
                # backward would silently lose one of the two contributions -- refuse loudly (detach the reducer, i.e. use
                # the module without DataParallelTrainer, for multi-forward losses: autograd then accumulates)
                prev = owner.__dict__.get("_direct_lease")
                if prev is not None and prev() is not None:
                    raise RuntimeError("dpft_amd backbone: a second grad-enabled forward while the previous one's backward "
                                       "is outstanding is not supported with direct-to-bucket gradients (grad_direct)")
            conv_g, bn_g, bn_b, flat = [None] * len(convs), [None] * len(bns), [None] * len(bns), None
            if need_grad:
                if direct is not None:
                    conv_g = [direct.grad_buffer(c.weight) for c in convs]
                    bn_g = [direct.grad_buffer(m.weight) for m in bns]
                    bn_b = [direct.grad_buffer(m.bias) for m in bns]
                if direct is None or any(g is None for g in conv_g + bn_g + bn_b):
                    if owner.grad_direct is not None:      # gradients will be ACCUMULATED by autograd this step: the
                        owner.grad_direct.clear(params)     # reducer no longer clears these (set_overwritten)
                    direct = None
                    total = sum(c.weight.numel() for c in convs) + 2 * sum(m.weight.numel() for m in bns)
                    flat = torch.empty(total, dtype=torch.float32, device=x.device)
                    off = 0
                    conv_g, bn_g, bn_b = [], [], []
                    for c in convs:
                        K, Ci, kh, kw = c.weight.shape
                        conv_g.append(flat[off:off + c.weight.numel()].view(K, kh, kw, Ci).permute(0, 3, 1, 2))
                        off += c.weight.numel()
                    for m in bns:
                        n = m.weight.numel()
                        bn_g.append(flat[off:off + n]); bn_b.append(flat[off + n:off + 2 * n])
                        off += 2 * n
            t = ResnetTables()
            keep = [_ptr_array(weights), _ptr_array(conv_g), _ptr_array([m.weight for m in bns]),
                    _ptr_array([m.bias for m in bns]), _ptr_array([m.running_mean for m in bns]),
                    _ptr_array([m.running_var for m in bns]), _ptr_array(bn_g), _ptr_array(bn_b)]
            (t.conv_w, t.conv_dw, t.bn_gamma, t.bn_beta, t.bn_rm, t.bn_rv, t.bn_dgamma, t.bn_dbeta) = \
                [C.cast(k, C.POINTER(C.c_void_p)) for k in keep]
            # cached only if every table entry aliases its parameter: khwc() COPIES a weight that is not physically
            # [K][kh][kw][C] (p.data reassigned, load_state_dict(assign=True)), and a cached copy would go stale
            if all(wk.data_ptr() == c.weight.data_ptr() for wk, c in zip(weights, convs)):
                if not need_grad:
                    owner.__dict__["_infer_tables"] = (sig, t, keep, weights)
                elif tsig is not None and direct is not None:
                    owner.__dict__["_train_tables"] = (tsig, t, keep, weights, conv_g, bn_g, bn_b)
        lib.call("dpft_resnet_forward", plan.handle, ptr(x), C.byref(t), ptr(arena), mode, stream())
        if train:      # num_batches_tracked += 1 of every BatchNorm: one launch per 256 counters
            ptrs = tuple(m.num_batches_tracked.data_ptr() for m in bns)
            cached = owner.__dict__.get("_nbt_ptrs")
            if cached is None or cached[0] != ptrs:
                cached = owner.__dict__["_nbt_ptrs"] = (ptrs, [(C.c_void_p * len(ptrs[i:i + 256]))(*ptrs[i:i + 256])
                                                               for i in range(0, len(ptrs), 256)])
            for arr in cached[1]:
                lib.call("dpft_i64_add_many", len(arr), C.cast(arr, C.c_void_p), 1, stream())
        af = arena.view(torch.float32)
        outs = []
        for off, shape in plan.outs:
            n = shape[0] * shape[1] * shape[2] * shape[3]
            outs.append(af[off:off + n].view(shape))
        if need_grad and lease is not None:
            # the captured backward stages read their output gradients from fixed addresses: offer those buffers to the
            # consumers' backward (ops.grad_sink) so that the gradients are produced in place
            for li, o in enumerate(outs):
                sd = plan.dout_static.get(li)
                if sd is None or sd.shape != o.shape:
                    sd = plan.dout_static[li] = torch.empty_like(o)
                ops.register_grad_sink(o, sd)
        if need_grad:
            ctx.state = dict(owner=owner, plan=plan, x=x, arena=arena, tables=t, keep=keep, weights=weights, mode=mode,
                             convs=convs, bns=bns, conv_g=conv_g, bn_g=bn_g, bn_b=bn_b, flat=flat, direct=direct,
                             params=params, lease=lease)
            if direct is not None:
                ctx.state["direct_lease"] = dl = _ArenaLease()
                owner.__dict__["_direct_lease"] = weakref.ref(dl)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        st = ctx.state
        owner, plan, direct = st["owner"], st["plan"], st["direct"]
        body = owner.body
        side = owner.side_stream
        if side is not None and plan.__dict__.get("_side") != side.cuda_stream:
            lib.call("dpft_resnet_plan_set_side_stream", plan.handle, C.c_void_p(side.cuda_stream))
            plan.__dict__["_side"] = side.cuda_stream
        douts = list(douts) + [None] * (4 - len(douts))
        if st["lease"] is not None:
            af = st["arena"].view(torch.float32)
            for off, shape in plan.outs:
                ops.drop_grad_sink(af[off:off + 1])
        # stage -> parameters whose gradients are complete after that stage's call (the module tree is fixed: built once)
        stage_params = owner.__dict__.get("_stage_params")
        if stage_params is None or any(a is not b for a, b in zip(stage_params[-1], st["params"])) \
                or len(stage_params[-1]) != len(st["params"]):
            stage_params = {li: [] for li in range(body.n_layers)}
            for li in range(body.n_layers):
                for blk in getattr(body, f"layer{li + 1}"):
                    stage_params[li] += list(blk.parameters())
            stage_params[0] += [body.conv1.weight, body.bn1.weight, body.bn1.bias]
            if owner.adjustment_layer is not None:
                stage_params[0].append(owner.adjustment_layer.weight)
            stage_params[-1] = tuple(st["params"])         # (the parameter objects the lists were built from)
            owner.__dict__["_stage_params"] = stage_params
        keep_alive = []
        for li in range(body.n_layers - 1, -1, -1):
            d = douts[li]
            if d is not None:
                d = d.contiguous()
                if plan.graphed and st["lease"] is not None:   # the captured stage reads its gradient from a fixed address
                    sd = plan.dout_static.get(li)
                    if sd is None or sd.shape != d.shape:
                        sd = plan.dout_static[li] = torch.empty_like(d)
                    if sd.data_ptr() != d.data_ptr():      # (not produced in place: see ops.grad_sink)
                        ops.memops([(sd, d)])
                    d = sd
                keep_alive.append(d)
            lib.call("dpft_resnet_backward_stage", plan.handle, li, ptr(st["x"]), C.byref(st["tables"]),
                     ptr(st["arena"]), ptr(d), int(st["mode"] == 2), stream())
            if direct is not None:                       # this stage's gradients are in the DP buckets: release them
                direct.mark_ready_many(stage_params[li])
        grads = {}
        if direct is None:
            for c, g in zip(st["convs"], st["conv_g"]):
                grads[c.weight] = g
            for m, g, b in zip(st["bns"], st["bn_g"], st["bn_b"]):
                grads[m.weight], grads[m.bias] = g, b
        ctx.state = None
        return (None, None, None, *[grads.get(p_) for p_ in st["params"]])


class BackboneBase(nn.Module):
    def __init__(self, depths, in_channels: int = 3, multi_scale: int = 1, channel_last: bool = True,
                 weights: "OrderedDict[str, Any]" = None, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.multi_scale = multi_scale
        self.channel_last = channel_last
        self.grad_direct = None     # optional DP reducer (grad_buffer / mark_ready), installed by the trainer
        self.side_stream = None     # optional torch stream for the plan's weight-gradient GEMMs (DPRT._place_streams)
        self.depths = tuple(depths)
        self._plans = {}
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

    # bf16 activation storage inside the body (mixed precision, BASELINE.json configs[4]): taken when the conv GEMMs run
    # in bf16 mode AND the maps are large (the camera encoder: >= 500k input pixels per batch) -- the small radar maps
    # live on split-K launches whose reduction kernels write fp32
    ACT16_MIN_PIXELS = 500_000

    def _plan(self, B: int, H: int, W: int) -> "_Plan":
        from dpft_amd.hip import ops as _ops
        import os as _os
        # DPFT_ACT16: 0 = keep fp32 storage, 1 = bf16 activations / fp32 weights with operand prologues (round 2),
        # 2 (default) = bf16 activations AND bf16 shadow weights, materialised BatchNorm+ReLU outputs, LDS-DMA bf16 GEMMs
        level = int(_os.environ.get("DPFT_ACT16", "2"))
        act16 = level if (_ops.conv_get_compute() == "bf16" and B * H * W >= self.ACT16_MIN_PIXELS) else 0
        key = (B, H, W, act16, _ops.conv_get_compute(), _ops.conv_get_split())      # tile shapes depend on the compute mode
        p = self._plans.get(key)
        if p is None:
            p = self._plans[key] = _Plan(self, B, H, W, act16)
        return p

    def use_plan_graphs(self, plan: "_Plan") -> bool:
        """hipGraph replay of this plan's launch sequences (DPFT_PLAN_GRAPHS: 0 off | 1 small views only | 2 default).  An
        encoder whose weight-gradient stream is the stream it runs on (the small views under DPRT._place_streams) has a
        single-stream backward: train forward and every backward stage are replayed.  The camera encoder's weight
        gradients need their own hardware queue, which a graph's internal branches do not guarantee: its backward stages
        stay eager launches, its train forward (single-stream, ~350 launches) is replayed (2)."""
        if plan.graphed:
            return True
        import os as _os
        if _os.environ.get("DPFT_PLAN_GRAPHS", "2") == "0" or self.side_stream is None:
            return False
        own = self.side_stream.cuda_stream == torch.cuda.current_stream().cuda_stream and self.side_stream.cuda_stream != 0
        # an encoder with its weight gradients on ANOTHER stream (the camera): its train forward is still a single-stream
        # sequence and is replayed; the C side (run_graphed) keeps such a plan's backward stages eager
        if not own and _os.environ.get("DPFT_PLAN_GRAPHS", "2") != "2":
            return False
        lib.call("dpft_resnet_plan_set_graph", plan.handle, 1)
        plan.graphed = True
        return True

    def overwritten_parameters(self):
        """Parameters whose gradients the native backward plan writes (not adds) into an attached reducer's bucket views:
        the conv weights and BatchNorm affine parameters of the plan's tables (``_ordered_modules``)."""
        convs, bns = _ordered_modules(self)
        return [c.weight for c in convs] + [m.weight for m in bns] + [m.bias for m in bns]

    def __getstate__(self):          # plans hold native handles: rebuild lazily after unpickling / deepcopy
        st = self.__dict__.copy()
        st["_plans"] = {}
        st["grad_direct"] = None
        st["side_stream"] = None
        for k in ("_ordered", "_infer_tables", "_train_tables", "_stage_params", "_plist", "_direct_lease", "_nbt_ptrs"):
            st.pop(k, None)
        return st

    def forward(self, batch: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        """(B,H,W,C) [channel_last] or (B,C,H,W) -> {'1': layer1, ...} in the input's channel format."""
        if not self.channel_last:
            batch = batch.movedim(1, -1)
        params = self.__dict__.get("_plist")
        if params is None:              # the Parameter objects of a backbone are fixed after construction
            params = self.__dict__["_plist"] = list(self.parameters())
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
