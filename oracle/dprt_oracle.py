"""TEST INFRASTRUCTURE -- fp32 CPU restatement of the DPFT hot path on torch primitives.

Every function cites the reference lines (relative to /root/reference) whose
arithmetic it restates.  All functions are *functional*: they take a flat
``state_dict`` (the same key names the reference model produces, SURVEY.md
App. D) so that product weights can be fed to the oracle unchanged.

Parity status: see ``oracle/__init__.py``.  The three third-party cores
(torchvision ResNet/FPN, the MSDA extension, pytorch3d box3d_overlap) are
"parity unpinned" by the reference; everything else is pinned by
``tests/golden`` fixtures produced from the imported reference.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

RESNET_DEPTHS = {
    "resnet50": (3, 4, 6, 3),
    "resnet101": (3, 4, 23, 3),
    "resnet152": (3, 8, 36, 3),
}


# --------------------------------------------------------------------------- #
# Backbone: torchvision ResNet body behind IntermediateLayerGetter
# (src/dprt/models/backbones/resnet.py:47-55, 80-107; architecture = torchvision
# 0.14 resnet.py, third-party, restated from SURVEY.md App. B)
# --------------------------------------------------------------------------- #
def _bn(x, sd, p, train: bool, eps: float = 1e-5):
    """BatchNorm2d(eps=1e-5, momentum=0.1). Train mode: batch statistics
    (biased var for normalisation); running buffers are NOT updated here."""
    if train:
        return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"],
                            training=True, momentum=0.0, eps=eps)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], training=False, eps=eps)


def bottleneck(x, sd, p: str, stride: int, train: bool):
    """torchvision Bottleneck v1.5: stride lives on the 3x3 conv."""
    out = F.conv2d(x, sd[p + ".conv1.weight"])
    out = F.relu(_bn(out, sd, p + ".bn1", train))
    out = F.conv2d(out, sd[p + ".conv2.weight"], stride=stride, padding=1)
    out = F.relu(_bn(out, sd, p + ".bn2", train))
    out = F.conv2d(out, sd[p + ".conv3.weight"])
    out = _bn(out, sd, p + ".bn3", train)
    if (p + ".downsample.0.weight") in sd:
        idn = F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride)
        idn = _bn(idn, sd, p + ".downsample.1", train)
    else:
        idn = x
    return F.relu(out + idn)


def resnet_body(x, sd, p: str, depths: Sequence[int], train: bool,
                multi_scale: int = 4) -> "OrderedDict[str, torch.Tensor]":
    """conv1 7x7/2 -> bn1 -> relu -> maxpool 3x3/2 -> layer1..4; avgpool/fc dropped
    by IntermediateLayerGetter (resnet.py:54-55). x is NCHW."""
    x = F.conv2d(x, sd[p + ".conv1.weight"], stride=2, padding=3)
    x = F.relu(_bn(x, sd, p + ".bn1", train))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    out = OrderedDict()
    for li, nblocks in enumerate(depths):
        for b in range(nblocks):
            stride = 2 if (li > 0 and b == 0) else 1
            x = bottleneck(x, sd, f"{p}.layer{li + 1}.{b}", stride, train)
        if li < multi_scale:
            out[str(li + 1)] = x
    return out


def backbone(x_nhwc, sd, p: str, name: str, train: bool, multi_scale: int = 4):
    """BackboneBase.forward (resnet.py:80-107): NHWC->NCHW view, optional 1x1
    adjustment conv (no bias, resnet.py:47-52), body, NHWC views out."""
    x = x_nhwc.movedim(-1, 1)
    if (p + ".adjustment_layer.weight") in sd:
        x = F.conv2d(x, sd[p + ".adjustment_layer.weight"])
    feats = resnet_body(x, sd, p + ".body", RESNET_DEPTHS[name.lower()], train, multi_scale)
    return OrderedDict((k, v.movedim(1, -1)) for k, v in feats.items())


# --------------------------------------------------------------------------- #
# Neck: torchvision FeaturePyramidNetwork (src/dprt/models/necks/fpn.py:39-43,70-83)
# --------------------------------------------------------------------------- #
def fpn(feats_nhwc: "OrderedDict[str, torch.Tensor]", sd, p: str):
    """inner 1x1 (+bias) -> top-down nearest upsample + add -> 3x3 (+bias, pad 1).
    No norm, no activation, no extra blocks."""
    xs = [v.movedim(-1, 1) for v in feats_nhwc.values()]
    keys = list(feats_nhwc.keys())
    n = len(xs)
    last = F.conv2d(xs[-1], sd[f"{p}.fpn.inner_blocks.{n - 1}.0.weight"],
                    sd[f"{p}.fpn.inner_blocks.{n - 1}.0.bias"])
    outs = [None] * n
    outs[-1] = F.conv2d(last, sd[f"{p}.fpn.layer_blocks.{n - 1}.0.weight"],
                        sd[f"{p}.fpn.layer_blocks.{n - 1}.0.bias"], padding=1)
    for i in range(n - 2, -1, -1):
        lat = F.conv2d(xs[i], sd[f"{p}.fpn.inner_blocks.{i}.0.weight"],
                       sd[f"{p}.fpn.inner_blocks.{i}.0.bias"])
        td = F.interpolate(last, size=lat.shape[-2:], mode="nearest")
        last = lat + td
        outs[i] = F.conv2d(last, sd[f"{p}.fpn.layer_blocks.{i}.0.weight"],
                           sd[f"{p}.fpn.layer_blocks.{i}.0.bias"], padding=1)
    return OrderedDict((k, o.movedim(1, -1)) for k, o in zip(keys, outs))


def nearest_index(dst: int, n_in: int, n_out: int) -> int:
    """F.interpolate(mode='nearest') source index:
    src = min(floor(dst * (in/out)), in-1) with the scale in fp32."""
    scale = torch.tensor(n_in, dtype=torch.float32) / torch.tensor(n_out, dtype=torch.float32)
    return min(int(torch.floor(torch.tensor(dst, dtype=torch.float32) * scale).item()), n_in - 1)


# --------------------------------------------------------------------------- #
# Sinusoidal embedding (src/dprt/models/embeddings/sinusoidal.py:63-110)
# --------------------------------------------------------------------------- #
def sinusoidal_table(H: int, W: int, num_feats: int = 16, temperature: float = 10000.0,
                     normalize: bool = True, scale: float = 2 * math.pi, eps: float = 1e-6,
                     offset: float = 0.0, dtype=torch.float32) -> torch.Tensor:
    """(H, W, num_feats) table = pos_x + pos_y, exactly the two in-place adds of
    sinusoidal.py:107-108 (cumsum starts at 1, :83-84; normalisation :86-90;
    dim_t :92-94; interleaved sin/cos :99-104)."""
    y_embed = torch.arange(1, H + 1, dtype=dtype).view(H, 1).expand(H, W)
    x_embed = torch.arange(1, W + 1, dtype=dtype).view(1, W).expand(H, W)
    if normalize:
        y_embed = (y_embed + offset) / (y_embed[-1:, :] + eps) * scale
        x_embed = (x_embed + offset) / (x_embed[:, -1:] + eps) * scale
    dim_t = torch.arange(num_feats, dtype=dtype)
    dim_t = temperature ** (2 * (dim_t // 2) / num_feats)
    pos_x = x_embed[:, :, None] / dim_t
    pos_y = y_embed[:, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, 0::2].sin(), pos_x[:, :, 1::2].cos()), dim=3).view(H, W, -1)
    pos_y = torch.stack((pos_y[:, :, 0::2].sin(), pos_y[:, :, 1::2].cos()), dim=3).view(H, W, -1)
    return pos_x, pos_y


def sinusoidal_embedding(x_nhwc: torch.Tensor, **kw) -> torch.Tensor:
    """x += pos_x; x += pos_y (two separate fp32 adds, same order as the reference)."""
    _, H, W, C = x_nhwc.shape
    pos_x, pos_y = sinusoidal_table(H, W, num_feats=kw.get("num_feats", C),
                                    temperature=kw.get("temperature", 10000.0),
                                    normalize=kw.get("normalize", False),
                                    dtype=x_nhwc.dtype)
    out = x_nhwc + pos_x
    out = out + pos_y
    return out


# --------------------------------------------------------------------------- #
# Querent (src/dprt/models/queries/data_agnostic.py:126-172) + spher2cart
# (src/dprt/models/utils/transformations.py:212-255)
# --------------------------------------------------------------------------- #
def querent(B: int, resolution, minimum, maximum, dtype=torch.float32) -> torch.Tensor:
    qs = [torch.linspace(0.0, 1.0, r, dtype=dtype) for r in resolution]
    qs = [torch.mul(1, q) for q in qs]                                  # 'linear' distribution
    scaled = []
    for q, mi, ma in zip(qs, minimum, maximum):                          # _min_max_scaling :117-124
        den = torch.max(q) - torch.min(q)
        if torch.isclose(den, torch.zeros_like(den)):
            den = 1.0
        scaled.append((q - torch.min(q)) / den * (ma - mi) + mi)
    grid = torch.meshgrid(*scaled, indexing="ij")
    pts = torch.stack([torch.flatten(g) for g in grid], dim=-1)         # (N, 3) = (r, phi, roh)
    pts = pts.repeat((B,) + (1,) * pts.dim())
    r, phi, roh = pts.split(1, -1)
    x = r * torch.cos(torch.deg2rad(phi)) * torch.cos(torch.deg2rad(roh))
    y = r * torch.sin(torch.deg2rad(phi)) * torch.cos(torch.deg2rad(roh))
    z = r * torch.sin(torch.deg2rad(roh))
    return torch.cat((x, y, z), -1)


# --------------------------------------------------------------------------- #
# Reference points (src/dprt/models/fusers/mpfusion.py:617-696) + cart2spher
# (src/dprt/models/utils/transformations.py:71-120)
# --------------------------------------------------------------------------- #
def cart2spher_deg(x, y, z):
    r = torch.linalg.norm(torch.dstack((x, y, z)), dim=-1).reshape_as(x)
    phi = torch.arctan2(y, x)
    c = torch.zeros_like(z)
    mask = r != 0
    c = torch.where(mask, z / torch.where(mask, r, torch.ones_like(r)), c)
    roh = torch.arcsin(c)
    return r, torch.rad2deg(phi), torch.rad2deg(roh)


def reference_points(center, transformation, projection, shape):
    """(B,N,3),(B,4,4),(B,4|3,4),(B,2)[H,W] -> (B,N,2) ordered (u=x/W, v=y/H), clipped to [0,1]."""
    q = center[..., :3]
    if bool(transformation.any()):
        hom = torch.dstack((q, torch.ones_like(q[..., 0])))
        p = torch.einsum("bij,bkj->bki", transformation, hom)
        r, phi, roh = cart2spher_deg(p[..., 0], p[..., 1], p[..., 2])
        q = torch.dstack((r, phi, roh))
    hom = torch.dstack((q[..., :3], torch.ones_like(q[..., 0])))
    p = torch.einsum("bij,bkj->bki", projection, hom)
    w = p[..., 2]
    mask = w != 0
    safe = torch.where(mask, w, torch.ones_like(w))
    u = torch.where(mask, p[..., 0] / safe, p[..., 0])
    v = torch.where(mask, p[..., 1] / safe, p[..., 1])
    u = (u - 0) / (shape[:, 1].unsqueeze(1) - 0) * (1 - 0) + 0
    v = (v - 0) / (shape[:, 0].unsqueeze(1) - 0) * (1 - 0) + 0
    return torch.clip(torch.dstack((u, v)), min=0.0, max=1.0)


# --------------------------------------------------------------------------- #
# MSDA core (third-party extension called at src/dprt/models/layers/ms_deform_attn.py:32-39)
# --------------------------------------------------------------------------- #
def msda_core(value, spatial_shapes, sampling_locations, attention_weights):
    """grid_sample(bilinear, zeros, align_corners=False) formulation of the upstream
    ms_deformable_im2col kernel.  value (N,S,M,D); spatial_shapes list of
    (H,W); loc (N,Lq,M,L,P,2) in [0,1] (x,y); attn (N,Lq,M,L,P) -> (N,Lq,M*D)."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    sizes = [int(h) * int(w) for h, w in spatial_shapes]
    value_list = value.split(sizes, dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lid, (H, W) in enumerate(spatial_shapes):
        v = value_list[lid].flatten(2).transpose(1, 2).reshape(N * M, D, int(H), int(W))
        g = grids[:, :, :, lid].transpose(1, 2).flatten(0, 1)           # (N*M, Lq, P, 2)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros",
                                     align_corners=False))              # (N*M, D, Lq, P)
    attn = attention_weights.transpose(1, 2).reshape(N * M, 1, Lq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * attn).sum(-1).view(N, M * D, Lq)
    return out.transpose(1, 2).contiguous()


def msda_core_scalar(value, spatial_shapes, level_start_index, loc, attn):
    """Pure-python restatement of the upstream im2col thread body (SURVEY App. C pseudo-code);
    small cases only.  Used to cross-check ``msda_core``."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = torch.zeros(N, Lq, M * D, dtype=value.dtype)
    for b in range(N):
        for q in range(Lq):
            for m in range(M):
                for c in range(D):
                    col = 0.0
                    for l in range(L):
                        H, W = int(spatial_shapes[l][0]), int(spatial_shapes[l][1])
                        base = int(level_start_index[l])
                        for p in range(P):
                            lw = float(loc[b, q, m, l, p, 0]); lh = float(loc[b, q, m, l, p, 1])
                            a = float(attn[b, q, m, l, p])
                            h_im = lh * H - 0.5; w_im = lw * W - 0.5
                            if not (h_im > -1 and w_im > -1 and h_im < H and w_im < W):
                                continue
                            h_lo = math.floor(h_im); w_lo = math.floor(w_im)
                            h_hi = h_lo + 1; w_hi = w_lo + 1
                            lh_ = h_im - h_lo; lw_ = w_im - w_lo; hh = 1 - lh_; hw = 1 - lw_

                            def at(y, x):
                                if 0 <= y <= H - 1 and 0 <= x <= W - 1:
                                    return float(value[b, base + y * W + x, m, c])
                                return 0.0
                            col += a * (hh * hw * at(h_lo, w_lo) + hh * lw_ * at(h_lo, w_hi)
                                        + lh_ * hw * at(h_hi, w_lo) + lh_ * lw_ * at(h_hi, w_hi))
                    out[b, q, m * D + c] = col
    return out


# --------------------------------------------------------------------------- #
# MSDeformAttn.forward (src/dprt/models/layers/ms_deform_attn.py:138-217)
# --------------------------------------------------------------------------- #
def ms_deform_attn(query, ref_points_2d, levels_nhwc: List[torch.Tensor], sd, p: str,
                   n_heads: int, n_points: int):
    """query (B,Q,C) [already with pos]; ref (B,Q,2); levels: list of (B,H,W,C)."""
    B, Q, C = query.shape
    L = len(levels_nhwc)
    shapes = [(l.shape[1], l.shape[2]) for l in levels_nhwc]
    input_flatten = torch.cat([l.flatten(1, 2) for l in levels_nhwc], dim=1)   # mpfusion.py:179
    value = F.linear(input_flatten, sd[p + ".value_proj.weight"], sd[p + ".value_proj.bias"])
    value = value.view(B, -1, n_heads, C // n_heads)
    off = F.linear(query, sd[p + ".sampling_offsets.weight"], sd[p + ".sampling_offsets.bias"])
    off = off.view(B, Q, n_heads, L, n_points, 2)
    aw = F.linear(query, sd[p + ".attention_weights.weight"], sd[p + ".attention_weights.bias"])
    aw = F.softmax(aw.view(B, Q, n_heads, L * n_points), -1).view(B, Q, n_heads, L, n_points)
    normalizer = torch.tensor([[w, h] for h, w in shapes], dtype=query.dtype)     # (W_l, H_l) :186-188
    ref = ref_points_2d.unsqueeze(2).repeat(1, 1, L, 1)                          # mpfusion.py:190
    loc = ref[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    out = msda_core(value, shapes, loc, aw)
    return F.linear(out, sd[p + ".output_proj.weight"], sd[p + ".output_proj.bias"])


# --------------------------------------------------------------------------- #
# MLFusion / MPFusion / IMPFusion (src/dprt/models/fusers/mpfusion.py)
# --------------------------------------------------------------------------- #
def mha(q_in, k_in, v_in, sd, p: str, n_heads: int, att_scale=None):
    """nn.MultiheadAttention(batch_first) explicit form: rows of
    in_proj_weight packed [q;k;v]; softmax(q k^T / sqrt(hd)) v; out_proj.
    ``att_scale`` (B,heads,Q,Q): explicit dropout on the attention probabilities (0 or 1/(1-p) per entry; torch applies
    ``dropout(softmax(...))`` there, functional.py multi_head_attention_forward) for tests that replay given masks."""
    C = q_in.shape[-1]
    Wi, bi = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(q_in, Wi[:C], bi[:C]); k = F.linear(k_in, Wi[C:2 * C], bi[C:2 * C])
    v = F.linear(v_in, Wi[2 * C:], bi[2 * C:])
    B, Q, _ = q.shape
    hd = C // n_heads
    q = q.view(B, Q, n_heads, hd).transpose(1, 2); k = k.view(B, Q, n_heads, hd).transpose(1, 2)
    v = v.view(B, Q, n_heads, hd).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    if att_scale is not None:
        att = att * att_scale
    out = (att @ v).transpose(1, 2).reshape(B, Q, C)
    return F.linear(out, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def mlfusion(query, levels, ref, query_pos, sd, p: str, n_heads: int, n_points: int,
             activation: str = "Mish"):
    """MLFusion.forward (mpfusion.py:231-263), dropout = identity (eval / p=0)."""
    qk = query + query_pos
    out = query + mha(qk, qk, query, sd, p + ".self_attn", n_heads)                # :122-148
    out = _ln(out, sd, p + ".norm1")
    ca = ms_deform_attn(out + query_pos, ref, levels, sd, p + ".ms_deform_attn", n_heads, n_points)
    out = _ln(out + ca, sd, p + ".norm2")                                           # :150-208
    act = getattr(F, activation.lower())
    ff = F.linear(act(F.linear(out, sd[p + ".ffn1.weight"], sd[p + ".ffn1.bias"])),
                  sd[p + ".ffn2.weight"], sd[p + ".ffn2.bias"])
    return _ln(out + ff, sd, p + ".norm3")                                          # :210-229


def mpfusion(query, views: List[List[torch.Tensor]], refs, query_pos, sd, p: str,
             n_heads, n_points, activation="Mish"):
    """MPFusion.forward + 'linear' reduce (mpfusion.py:472-514, 434-438): stack views on a
    trailing axis and view(B,N,C*V) => channel-major / view-minor interleave."""
    outs = [mlfusion(query, lv, rf, query_pos, sd, f"{p}.ml_fusion_layers.ms_deform_attn{v}",
                     n_heads[v], n_points[v], activation)
            for v, (lv, rf) in enumerate(zip(views, refs))]
    queries = torch.stack(outs, dim=-1)
    B, N = query.shape[:2]
    return F.linear(queries.reshape(B, N, -1), sd[p + ".reduction_layer.weight"])


def detection_head(x, ref_center, sd, p: str):
    """LinearDetectionHead.forward (src/dprt/models/heads/detection.py:252-275): four bias-free
    MLPs (Linear,ReLU,Dropout)x2 + Linear; Identity/ReLU/Tanh/Identity; center += ref."""
    def branch(name):
        h = F.relu(F.linear(x, sd[f"{p}.layers.{name}_head.0.weight"]))
        h = F.relu(F.linear(h, sd[f"{p}.layers.{name}_head.3.weight"]))
        return F.linear(h, sd[f"{p}.layers.{name}_head.6.weight"])
    out = OrderedDict()
    out["center"] = branch("center") + ref_center[..., :3]
    out["size"] = F.relu(branch("size"))
    out["angle"] = torch.tanh(branch("angle"))
    out["class"] = branch("class")
    return out


def impfusion(views, shapes, projections, center0, sd, p: str, cfg: dict):
    """IMPFusion.forward (mpfusion.py:698-745)."""
    B = center0.shape[0]
    query = sd[p + ".query"].unsqueeze(0).repeat(B, 1, 1)
    query_pos = sd[p + ".query_embedding.weight"].unsqueeze(0).repeat(B, 1, 1)
    out = OrderedDict(center=center0)
    for it in range(cfg["i_iter"]):
        refs = [reference_points(out["center"][..., :3], t, pr, s)
                for (t, pr), s in zip(projections, shapes)]
        query = mpfusion(query, views, refs, query_pos, sd, f"{p}.mpfusion.fusion{it}",
                         cfg["n_heads"], cfg["n_points"], cfg.get("activation", "ReLU"))
        out = detection_head(query, out["center"], sd, f"{p}.heads.{it}")
    return out


# --------------------------------------------------------------------------- #
# DPRT.forward (src/dprt/models/dprt.py:200-244)
# --------------------------------------------------------------------------- #
def dprt_forward(sd: Dict[str, torch.Tensor], config: dict, batch: Dict[str, torch.Tensor],
                 train: bool = False, return_features: bool = False):
    model = config["model"]
    inputs = model["inputs"]
    feats = {}
    for name in inputs:
        bb = model["backbones"][name]
        f = backbone(batch[name], sd, f"backbones.{name}", bb["name"], train,
                     bb.get("multi_scale", 1))
        if model["skiplinks"].get(name, False):                              # dprt.py:164-179
            f["0"] = batch[name]
            f.move_to_end("0", last=False)
        f = fpn(f, sd, f"necks.{name}")
        emb = model["embeddings"][name]
        f = OrderedDict((k, sinusoidal_embedding(v, **emb)) for k, v in f.items())
        feats[name] = f
    q = model["querent"]
    first = batch[list(batch.keys())[0]]                                     # data_agnostic.py:92-99
    center0 = querent(first.shape[0], q["resolution"], q["minimum"], q["maximum"], first.dtype)
    out = impfusion([list(feats[n].values()) for n in inputs],
                    [batch[f"{n}_shape"][:, :2] for n in inputs],
                    [(batch[f"label_to_{n}_t"], batch[f"label_to_{n}_p"]) for n in inputs],
                    center0, sd, "fuser", model["fuser"])
    if return_features:
        return out, feats
    return out


# --------------------------------------------------------------------------- #
# Loss path (NEXT-1 row of SURVEY 8f; needed for a faithful training step)
# --------------------------------------------------------------------------- #
def focal_loss(inputs, targets, alpha: float = 0.75, gamma: float = 2.0):
    """src/dprt/training/loss.py:17-60 incl. the raw-logit p_t quirk (:44)."""
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = inputs * targets + (1 - inputs) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    alpha_t = alpha * targets + (1 - alpha) * (1 - targets)
    return alpha_t * loss


def box_corners(center, size, angle):
    """src/dprt/utils/bbox.py:4-74 -> (..., 8, 3)."""
    sx = torch.tensor([-1, 1, 1, -1, -1, 1, 1, -1], dtype=center.dtype)
    sy = torch.tensor([-1, -1, 1, 1, -1, -1, 1, 1], dtype=center.dtype)
    sz = torch.tensor([-1, -1, -1, -1, 1, 1, 1, 1], dtype=center.dtype)
    xc = (size[..., 0] / 2)[..., None] * sx
    yc = (size[..., 1] / 2)[..., None] * sy
    zc = (size[..., 2] / 2)[..., None] * sz
    c, s = torch.cos(angle)[..., None], torch.sin(angle)[..., None]
    x = c * xc - s * yc + center[..., None, 0]
    y = s * xc + c * yc + center[..., None, 1]
    z = zc + center[..., None, 2]
    return torch.stack((x, y, z), dim=-1)


def _poly_clip_area(p: List[Tuple[float, float]], q: List[Tuple[float, float]]) -> float:
    """Sutherland-Hodgman clip of convex CCW polygon p by convex CCW polygon q; float64."""
    def area(poly):
        a = 0.0
        for i in range(len(poly)):
            x1, y1 = poly[i]; x2, y2 = poly[(i + 1) % len(poly)]
            a += x1 * y2 - x2 * y1
        return a / 2
    if area(p) < 0:
        p = p[::-1]
    if area(q) < 0:
        q = q[::-1]
    out = p
    for i in range(len(q)):
        ax, ay = q[i]; bx, by = q[(i + 1) % len(q)]
        inp, out = out, []
        if not inp:
            break
        for j in range(len(inp)):
            cx, cy = inp[j]; dx, dy = inp[(j + 1) % len(inp)]
            sc = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)
            sd_ = (bx - ax) * (dy - ay) - (by - ay) * (dx - ax)
            if sc >= 0:
                out.append((cx, cy))
            if (sc >= 0) != (sd_ >= 0):
                t = sc / (sc - sd_)
                out.append((cx + t * (dx - cx), cy + t * (dy - cy)))
    return abs(area(out)) if len(out) >= 3 else 0.0


def giou3d_yaw(c1, s1, a1, c2, s2, a2):
    """GIoU3D for yaw-only boxes, float64 exact geometry; mirrors src/dprt/utils/iou.py:121-210
    with pytorch3d.box3d_overlap (third-party, unpinned) replaced by BEV polygon clip x z-overlap.
    Enclosing box is axis-aligned over all 16 corners (src/dprt/utils/bbox.py:77-134).
    Inputs (N,3),(N,3),(N,),(M,3),(M,3),(M,) -> (N,M).  Degenerate (zero-area face) boxes give
    GIoU with iou=vol=0 like the reference's validity mask (iou.py:39-69,162-175)."""
    N, M = c1.shape[0], c2.shape[0]
    k1 = box_corners(c1.double(), s1.double(), a1.double())
    k2 = box_corners(c2.double(), s2.double(), a2.double())
    out = torch.zeros(N, M, dtype=torch.float64)
    eps = 1e-4

    def valid(s):
        # _check_nonzero (iou.py:39-69): every face-triangle area (= half a face) > eps
        l, w, h = [float(t) for t in s]
        return min(l * w, l * h, w * h) / 2 > eps

    for i in range(N):
        for j in range(M):
            A, Bx = k1[i], k2[j]
            if not (valid(s1[i]) and valid(s2[j])):
                # iou = vol = uni = 0, evol keeps its -1 initialiser (iou.py:159,185-208)
                out[i, j] = 0.0 - ((-1.0) - 0.0) / (-1.0)
                continue
            v1 = float(s1[i].double().prod()); v2 = float(s2[j].double().prod())
            allc = torch.cat((A, Bx), 0)
            evol = float((allc.max(0).values - allc.min(0).values).prod())
            pa = [(float(A[t, 0]), float(A[t, 1])) for t in range(4)]
            pb = [(float(Bx[t, 0]), float(Bx[t, 1])) for t in range(4)]
            inter_a = _poly_clip_area(pa, pb)
            zlo = max(float(A[0, 2]), float(Bx[0, 2])); zhi = min(float(A[4, 2]), float(Bx[4, 2]))
            vol = inter_a * max(0.0, zhi - zlo)
            iou = vol / (v1 + v2 - vol) if vol > 0 else 0.0
            uni = vol / iou if iou != 0 else 0.0                     # iou.py:187-188
            out[i, j] = (iou - (evol - uni) / evol) if evol != 0 else 0.0
    return out


def hungarian(out: Dict[str, torch.Tensor], tgt: Dict[str, torch.Tensor], w: Dict[str, float],
              giou_weight: float = 1.0):
    """HungarianAnassigner.forward for one sample (src/dprt/training/assigner.py:58-143).
    out tensors (N,.), tgt tensors (M,.). Returns (index_i, index_j) int64."""
    from scipy.optimize import linear_sum_assignment
    gt_ids = torch.argmax(tgt["gt_class"], dim=-1)
    cost_class = -out["class"][:, gt_ids]
    cost_center = torch.cdist(out["center"], tgt["gt_center"], p=1)
    cost_size = torch.cdist(out["size"], tgt["gt_size"], p=1)
    cost_angle = torch.cdist(out["angle"], tgt["gt_angle"], p=1)
    oa = torch.atan2(out["angle"][..., 0], out["angle"][..., 1])
    ga = torch.atan2(tgt["gt_angle"][..., 0], tgt["gt_angle"][..., 1])
    giou = giou3d_yaw(out["center"], out["size"], oa, tgt["gt_center"], tgt["gt_size"], ga)
    C = (w["total_class"] * cost_class + w["center"] * cost_center + w["size"] * cost_size
         + w["angle"] * cost_angle + giou_weight * (-giou.to(cost_class.dtype)))
    i, j = linear_sum_assignment(C.detach().cpu().numpy())
    return torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64), C


def set_criterion(out: Dict[str, torch.Tensor], tgt: Dict[str, torch.Tensor], i, j):
    """SetCriterion.forward for one sample with a leading batch dim of 1
    (src/dprt/training/loss.py:234-373)."""
    N = out["class"].shape[0]
    M = j.numel()
    C = out["class"].shape[1]
    one_hot = torch.zeros(N, C, dtype=out["class"].dtype)
    one_hot[:, 0] = 1.0
    one_hot[i] = tgt["gt_class"][torch.arange(M)]            # scatter_ with src=targets (:305-306)
    total = (focal_loss(out["class"], one_hot).mean(0).sum() / M) * N
    obj = (focal_loss(out["class"][i], tgt["gt_class"][j]).mean(0).sum() / M) * N
    losses = OrderedDict(total_class=total, object_class=obj)
    for k in ("center", "size", "angle"):
        losses[k] = F.l1_loss(out[k][i], tgt["gt_" + k][j], reduction="mean")
    return losses


def loss_forward(out: Dict[str, torch.Tensor], targets: List[Dict[str, torch.Tensor]],
                 weights: Dict[str, float]):
    """Loss.forward (src/dprt/training/loss.py:486-564): per-sample assign + criterion, weight,
    batch mean, sum."""
    per = []
    for b, tgt in enumerate(targets):
        o = {k: v[b] for k, v in out.items()}
        if not all(t.numel() for t in tgt.values()):
            per.append({k: torch.zeros((), dtype=o["class"].dtype) for k in weights})
            continue
        with torch.no_grad():
            i, j, _ = hungarian(o, tgt, weights)
        losses = set_criterion(o, tgt, i, j)
        per.append({k: losses[k] * weights[k] for k in weights})
    batch_losses = {k: torch.stack([p[k] for p in per]).mean() for k in weights}
    total = torch.stack(tuple(batch_losses.values())).sum(dim=-1)
    return total, batch_losses
