"""TEST INFRASTRUCTURE (build container only) -- make the reference's Python importable.

/root/reference is pure Python but its package __init__s import third-party modules
that are not installed here (torchvision, MultiScaleDeformableAttention, pytorch3d,
tensorboard, deepspeed).  This module registers *namespace-only* stand-ins in
``sys.modules`` (no arithmetic, except the MSDA entry points which are bound to the
oracle's own grid_sample core -- the reference ships no CPU core) and
puts /root/reference/src on ``sys.path``.  Nothing is copied; nothing is written there.

Only ``oracle/gen_golden.py`` and the ``-m "not gpu"`` reference cross-check tests use it,
and they skip when /root/reference is absent (the GPU box).
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"


def available() -> bool:
    return os.path.isdir(REFERENCE_SRC)


STUBS = []          # names of the namespace-only stub modules install() put into sys.modules


def _mod(name: str, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    STUBS.append(name)
    return m


class stubs_hidden:
    """Context manager: take the stub modules out of ``sys.modules`` while an independent package (``transformers``)
    is imported and used -- it probes ``torchvision`` / ``deepspeed`` with ``importlib.util.find_spec`` and chokes on
    spec-less stubs -- and put them back afterwards.  Makes the second-opinion tests independent of test order."""

    def __enter__(self):
        self.saved = {k: sys.modules.pop(k) for k in STUBS if k in sys.modules}
        return self

    def __exit__(self, *exc):
        sys.modules.update(self.saved)
        return False


class _Placeholder:
    def __init__(self, *a, **k):
        raise RuntimeError("third-party placeholder: not available in the oracle harness")


def install():
    """Idempotent.  Returns the imported ``dprt`` package."""
    if "dprt" in sys.modules:
        return sys.modules["dprt"]
    if not available():
        raise RuntimeError("/root/reference is not present")
    sys.dont_write_bytecode = True
    import torch
    from oracle import dprt_oracle as O

    tv = _mod("torchvision")
    tv.models = _mod("torchvision.models")
    tv.models._utils = _mod("torchvision.models._utils", IntermediateLayerGetter=_Placeholder)
    for sub, cls in (("regnet", "RegNet"), ("convnext", "ConvNeXt"),
                     ("swin_transformer", "SwinTransformer")):
        setattr(tv.models, sub, _mod(f"torchvision.models.{sub}", **{cls: _Placeholder}))
    tv.ops = _mod("torchvision.ops", FeaturePyramidNetwork=_Placeholder)
    tv.io = _mod("torchvision.io", read_image=_Placeholder)
    tv.transforms = _mod("torchvision.transforms")
    tv.transforms.functional = _mod("torchvision.transforms.functional", resize=_Placeholder)

    def _shapes(spatial_shapes):
        return [(int(h), int(w)) for h, w in spatial_shapes.tolist()]

    def ms_deform_attn_forward(value, spatial_shapes, level_start_index, loc, attn, im2col_step):
        return O.msda_core(value, _shapes(spatial_shapes), loc, attn)

    def ms_deform_attn_backward(value, spatial_shapes, level_start_index, loc, attn, grad_out,
                                im2col_step):
        with torch.enable_grad():
            v = value.detach().requires_grad_(True)
            l = loc.detach().requires_grad_(True)
            a = attn.detach().requires_grad_(True)
            out = O.msda_core(v, _shapes(spatial_shapes), l, a)
            return torch.autograd.grad(out, (v, l, a), grad_out)

    _mod("MultiScaleDeformableAttention", ms_deform_attn_forward=ms_deform_attn_forward,
         ms_deform_attn_backward=ms_deform_attn_backward)
    # dataset-preparation imports of dprt.datasets.kradar.processor (only its numpy reductions are exercised)
    _mod("cv2")
    _mod("pypcd", pypcd=_mod("pypcd.pypcd"))
    p3 = _mod("pytorch3d")
    p3.ops = _mod("pytorch3d.ops", box3d_overlap=_Placeholder)
    _mod("torch.utils.tensorboard", SummaryWriter=_Placeholder)
    ds = _mod("deepspeed")
    ds.profiling = _mod("deepspeed.profiling")
    ds.profiling.flops_profiler = _mod("deepspeed.profiling.flops_profiler",
                                       get_model_profile=_Placeholder)
    ds.accelerator = _mod("deepspeed.accelerator", get_accelerator=_Placeholder)
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import dprt  # noqa: F401
    return sys.modules["dprt"]
