"""Reading the REFERENCE's whole-module checkpoints (VERDICT r1 missing #1).

``dprt.train`` saves ``torch.save(model, "<timestamp>_checkpoint_<epoch>.pt")`` (src/dprt/training/trainer.py:256-258) and
``dprt.models.load`` simply unpickles it (src/dprt/models/__init__.py:15-18).  Such a file names classes of ``dprt.*``,
``torchvision.*`` (ResNet / Bottleneck / IntermediateLayerGetter / FeaturePyramidNetwork / Conv2dNormActivation) and the
MSDA extension's autograd function -- none of which exist next to this package.  Nothing of those classes' CODE is
needed: a pickled ``nn.Module`` is its ``__dict__`` (``_parameters`` / ``_buffers`` / ``_modules`` + the constructor
arguments the reference keeps as attributes).  So:

  1. unpickle with a ``find_class`` that maps every class it cannot (or must not) import onto a featureless stand-in
     (``nn.Module`` subclass for module classes, inert object otherwise);
  2. read the tensors (``state_dict()`` of the stand-in tree -- parameter names are identical to this package's,
     tests/test_host.py) and the hyper-parameters (attributes / tensor shapes) -> a config in the reference's schema;
  3. ``build("dprt", config)`` and ``load_state_dict``.
"""
from __future__ import annotations

import pickle
from typing import Any, Dict, Tuple

import torch
from torch import nn

_FOREIGN = ("dprt", "torchvision", "MultiScaleDeformableAttention", "pytorch3d", "deepspeed")


class ForeignModule(nn.Module):
    """Stand-in for a pickled module class that is not importable here: keeps the pickled ``__dict__`` only."""
    _foreign_path = ""

    def forward(self, *a, **k):                                          # pragma: no cover
        raise RuntimeError(f"{self._foreign_path}: stand-in of a foreign checkpoint class, not executable")


class ForeignObject:
    """Stand-in for a non-module foreign class / function (callable placeholders, enums, autograd functions)."""
    _foreign_path = ""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):                                         # pragma: no cover
        raise RuntimeError(f"{self._foreign_path}: stand-in of a foreign checkpoint object, not executable")

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"state": state})


_MODULE_HINTS = ("models.", "ops.", "nn.", "layers.", "backbones.", "necks.", "heads.", "fusers.", "embeddings.",
                 "queries.")
_cache: Dict[Tuple[str, str], type] = {}


def _stand_in(module: str, name: str) -> type:
    key = (module, name)
    if key not in _cache:
        # module classes: everything under the model packages that looks like a class (CamelCase); the rest is inert
        is_module_cls = name[:1].isupper() and any(h in module + "." for h in _MODULE_HINTS) and not name.endswith("Function")
        base = ForeignModule if is_module_cls else ForeignObject
        _cache[key] = type(name, (base,), {"_foreign_path": f"{module}.{name}", "__module__": __name__})
    return _cache[key]


class _Unpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str):
        root = module.split(".")[0]
        if root in _FOREIGN:
            return _stand_in(module, name)
        # everything else (torch, collections, this package, builtins, ...) must resolve: a renamed or missing class of
        # dpft_amd / torch is an error of the checkpoint, not something an inert stand-in may paper over.  super() also
        # applies pickle's py2 -> py3 name fixes (__builtin__ ...).
        return super().find_class(module, name)


class _PickleModule:
    """``pickle_module`` for torch.load: torch's (de)serialisation of storages stays intact, only class lookup changes."""
    __name__ = "dpft_amd.models.checkpoint"
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL


def read_foreign(path: str):
    """The unpickled object of ``path`` with foreign classes replaced by stand-ins (map_location = cpu)."""
    return torch.load(path, map_location="cpu", pickle_module=_PickleModule, weights_only=False)


def _attr(obj, name, default=None):
    return obj.__dict__.get(name, default) if hasattr(obj, "__dict__") else default


def _child(mod: nn.Module, name: str):
    return mod._modules.get(name) if isinstance(mod, nn.Module) else None


_TRANSFORMATIONS = {"Spher2Cart": "spher2cart", "Cart2Spher": "cart2spher", "Polar2Cart": "polar2cart",
                    "Cart2Polar": "cart2polar"}


def _hyper(obj, name, default, what):
    """Constructor argument the reference keeps as an attribute of ``obj``; a module without it is not one this loader
    understands (the value is NOT a tensor, so a silently assumed default would compute something different)."""
    if obj is None or not hasattr(obj, "__dict__") or name not in obj.__dict__:
        raise ValueError(f"checkpoint: {what} has no attribute {name!r}; cannot infer it (default would be {default!r})")
    return obj.__dict__[name]


def infer_config(model: nn.Module) -> Dict[str, Any]:
    """Config (reference schema, sections computing / model) of a DPRT module tree -- from the constructor arguments the
    reference's modules keep as attributes and, for the third-party parts, from tensor shapes.  Every hyper-parameter that
    is not visible in a tensor shape (temperature / scale / offset of the embedding, the querent's distribution and
    transformation, head dropout / bias, the fuser's ffn_layer, the backbone's multi_scale and norm layer) is READ from
    the pickled modules; one that cannot be read or mapped raises instead of falling back to a default."""
    sd = model.state_dict()
    inputs = list(_attr(model, "inputs") or [])
    if not inputs:
        raise ValueError("checkpoint: the pickled object has no 'inputs' attribute -- not a DPRT module")
    backbones, necks, embeddings = {}, {}, {}
    for v in inputs:
        p = f"backbones.{v}.body."
        if any(k.startswith(p) for k in sd):
            bmod = _child(_child(model, "backbones"), v)
            layers = [i for i in (1, 2, 3, 4) if any(k.startswith(f"{p}layer{i}.") for k in sd)]
            blocks = [len({k.split(".")[4] for k in sd if k.startswith(f"{p}layer{i}.")}) for i in layers]
            full = {(3, 4, 6, 3): "ResNet50", (3, 4, 23, 3): "ResNet101", (3, 8, 36, 3): "ResNet152"}
            name = next((n for d, n in full.items() if tuple(blocks) == d[:len(blocks)]), None) if blocks else None
            if len(blocks) < 3 and name is not None:      # (3,) / (3, 4) prefixes are ambiguous between the depths
                name = None
            if name is None or f"{p}layer1.0.conv3.weight" not in sd:
                raise ValueError(f"checkpoint: backbone of {v!r} has block counts {blocks}; dpft_amd supports the "
                                 "ResNet-50 / -101 / -152 bottleneck bodies with at least three stages")
            # IntermediateLayerGetter drops every stage behind the last returned one: multi_scale is visible in the tensors
            multi_scale = len(layers)
            ms_attr = _attr(bmod, "_multi_scale")      # the reference stores it behind a property (resnet.py:62-68)
            if ms_attr is None:
                ms_attr = _attr(bmod, "multi_scale")
            if ms_attr is not None and max(1, min(4, int(ms_attr))) != len(layers):
                raise ValueError(f"checkpoint: backbone {v!r} has multi_scale={ms_attr} but {len(layers)} stages")
            # norm layer: torchvision's BatchNorm2d keeps running statistics AND counts batches; anything else
            # (FrozenBatchNorm2d, GroupNorm, ...) is not what the HIP plan computes
            if f"{p}bn1.num_batches_tracked" not in sd or f"{p}bn1.running_var" not in sd:
                raise ValueError(f"checkpoint: backbone {v!r} does not use BatchNorm2d")
            bb = {"name": name, "weights": "", "multi_scale": multi_scale, "norm_layer": "BatchNorm2d"}
            adj = f"backbones.{v}.adjustment_layer.weight"
            if adj in sd:
                bb["in_channels"] = int(sd[adj].shape[1])
            backbones[v] = bb
        p = f"necks.{v}.fpn.inner_blocks."
        n_in = len({k.split(".")[4] for k in sd if k.startswith(p)})
        if n_in:
            nmod = _child(_child(model, "necks"), v)
            if _attr(nmod, "norm_layer") is not None or any(k.split(".")[5] != "0" for k in sd if k.startswith(f"necks.{v}.fpn.")):
                raise ValueError(f"checkpoint: neck {v!r} uses a norm layer; dpft_amd's FPN has none (no reference config does)")
            necks[v] = {"name": "FPN", "in_channels_list": [int(sd[f"{p}{i}.0.weight"].shape[1]) for i in range(n_in)],
                        "out_channels": int(sd[f"{p}0.0.weight"].shape[0])}
        emb = _child(_child(model, "embeddings"), v) if _child(model, "embeddings") is not None else None
        if emb is not None and _attr(emb, "n_levels") is not None:
            layers_ = list(emb._modules.get("embedding_layers", nn.ModuleDict())._modules.values())
            if not layers_:
                raise ValueError(f"checkpoint: embedding {v!r} has no embedding layers")
            what = f"embedding {v!r}"
            keys = ("num_feats", "temperature", "normalize", "scale", "eps", "offset")
            per_level = [tuple(_hyper(l, k, None, what) for k in keys) for l in layers_]
            if any(t != per_level[0] for t in per_level):
                raise ValueError(f"checkpoint: {what} has per-level hyper-parameters {per_level}; one set is supported")
            embeddings[v] = {"name": "sinusoidal_embedding", "n_levels": int(_attr(emb, "n_levels")),
                             **dict(zip(keys, per_level[0]))}
    fuser, head, querent = _child(model, "fuser"), _child(model, "head"), _child(model, "querent")
    cfg_model: Dict[str, Any] = {"name": "dprt", "inputs": inputs,
                                 "skiplinks": dict(_attr(model, "skiplinks") or {v: True for v in inputs}),
                                 "backbones": backbones, "necks": necks, "embeddings": embeddings}
    if querent is not None and _attr(querent, "resolution") is not None:
        tr = _hyper(querent, "transformation", None, "querent") if "transformation" in querent.__dict__ \
            else querent._modules.get("transformation")
        tname = type(tr).__name__
        if tr is None or tname == "Identity":
            transformation = None
        elif tname in _TRANSFORMATIONS:
            transformation = _TRANSFORMATIONS[tname]
        else:
            raise ValueError(f"checkpoint: querent transformation {tname!r} cannot be mapped")
        cfg_model["querent"] = {"name": "data_agnostic_static_querent", "transformation": transformation,
                                "resolution": list(_attr(querent, "resolution")),
                                "minimum": list(_attr(querent, "minimum")), "maximum": list(_attr(querent, "maximum")),
                                "distribution": list(_hyper(querent, "distribution", "linear", "querent"))}
    if fuser is not None and _attr(fuser, "i_iter") is not None:
        keys = ("i_iter", "m_views", "d_model", "d_ffn", "n_queries", "n_levels", "n_heads", "n_points", "norm", "dropout",
                "reduction", "activation", "ffn_layer")
        cfg_model["fuser"] = {"name": "IMPFusion", **{k: _hyper(fuser, k, None, "fuser") for k in keys}}
    if head is not None and _attr(head, "num_classes") is not None:
        keys = ("in_channels", "num_classes", "num_reg_layers", "num_cls_layers", "bias", "dropout")
        cfg_model["head"] = {"name": "linear_detection_head", **{k: _hyper(head, k, None, "head") for k in keys}}
        if type(head).__name__ not in ("LinearDetectionHead",):
            raise ValueError(f"checkpoint: head class {type(head).__name__!r}; dpft_amd builds the linear detection head")
    return {"computing": {"dtype": "float32", "device": "cuda" if torch.cuda.is_available() else "cpu"},
            "model": cfg_model}


def load_reference_checkpoint(path, config: Dict[str, Any] = None) -> nn.Module:
    """A ``torch.save(model)`` file written by the reference's trainer (or the object ``read_foreign`` made of it) -> an
    equivalent ``dpft_amd`` DPRT (same weights, buffers and hyper-parameters).  ``config`` overrides the inferred one
    (e.g. a different dropout)."""
    from dpft_amd.models.dprt import build_dprt
    foreign = read_foreign(path) if isinstance(path, (str, bytes)) or hasattr(path, "__fspath__") else path
    if not isinstance(foreign, nn.Module):
        raise ValueError(f"{path}: expected a pickled torch.nn.Module, got {type(foreign).__name__}")
    cfg = config if config is not None else infer_config(foreign)
    model = build_dprt(cfg)
    missing, unexpected = model.load_state_dict(foreign.state_dict(), strict=False)
    if missing or unexpected:
        raise ValueError(f"{path}: state_dict mismatch after rebuilding the model from the checkpoint's own "
                         f"hyper-parameters (missing {missing[:5]}, unexpected {unexpected[:5]})")
    model.train(bool(foreign.__dict__.get("training", True)))
    return model
