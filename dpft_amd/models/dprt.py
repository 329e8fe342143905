"""Dual Perspective Radar Transformer -- top-level module.

Mirror of ``src/dprt/models/dprt.py`` (DPRT :67-244): same constructor, ``from_config``, batch-dict
contract (``X``, ``X_shape``, ``label_to_X_t``, ``label_to_X_p`` per input) and output dict
(``center, size, angle, class``), same ``state_dict`` names.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Any, Callable, Dict, List, Tuple

import torch
from torch import nn

from dpft_amd.models.backbones import build_backbone
from dpft_amd.models.embeddings import build_embedding
from dpft_amd.models.fusers import build_fuser
from dpft_amd.models.heads import build_head
from dpft_amd.models.necks import build_neck
from dpft_amd.models.queries import build_querent


def _build_module(build_fn: Callable, module_name: str, config: Dict[str, Any], computing: Dict[str, Any],
                  *args, **kwargs) -> nn.Module:
    module = config.get(module_name)
    if module is not None:
        return build_fn(module["name"], dict(computing | module), *args, **kwargs)
    return None


def _build_modules(build_fn: Callable, module_name: str, config: Dict[str, Any], computing: Dict[str, Any],
                   *args, **kwargs) -> Dict[str, nn.Module]:
    modules = config.get(module_name)
    if modules is not None:
        return {k: _build_module(build_fn, k, modules, computing, *args, **kwargs) for k in modules.keys()}
    return None


class DPRT(nn.Module):
    def __init__(self, inputs: List[str], skiplinks: Dict[str, bool] = None,
                 backbones: Dict[str, nn.Module] = None, necks: Dict[str, nn.Module] = None,
                 embeddings: Dict[str, nn.Module] = None, querent: nn.Module = None, fuser: nn.Module = None,
                 head: nn.Module = None, **kwargs):
        super().__init__()
        self.inputs = inputs
        self.skiplinks = skiplinks if skiplinks is not None else {}
        self.skiplinks = {i: self.skiplinks.get(i, False) for i in inputs}
        self.backbones = self._init_unspecified(backbones if backbones is not None else {})
        self.necks = self._init_unspecified(necks if necks is not None else {})
        self.embeddings = self._init_unspecified(embeddings if embeddings is not None else {})
        self.querent = self._module_or_identity(querent)
        self.fuser = self._module_or_identity(fuser)
        self.head = self._module_or_identity(head)

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "DPRT":
        computing, model = config["computing"], config["model"]
        head = _build_module(build_head, "head", model, computing)
        fuser = _build_module(build_fuser, "fuser", model, computing, head=head)
        return cls(inputs=model.get("inputs"), skiplinks=model.get("skiplinks"),
                   backbones=_build_modules(build_backbone, "backbones", model, computing),
                   necks=_build_modules(build_neck, "necks", model, computing),
                   embeddings=_build_modules(build_embedding, "embeddings", model, computing),
                   querent=_build_module(build_querent, "querent", model, computing),
                   fuser=fuser, head=head)

    def _init_unspecified(self, submodule: Dict[str, nn.Module]) -> nn.ModuleDict:
        return nn.ModuleDict({i: self._module_or_identity(submodule.get(i)) for i in self.inputs})

    @staticmethod
    def _module_or_identity(module: nn.Module = None) -> nn.Module:
        return module if module is not None else nn.Identity()

    @staticmethod
    def _add_raw_data(features: "OrderedDict[str, torch.Tensor]", raw_data: torch.Tensor):
        features["0"] = raw_data
        features.move_to_end("0", last=False)
        return features

    @staticmethod
    def _get_projetions(inputs: List[str], batch: Dict[str, torch.Tensor]) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        return [(batch[f"label_to_{i}_t"], batch[f"label_to_{i}_p"]) for i in inputs]

    def forward(self, batch: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
        shapes = {i: batch[f"{i}_shape"] for i in self.inputs}
        features = self._encode_views(batch)
        out = self.querent(batch)
        projection = self._get_projetions(self.inputs, batch)
        graphed = self.__dict__.get("_graphed_fuser")
        if graphed is not None and self.training and torch.is_grad_enabled():
            return graphed(features, shapes, projection, out)
        return self.fuser(batch=[features[i] for i in self.inputs],
                          shape=[shapes[i][:, :2] for i in self.inputs],
                          projection=projection, out=out)

    def _encode_view(self, i: str, batch: Dict[str, torch.Tensor]):
        f = self.backbones[i](batch[i])                                         # dprt.py:219
        if self.skiplinks[i]:
            f = self._add_raw_data(f, batch[i])                                 # :222-225
        bufs = None
        graphed = self.__dict__.get("_graphed_fuser")
        if graphed is not None and self.training and torch.is_grad_enabled():
            bufs = graphed.level_buffers(i, f)           # the decoder graph's static inputs: the neck writes them directly
        # neck -> embedding (:228, :231).  A sinusoidal embedding behind an FPN neck is added by the neck's output convs themselves
        # (dpft_fpn_output_f32: the level is written once instead of written, read and written again)
        emb, neck = self.embeddings[i], self.necks[i]
        pos = emb.level_tables(f, neck.out_channels) if hasattr(emb, "level_tables") and getattr(neck, "channel_last", False) \
            and hasattr(neck, "fpn") and os.environ.get("DPFT_FPN_FUSE", "1") != "0" else None
        if pos is not None:
            return neck(f, out_buffers=bufs, pos=pos)
        if bufs is not None:
            return emb(neck(f, out_buffers=bufs))
        return emb(neck(f))

    def _encode_views(self, batch: Dict[str, torch.Tensor]) -> Dict[str, "OrderedDict[str, torch.Tensor]"]:
        """backbone -> (raw skip link) -> neck -> embedding per input.  On the GPU the views run on separate HIP
        streams: the radar branches are tiny GEMMs that cannot fill 256 CUs and hide inside the camera
        branch (autograd replays each view's backward on its forward stream, so the backward overlaps too)."""
        first = batch[self.inputs[0]]
        if len(self.inputs) == 1 or not first.is_cuda or not self.concurrent_views:
            return {i: self._encode_view(i, batch) for i in self.inputs}
        main = torch.cuda.current_stream(first.device)
        if self.__dict__.get("_view_streams") is None or len(self._view_streams) != len(self.inputs) - 1:
            self._place_streams(first.device)
        features = {}
        if torch.is_grad_enabled():
            # training: the small views first.  The host runs ahead of the GPU here (the previous step's backward is still
            # executing), so the issue order of the forward costs nothing, and autograd replays the LAST-created branch
            # first: the camera's backward chain -- the critical path of the step -- is queued before the radar ones.
            for i, s in zip(self.inputs[1:], self._view_streams):
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    features[i] = self._encode_view(i, batch)
            features[self.inputs[0]] = self._encode_view(self.inputs[0], batch)
        else:
            # inference (fwd ms/frame, evaluator.py:109-125): the GPU is idle when the forward starts, so the longest
            # chain must be queued first -- with the radar views first the camera's first kernel waited 3.7 ms for the
            # host to issue their ~260 launches (tools/eval_trace.sh).  The side streams wait for what preceded the
            # forward on the main stream (an event), not for the camera.
            ev = main.record_event()
            features[self.inputs[0]] = self._encode_view(self.inputs[0], batch)
            for i, s in zip(self.inputs[1:], self._view_streams):
                s.wait_event(ev)
                with torch.cuda.stream(s):
                    features[i] = self._encode_view(i, batch)
        for i, s in zip(self.inputs[1:], self._view_streams):
            main.wait_stream(s)
            for t in features[i].values():
                t.record_stream(main)
        return {i: features[i] for i in self.inputs}

    concurrent_views = True

    def _place_streams(self, device):
        """One hardware queue per concurrent chain.  The runtime multiplexes HIP streams onto 4 hardware queues and streams
        that share a queue run in order, so the step's overlap exists only if the chains sit on different queues
        (tools/probes/stream_queues.py).  dpft_stream_set finds streams on queues distinct from the main stream's and from
        each other: view i > 0 runs (forward, backward AND its plan's weight gradients) on its own queue; the first
        view -- the camera encoder, the critical path -- keeps the main stream and gets the remaining queue for its
        weight-gradient GEMMs.  DPFT_STREAM_PLACEMENT=pool restores pool streams / plan-created side streams (A/B)."""
        n_side = len(self.inputs) - 1
        if os.environ.get("DPFT_STREAM_PLACEMENT", "probe") == "pool":
            if os.environ.get("DPFT_SHARED_VIEW_STREAM", "0") == "1":
                one = torch.cuda.Stream(device)
                self.__dict__["_view_streams"] = [one for _ in self.inputs[1:]]
            else:
                self.__dict__["_view_streams"] = [torch.cuda.Stream(device) for _ in self.inputs[1:]]
            return
        from dpft_amd.hip.lib import stream_set
        streams, distinct = stream_set(device, 3)
        views = [streams[i % len(streams)] for i in range(n_side)]
        self.__dict__["_view_streams"] = views
        self.__dict__["_queues_found"] = distinct
        for i, s in zip(self.inputs[1:], views):
            if hasattr(self.backbones[i], "side_stream"):
                self.backbones[i].side_stream = s                  # in order with the view's own chain
        if hasattr(self.backbones[self.inputs[0]], "side_stream") and n_side < len(streams):
            self.backbones[self.inputs[0]].side_stream = streams[n_side]

    def enable_fuser_graph(self, sample_batch: Dict[str, torch.Tensor], grad_direct=None):
        """Capture the launch-bound fusion decoder (forward and backward) into hipGraphs for training steps
        with the static shapes of ``sample_batch`` (dpft_amd/models/fusers/graphed.py).  ``grad_direct``: a
        GradBucketReducer whose bucket views the backward graph adds the parameter gradients into."""
        from dpft_amd.models.fusers.graphed import GraphedFuser
        self.__dict__["_graphed_fuser"] = GraphedFuser(self, sample_batch, grad_direct=grad_direct)

    def disable_fuser_graph(self):
        self.__dict__.pop("_graphed_fuser", None)

    # process-local execution state parked in __dict__ (HIP streams, the probed queue set, the captured decoder graphs):
    # none of it belongs in ``torch.save(model)`` (trainer.py:256-258 pickles the whole module every epoch) or in a
    # deepcopy; all of it is rebuilt lazily by the next forward / ``enable_fuser_graph``
    _RUNTIME_STATE = ("_view_streams", "_queues_found", "_graphed_fuser")

    def __getstate__(self):
        st = self.__dict__.copy()
        for k in self._RUNTIME_STATE:
            st.pop(k, None)
        return st


def build_dprt(*args, **kwargs):
    return DPRT.from_config(*args, **kwargs)
