"""hipGraph capture of the fusion decoder for training.

The decoder is ~700 tiny kernels per forward (B*400 x 16 tensors) and as many again backward: on MI355X
it is bound by host launch time, not by the GPU (≈15 ms forward / ≈25 ms backward of host time for
≈5 ms of kernels).  Its forward and its backward are recorded once each (static shapes, static
addresses) and replayed with a single launch.  The fused cross-attention HIP kernels are launched on
the capturing stream through the C-ABI, so they are part of the graphs.

Capture is done by hand (not torch.cuda.make_graphed_callables) so that the two graphs live in
SEPARATE memory pools: the backward graph's static gradient buffers must never share addresses with
forward temporaries of the same pool.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional

import torch
from torch import nn

from dpft_amd.hip import ops
from dpft_amd.hip.lib import lib, stream


# Other threads of the process (the RCCL watchdog of torch.distributed polls its events) must not invalidate a capture
# in progress: only this thread's calls are checked.
CAPTURE_MODE = "thread_local"


class _FlatFuser(nn.Module):
    """Tensor-only signature around IMPFusion.forward."""

    def __init__(self, fuser: nn.Module, level_keys: List[List[str]], flags: List[bool]):
        super().__init__()
        self.fuser = fuser
        self.level_keys = level_keys
        self.flags = flags

    def forward(self, center0, *tensors):
        it = iter(tensors)
        views = [OrderedDict((k, next(it)) for k in keys) for keys in self.level_keys]
        n = len(self.level_keys)
        shapes = [next(it)[:, :2] for _ in range(n)]
        proj = [(next(it), next(it)) for _ in range(n)]
        out = self.fuser(batch=views, shape=shapes, projection=proj, out=OrderedDict(center=center0),
                         has_transformation=self.flags)
        return out["center"], out["size"], out["angle"], out["class"]


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g: "GraphedFuser", *inputs):
        # this step's small inputs (shapes, projection matrices) into the graph's static buffers: one launch for all of them
        if g.pace_event is not None:
            g.pace_event.record()                           # (DataParallelTrainer.train_step waits for it on the HOST)
        ops.memops([(s, a) for s, a in zip(g.static_inputs, inputs[:len(g.static_inputs)]) if s.data_ptr() != a.data_ptr()])
        g.fwd_graph.replay()
        ctx.g = g
        g.last_inputs = inputs[1:1 + g.n_levels]            # this step's pyramid tensors (see backward_from)
        if not g.clone_outputs:                             # the caller consumes the outputs before the next replay
            return tuple(o.detach() for o in g.static_outputs)
        return tuple(o.detach().clone() for o in g.static_outputs)

    @staticmethod
    def backward(ctx, *grads):
        g = ctx.g
        for s, a in zip(g.static_grad_outputs, grads):
            s.copy_(a) if a is not None else s.zero_()
        g.bwd_graph.replay()
        n_other = len(g.static_inputs) - 1 - g.n_levels          # shapes + projection matrices
        # The static gradient buffers are handed out as they are: their consumers (embedding / FPN backward,
        # gradient accumulation) run before the next replay overwrites them.
        gi = [None if t is None else t.detach() for t in g.static_grad_inputs]
        if g.grad_direct is not None:      # the graph already added the parameter gradients into the buckets
            g.grad_direct.mark_ready_many(g.params_with_grad)
            return (None, None, *gi[:g.n_levels], *([None] * (n_other + len(g.params))))
        return (None, None, *gi[:g.n_levels], *([None] * n_other), *gi[g.n_levels:])


class GraphedFuser:
    last_inputs = None
    clone_outputs = True        # False: hand out the graph's static output buffers (DataParallelTrainer.train_step)
    pace_event = None           # a torch.cuda.Event recorded in front of the decoder's forward graph when the trainer sets one

    def level_buffers(self, view: str, feats) -> "Optional[List[torch.Tensor]]":
        """The static pyramid inputs of ``view`` (in the neck's level order) as plain tensors sharing their storage, or None
        when this step's shapes differ from the captured ones."""
        try:
            vi = self.inputs.index(view)
        except ValueError:
            return None
        start = 1 + sum(len(k) for k in self.level_keys[:vi])
        bufs = self.static_inputs[start:start + len(self.level_keys[vi])]
        if list(feats.keys()) != self.level_keys[vi] or getattr(self, "_neck_direct", True) is False:
            return None
        for b, x in zip(bufs, feats.values()):
            if b.shape[0] != x.shape[0] or tuple(b.shape[1:3]) != tuple(x.shape[1:3]):
                return None
        return [b.detach() for b in bufs]

    def backward_from(self, write_output_grads) -> None:
        """The backward of the replayed decoder WITHOUT the autograd engine in front of it: ``write_output_grads`` fills
        ``static_grad_outputs`` (center, size, angle, class) in place, the backward graph is launched at once, and only
        then autograd is started on the pyramid tensors with the graph's input gradients -- the engine's start-up
        (~0.2 ms on the host, after the step's host sync, with an idle GPU) hides behind the decoder's backward."""
        levels = self.last_inputs
        self.last_inputs = None
        write_output_grads(self.static_grad_outputs)
        self.bwd_graph.replay()
        if self.grad_direct is not None:
            self.grad_direct.mark_ready_many(self.params_with_grad)
        else:
            for p, t in zip(self.params, self.static_grad_inputs[self.n_levels:]):
                if t is not None:
                    p.grad = t.detach().clone() if p.grad is None else p.grad.add_(t)
        pairs = [(l, t.detach()) for l, t in zip(levels, self.static_grad_inputs[:self.n_levels])
                 if t is not None and l.requires_grad]
        torch.autograd.backward([l for l, _ in pairs], [t for _, t in pairs])

    def __init__(self, model: nn.Module, sample_batch: Dict[str, torch.Tensor], warmup: int = 3, grad_direct=None):
        self.inputs = list(model.inputs)
        self.grad_direct = grad_direct
        was_training = model.training
        # one eval-mode pass up to the fuser (no running-stat update, no autograd) for correctly shaped samples
        model.eval()
        with torch.no_grad():
            feats = {i: model.backbones[i](sample_batch[i]) for i in self.inputs}
            feats = {i: model._add_raw_data(feats[i], sample_batch[i]) for i in self.inputs}
            feats = {i: model.embeddings[i](model.necks[i](feats[i])) for i in self.inputs}
        proj = model._get_projetions(self.inputs, sample_batch)
        self.flags = model.fuser.transformation_flags(proj)
        self.level_keys = [list(feats[i].keys()) for i in self.inputs]
        center0 = model.querent(sample_batch)["center"]
        static = [center0.detach()]          # the querent's cached constant: the same storage every step, never copied
        for i in self.inputs:
            static += [v.detach().clone().requires_grad_(True) for v in feats[i].values()]
        self.n_levels = len(static) - 1
        # (the whole (B, 3) shape rows: contiguous, so that a step's rows go in with the other small inputs; the fuser reads
        # the first two columns)
        static += [sample_batch[f"{i}_shape"].clone() for i in self.inputs]
        for t, p in proj:
            static += [t.detach().clone(), p.detach().clone()]
        model.train()
        self.flat = _FlatFuser(model.fuser, self.level_keys, self.flags).train()
        self.params = [p for p in self.flat.parameters() if p.requires_grad]
        self.static_inputs = static
        self.shapes = [tuple(s.shape) for s in static]
        diff_inputs = static[1:1 + self.n_levels] + self.params

        # warm-up on a side stream (lazy initialisations, allocator growth) -- nothing is accumulated into .grad
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.enable_grad(), torch.cuda.stream(side):
            for _ in range(warmup):
                outs = self.flat(*static)
                torch.autograd.grad(outs, diff_inputs, [torch.ones_like(o) for o in outs], allow_unused=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        self.fwd_graph = torch.cuda.CUDAGraph()
        with torch.enable_grad(), torch.cuda.graph(self.fwd_graph, capture_error_mode=CAPTURE_MODE):
            self.static_outputs = self.flat(*static)
        self.static_grad_outputs = [torch.zeros_like(o) for o in self.static_outputs]
        self.bwd_graph = torch.cuda.CUDAGraph()
        # (dst, src, bytes) rows of the parameter-gradient additions: the addresses only exist once the capture has run, so
        # the launch is captured against an empty DEVICE table that is filled right after
        add_table = torch.empty((max(len(self.params), 1), 3), dtype=torch.int64, device=center0.device) if grad_direct is not None else None
        add_rows = []
        with torch.enable_grad(), torch.cuda.graph(self.bwd_graph, capture_error_mode=CAPTURE_MODE):   # its own private pool (see module docstring)
            grads = torch.autograd.grad(self.static_outputs, diff_inputs, self.static_grad_outputs,
                                        allow_unused=True)
            from dpft_amd.models.fusers import train_fused as _tf
            _tf.join_forked()      # the weight-gradient branches of the capture end here
            if grad_direct is not None:
                pairs = [(grad_direct.grad_buffer(p), t) for p, t in zip(self.params, grads[self.n_levels:]) if t is not None]
                if any(v is None for v, _ in pairs):
                    raise RuntimeError("GraphedFuser: a decoder parameter is not owned by the gradient reducer")
                plain = [(v, t) for v, t in pairs if v.is_contiguous() and t.is_contiguous() and v.numel() == t.numel()
                         and v.dtype == t.dtype == torch.float32]
                other = [(v, t) for v, t in pairs if not any(v is pv for pv, _ in plain)]
                if plain:      # one launch for all of them (dpft_add_many_f32)
                    add_rows = [(v.data_ptr(), t.data_ptr(), 4 * v.numel()) for v, t in plain]
                    lib.call("dpft_add_many_f32", len(add_rows), add_table.data_ptr(), stream())
                if other:
                    torch._foreach_add_([v for v, _ in other], [t for _, t in other])
        if add_rows:
            add_table[:len(add_rows)].copy_(torch.tensor(add_rows, dtype=torch.int64))
        self._add_table = add_table
        self.static_grad_inputs = list(grads)
        self.params_with_grad = [p for p, t in zip(self.params, grads[self.n_levels:]) if t is not None]
        torch.cuda.synchronize()
        # inference replay: eval mode (dropout off, MHA fast path), no autograd
        self.flat.eval()
        self.eval_inputs = [t.detach().clone() for t in static]
        with torch.no_grad():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.flat(*self.eval_inputs)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.eval_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.eval_graph, capture_error_mode=CAPTURE_MODE):
                self.eval_outputs = self.flat(*self.eval_inputs)
        torch.cuda.synchronize()
        self.flat.train()
        model.train(was_training)

    def __call__(self, features, shapes, projection, out):
        args = [out["center"]]
        for i in self.inputs:
            args += list(features[i].values())
        args += [shapes[i] for i in self.inputs]
        for t, p in projection:
            args += [t, p]
        if [tuple(a.shape) for a in args] != self.shapes:
            raise RuntimeError("graphed fuser called with shapes different from the captured ones")
        if not torch.is_grad_enabled():
            ops.memops(list(zip(self.eval_inputs, args)))
            self.eval_graph.replay()
            c, s, a, k = (o.clone() for o in self.eval_outputs)
            return OrderedDict([("center", c), ("size", s), ("angle", a), ("class", k)])
        # parameters are passed so that autograd routes their gradients (their values are read in place)
        c, s, a, k = _Replay.apply(self, *args, *self.params)
        return OrderedDict([("center", c), ("size", s), ("angle", a), ("class", k)])
