"""Data-parallel evaluation loop: the counterpart of ``CentralizedEvaluator`` (src/dprt/evaluation/evaluator.py:19-214) on this
package's device-side pieces.

``evaluate_one_epoch`` (evaluator.py:138-177) = eval-mode forward -> ``Metric`` (two launches per batch) -> exporter (one
selection launch per batch and threshold chunk) for every batch of the test split.  Under ``torch.distributed`` every rank
takes a CONTIGUOUS block of the split (``BlockShardedSampler``: sample k keeps the file name it has in a one-process run),
exports into a private root, and rank 0 merges the rank trees afterwards: numbered files are moved, the appended ``val.txt``
lists are concatenated in rank (= sample) order -- the merged tree is the one-process tree, file for file.  Metric means are
sums over all steps of all ranks divided by the global step count (one collective).

``evaluate_inference_time`` keeps the reference's protocol (evaluator.py:109-136: 10 warm-up + 300 event-timed forwards of one
batch, every forward followed by a device sync).  ``evaluate_complexity`` (:67-93) wraps deepspeed's FLOP profiler, which is
not installed here: the parameter count is logged, FLOPS / MACS are left to ``bench.py``'s algorithmic counts."""
from __future__ import annotations

import os
import os.path as osp
import shutil
from typing import Any, Callable, Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

from dpft_amd.evaluation.exporters import build as build_exporter
from dpft_amd.evaluation.metric import build_metric


APPENDED_LISTS = ("val.txt",)      # the split lists the exporter appends one line per sample to (exporters: kitti)


def merge_rank_exports(dst: str, world: int) -> None:
    """``<dst>/_rank<r>/exports/...`` of ranks 0 .. world-1 -> ``<dst>/exports/...``.  Only the appended split lists
    (``APPENDED_LISTS``) may exist in several rank trees: they are concatenated in rank order.  Any OTHER name collision
    means two ranks numbered their per-sample files from the same index (a missing ``shard_start``): that raises instead of
    silently gluing two samples' predictions into one file."""
    for r in range(world):
        root = osp.join(dst, f"_rank{r}")
        if not osp.isdir(root):
            continue
        for cur, _, files in os.walk(root):
            rel = osp.relpath(cur, root)
            out_dir = osp.join(dst, rel) if rel != "." else dst
            os.makedirs(out_dir, exist_ok=True)
            for f in sorted(files):
                src, out = osp.join(cur, f), osp.join(out_dir, f)
                if osp.exists(out):
                    if f not in APPENDED_LISTS:
                        raise RuntimeError(f"merge_rank_exports: rank {r} wrote {osp.join(rel, f)}, which another rank wrote too -- "
                                           "the ranks' blocks overlap (evaluate_one_epoch needs each rank's shard_start)")
                    with open(out, "a") as fo, open(src) as fi:
                        shutil.copyfileobj(fi, fo)
                else:
                    shutil.move(src, out)
        shutil.rmtree(root)


class DataParallelEvaluator:
    latency_reps, latency_warmup = 300, 10      # evaluator.py:110,114

    def __init__(self, metric: Optional[torch.nn.Module] = None, exporter: Optional[Callable] = None,
                 device: Optional[torch.device] = None, logging: Optional[str] = None):
        """``logging``: None, 'step' or 'epoch' (evaluator.py:24-28)."""
        self.eval_fn, self.export_fn, self.logging = metric, exporter, logging
        self.device = torch.device(device) if device is not None else torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1

    @classmethod
    def from_config(cls, config: Dict[str, Any], *args, **kwargs) -> "DataParallelEvaluator":
        """evaluator.py:36-52 (``computing.device`` 'cuda' -> this rank's GPU)."""
        device = torch.device(config["computing"]["device"])
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
        return cls(metric=build_metric(config["evaluate"]),
                   exporter=build_exporter(config["evaluate"]["exporter"]["name"], config),
                   device=device, logging=config["train"].get("logging"))

    def __call__(self, *args, **kwargs):
        return self.evaluate(*args, **kwargs)

    def _dict_to(self, data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        return {k: (v.to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in data.items()}

    @staticmethod
    def log_scalars(writer, scalars: Dict[str, Any], epoch: int, prefix: str = None) -> None:
        if writer is None:
            return
        prefix = f"{prefix}/" if prefix is not None else ""
        for name, scalar in scalars.items():
            writer.add_scalar(prefix + name, float(scalar), epoch)

    # ------------------------------------------------------------------------------------------------------ the epoch
    @torch.no_grad()
    def evaluate_one_epoch(self, epoch: int, model: torch.nn.Module, data_loader: Iterable, writer=None, dst: str = None,
                           shard_start: Optional[int] = None, rank: Optional[int] = None, world: Optional[int] = None
                           ) -> Dict[str, float]:
        """evaluator.py:138-177 on this rank's block of the split.  ``shard_start`` = global index of the block's first sample
        (default: the loader's ``sampler.start``, else 0) -- the exporter numbers its files from it.  ``rank`` / ``world``
        default to the process group's (tests pass them to play several ranks in one process).  Returns the metric means over
        ALL ranks' steps."""
        rank = self.rank if rank is None else rank
        world = self.world if world is None else world
        if shard_start is None:
            sampler = getattr(data_loader, "sampler", None)
            if sampler is None:      # PrefetchLoader / DataLoader wrappers: the loader they wrap
                sampler = getattr(getattr(data_loader, "source", None), "sampler", None)
            start = getattr(sampler, "start", None)
            if start is None and world > 1 and dst is not None and self.export_fn is not None:
                raise ValueError("evaluate_one_epoch: with more than one rank the exporter needs the global index of this rank's "
                                 "first sample: pass shard_start= or a loader over a BlockShardedSampler "
                                 "(dpft_amd.data.loader.load_listed_eval)")
            shard_start = int(start or 0)
        model.eval()
        root = osp.join(dst, f"_rank{rank}") if (dst is not None and world > 1) else dst
        sums: Dict[str, torch.Tensor] = {}
        steps, seen = 0, 0
        n_steps = len(data_loader) if hasattr(data_loader, "__len__") else 0
        for i, (data, labels) in enumerate(data_loader):
            labels = [self._dict_to(l) for l in labels]
            data = self._dict_to(data)
            output = model(data)
            metrics = self.eval_fn(output, labels) if self.eval_fn is not None else {}
            if self.logging == "step" and rank == 0:
                self.log_scalars(writer, metrics, i + epoch * n_steps, "test")
            for k, v in metrics.items():
                v = torch.as_tensor(v, device=self.device).detach().float().reshape(())
                sums[k] = sums[k] + v if k in sums else v.clone()
            if self.export_fn is not None and root is not None:
                self.export_fn(output, labels, shard_start + seen, root)
            seen += len(labels)
            steps += 1
        keys = sorted(sums)
        vec = torch.stack([sums[k] for k in keys] + [torch.tensor(float(steps), device=self.device)]) if keys else \
            torch.tensor([float(steps)], device=self.device)
        if dist.is_initialized() and self.world > 1:
            n = torch.tensor([len(keys), -len(keys)], dtype=torch.int64, device=self.device)      # same keys on every rank?
            dist.all_reduce(n, op=dist.ReduceOp.MAX)
            if int(n[0]) != -int(n[1]):
                raise RuntimeError(f"rank {rank}: {len(keys)} metric keys, another rank has {int(n[0])} / {-int(n[1])} "
                                   "(a rank saw no batches)")
            dist.all_reduce(vec, op=dist.ReduceOp.SUM)
            dist.barrier()                              # every rank's files are on disk
            if rank == 0 and dst is not None and self.export_fn is not None:
                merge_rank_exports(dst, world)
            dist.barrier()
        total = max(float(vec[-1]), 1.0)
        scalars = {k: float(vec[j]) / total for j, k in enumerate(keys)}
        if self.logging == "epoch" and rank == 0:
            self.log_scalars(writer, scalars, epoch, "test")
        return scalars

    @torch.no_grad()
    def evaluate_inference_time(self, epoch: int, model: torch.nn.Module, data_loader: Iterable, writer=None,
                                repetitions: int = None, warmup: int = None):
        """evaluator.py:95-136: one batch, ``warmup`` untimed forwards, then ``repetitions`` forwards each bracketed by events
        and followed by a device sync.  Returns (mean ms, std ms) per batch."""
        model.eval()
        repetitions = self.latency_reps if repetitions is None else repetitions
        warmup = self.latency_warmup if warmup is None else warmup
        data, _ = next(iter(data_loader))
        data = self._dict_to(data)
        starter, ender = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        timings = torch.zeros((repetitions, 1))
        for _ in range(warmup):
            model(data)
        for rep in range(repetitions):
            starter.record()
            model(data)
            ender.record()
            torch.cuda.synchronize()
            timings[rep] = starter.elapsed_time(ender)
        mean_syn, std_syn = torch.sum(timings) / repetitions, torch.std(timings)
        if self.rank == 0:
            self.log_scalars(writer, {"Inference_time_mean_ms": mean_syn, "Inference_time_std_ms": std_syn}, epoch, "test")
        return float(mean_syn), float(std_syn)

    @torch.no_grad()
    def evaluate_complexity(self, epoch: int, model: torch.nn.Module, data_loader: Iterable, writer=None) -> Dict[str, float]:
        """evaluator.py:70-94: FLOPS, MACS and Parameters of one forward on the first batch.  deepspeed's profiler is absent
        (and blind to launch plans): the counts come from the library's conv launch log + the decoder's analytic table
        (dpft_amd/evaluation/complexity.py)."""
        from dpft_amd.evaluation.complexity import model_complexity
        data, _ = next(iter(data_loader))
        c = model_complexity(model, self._dict_to(data))
        out = {k: c[k] for k in ("FLOPS", "MACS", "Parameters")}
        if self.rank == 0:
            self.log_scalars(writer, out, epoch, "test")
        return out

    def evaluate(self, checkpoint: str, data_loader: Iterable, dst: str = None) -> Dict[str, float]:
        """evaluator.py:179-210: model from a checkpoint (this package's or the reference's pickle), metrics + export over
        the split, inference time, complexity; scalars go to rank 0's writer."""
        from dpft_amd.models import load as load_model
        from dpft_amd.training.trainer import _ScalarLog
        model, epoch, timestamp = load_model(checkpoint)
        model.to(self.device)
        if self.logging is not None and dst is not None:
            dst = osp.join(dst, timestamp)
        writer = _ScalarLog(dst) if (self.logging is not None and dst is not None and self.rank == 0) else None
        scalars = self.evaluate_one_epoch(epoch, model, data_loader, writer, dst)
        self.evaluate_inference_time(epoch, model, data_loader, writer)
        self.evaluate_complexity(epoch, model, data_loader, writer)
        if writer is not None:
            writer.close()
        return scalars


def build_evaluator(*args, **kwargs):
    return DataParallelEvaluator.from_config(*args, **kwargs)
