"""Evaluation loop, mirror of ``src/dprt/evaluation/evaluator.py`` (``CentralizedEvaluator`` :19-209,
``build_evaluator`` :212-213): metrics + K-Radar export over a data loader, the 10 + 300 event-timed latency protocol
(:97-135) and the model complexity figures (:70-95).

Differences, all behind the same method names:
* metrics and export selection of a batch are a handful of HIP launches (``dpft_amd.evaluation.metric``,
  ``dpft_amd.evaluation.exporters.kradar``) instead of per-sample tensor-op chains;
* ``evaluate_complexity`` does not need the deepspeed profiler: the conv family's FLOPs are read from the library's
  own launch log (``dpft_profile_*``, the same counters bench.py's roofline uses) during one forward; MACs = FLOPs / 2
  of the convolutions (94 % of the model), parameters are counted from the module;
* scalars go to any object with ``add_scalar(name, value, step)``; without tensorboard a JSON-lines file is written.
"""
from __future__ import annotations

import json
import os
import os.path as osp
from typing import Any, Callable, Dict, Iterable, List

import torch


class JsonlWriter:
    """``SummaryWriter`` stand-in: one ``{"tag", "value", "step"}`` object per line in ``<log_dir>/scalars.jsonl``."""

    def __init__(self, log_dir: str):
        os.makedirs(log_dir, exist_ok=True)
        self._f = open(osp.join(log_dir, "scalars.jsonl"), "a")

    def add_scalar(self, tag: str, value, step: int) -> None:
        self._f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")

    def flush(self) -> None:
        self._f.flush()

    def close(self) -> None:
        self._f.close()


def make_writer(log_dir: str):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=log_dir)
    except Exception:       # tensorboard is not part of this image
        return JsonlWriter(log_dir)


class CentralizedEvaluator:
    def __init__(self, metric: torch.nn.Module = None, exporter: Callable = None, device: str = None,
                 logging: str = None):
        self.eval_fn = metric
        self.export_fn = exporter
        self.device = device
        self.logging = logging

    @classmethod
    def from_config(cls, config: Dict[str, Any], *args, **kwargs) -> "CentralizedEvaluator":
        from dpft_amd.evaluation.exporters import build as build_exporter
        from dpft_amd.evaluation.metric import build_metric
        return cls(metric=build_metric(config["evaluate"]),
                   exporter=build_exporter(config["evaluate"]["exporter"]["name"], config),
                   device=torch.device(config["computing"]["device"]), logging=config["train"].get("logging"))

    def __call__(self, *args: Any, **kwargs: Any) -> Any:
        self.evaluate(*args, **kwargs)

    @staticmethod
    def _dict_to(data: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
        return {k: v.to(device, non_blocking=True) for k, v in data.items()}

    @staticmethod
    def log_scalars(writer, scalars: Dict[str, Any], epoch: int, prefix: str = None) -> None:
        if writer is None:
            return
        prefix = f"{prefix}/" if prefix is not None else ""
        for name, scalar in scalars.items():
            writer.add_scalar(prefix + name, scalar, epoch)

    @torch.no_grad()
    def evaluate_complexity(self, epoch: int, model: torch.nn.Module, data_loader: Iterable, writer=None):
        """FLOPS / MACS / Parameters of one forward on the loader's first batch (evaluator.py:70-95)."""
        from dpft_amd.hip import ops
        model.eval()
        data, _ = next(iter(data_loader))
        data = self._dict_to(data, self.device)
        model(data)                                        # warm: plans, packed weights
        ops.profile_start()
        model(data)
        flops = sum(f for kind, f, _, _ in ops.profile_collect() if kind == "fwd")
        params = sum(p.numel() for p in model.parameters())
        scalars = {"FLOPS": flops, "MACS": flops / 2, "Parameters": params}
        self.log_scalars(writer, scalars, epoch, "test")
        return scalars

    @torch.no_grad()
    def evaluate_inference_time(self, epoch: int, model: torch.nn.Module, data_loader: Iterable, writer=None,
                                warmup: int = 10, repetitions: int = 300):
        """mean / std forward latency in ms over 300 event-timed passes after 10 warm-ups (evaluator.py:97-135)."""
        model.eval()
        data, _ = next(iter(data_loader))
        data = self._dict_to(data, self.device)
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
        scalars = {"Inference_time_mean_ms": torch.sum(timings) / repetitions, "Inference_time_std_ms": torch.std(timings)}
        self.log_scalars(writer, scalars, epoch, "test")
        return scalars

    @torch.no_grad()
    def evaluate_one_epoch(self, epoch: int, model: torch.nn.Module, data_loader: Iterable, writer=None,
                           dst: str = None):
        """metrics of every batch (+ export), averaged over the epoch when ``logging == 'epoch'`` (evaluator.py:137-183)."""
        model.eval()
        scalars: Dict[str, torch.Tensor] = {}
        i = -1
        for i, (data, labels) in enumerate(data_loader):
            labels: List[Dict[str, torch.Tensor]] = [self._dict_to(label, self.device) for label in labels]
            data = self._dict_to(data, self.device)
            output = model(data)
            metrics = self.eval_fn(output, labels)
            if self.logging == "step":
                self.log_scalars(writer, metrics, i + epoch * len(data_loader), "test")
            if self.logging == "epoch":
                for k, v in metrics.items():
                    scalars[k] = scalars.get(k, 0) + v
            if self.export_fn is not None:
                self.export_fn(output, labels, i * len(labels), dst)
        if self.logging == "epoch":
            scalars = {k: v / (i + 1) for k, v in scalars.items()}
            self.log_scalars(writer, scalars, epoch, "test")
        return scalars

    def evaluate(self, checkpoint: str, data_loader: Iterable, dst: str = None):
        """Loads ``<timestamp>_checkpoint_<epoch>.pt`` and runs the three evaluations (evaluator.py:185-209)."""
        from dpft_amd.models import load as load_model
        model, epoch, timestamp = load_model(checkpoint)
        model.to(self.device)
        writer = None
        if self.logging is not None:
            dst = osp.join(dst, timestamp)
            writer = make_writer(dst)
        self.evaluate_one_epoch(epoch, model, data_loader, writer, dst)
        self.evaluate_inference_time(epoch, model, data_loader, writer)
        self.evaluate_complexity(epoch, model, data_loader, writer)
        if writer is not None:
            writer.flush()
            writer.close()


def build_evaluator(*args, **kwargs):
    return CentralizedEvaluator.from_config(*args, **kwargs)
