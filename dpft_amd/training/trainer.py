"""Training / validation / latency loops with the reference's step order and timing protocol.

Mirror of ``CentralizedTrainer.train_one_epoch`` (src/dprt/training/trainer.py:99-160: zero_grad ->
forward -> loss -> if loss > 0: backward, optimizer.step) and of
``CentralizedEvaluator.evaluate_inference_time`` (src/dprt/evaluation/evaluator.py:97-135: 10 warm-up
+ 300 event-timed forwards), extended to one-process-per-GPU data parallelism; the epoch loop
(``train`` / ``train_one_epoch`` / ``validate_one_epoch``, trainer.py:99-263) is the DP counterpart of the
reference's: every rank walks its shard, scalars are averaged over ranks, rank 0 alone writes logs and the
whole-module checkpoint ``<timestamp>_checkpoint_<epoch>.pt`` that ``dpft_amd.models.load`` (and the reference's
``dprt.train --checkpoint``, train.py:47-48) resumes from.
"""
from __future__ import annotations

import datetime
import json
import os
import os.path as osp

from typing import Any, Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

from dpft_amd.training.distributed import GradBucketReducer, broadcast_module
from dpft_amd.training.loss import build_loss
from dpft_amd.training.optimizer import FusedAdamW, build_optimizer


class _ScalarLog:
    """Rank-0 scalar sink with ``SummaryWriter``'s ``add_scalar`` signature.  TensorBoard when it is importable (the
    reference's writer, trainer.py:225), else one JSON line per scalar in ``<log_dir>/scalars.jsonl``."""

    def __init__(self, log_dir: str):
        os.makedirs(log_dir, exist_ok=True)
        self._tb, self._fh = None, None
        try:
            from torch.utils.tensorboard import SummaryWriter
            self._tb = SummaryWriter(log_dir=log_dir)
        except Exception:
            self._fh = open(osp.join(log_dir, "scalars.jsonl"), "a")

    def add_scalar(self, tag: str, value, step: int):
        if self._tb is not None:
            self._tb.add_scalar(tag, value, step)
        else:
            self._fh.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")

    def close(self):
        if self._tb is not None:
            self._tb.flush()
            self._tb.close()
        else:
            self._fh.close()


class DataParallelTrainer:
    def __init__(self, model: torch.nn.Module, config: Dict[str, Any], device, bucket_mb: Optional[float] = None,
                 comm_dtype: Optional[str] = None, force_collectives: bool = False):
        """``force_collectives``: take the N>1 exchange path (bucket all-reduces on the communicator's stream, the
        all-ranks step decision) even in a one-rank process group -- the RCCL path on a single GPU."""
        device = torch.device(device)
        self.model = model.to(device)
        self.device = device
        # optional mixed precision (BASELINE.json configs[4]): config["computing"]["conv_compute"] = "bf16" runs the
        # forward / data-gradient conv GEMMs with bf16 operands and fp32 accumulation; default = the reference's fp32
        if device.type == "cuda":
            from dpft_amd.hip import ops as _ops
            _ops.conv_set_compute(config.get("computing", {}).get("conv_compute") or os.environ.get("DPFT_CONV_COMPUTE", "fp32"))
        train = config["train"]
        self.loss_fn = build_loss(train)
        if self.device.type == "cuda" and hasattr(self.loss_fn, "assign_on_device"):
            # assignments on the device (csrc/lsap.hip): the step has no host sync; a non-finite cost matrix surfaces as the
            # ValueError scipy raises, at the next place this class reads values back (_check_matcher)
            self.loss_fn.assign_on_device = True
        # the reference evaluates mAP3D / mGIoU3D in every training step (trainer.py:134); optional here
        self.eval_fn = None
        if config.get("evaluate", {}).get("metrics"):
            from dpft_amd.evaluation import build_metric
            self.eval_fn = build_metric(config["evaluate"])
        opt = dict(train["optimizer"])
        name = opt.pop("name")
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        broadcast_module(self.model)
        # one bucket group per view encoder (their backward runs on separate streams), one for the rest
        group_of = {}
        for n, p in self.model.named_parameters():
            parts = n.split(".")
            group_of[id(p)] = ".".join(parts[:2]) if parts[0] in ("backbones", "necks") else "decoder"
        # exchange knobs: argument > config["train"]["dp"] > environment > default (25 MiB fp32 buckets)
        dp = train.get("dp", {})
        bucket_mb = float(bucket_mb or dp.get("bucket_mb") or os.environ.get("DPFT_BUCKET_MB") or 25)
        comm_dtype = comm_dtype or dp.get("comm_dtype") or os.environ.get("DPFT_COMM_DTYPE") or "fp32"
        wire = {"fp32": None, "float32": None, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}[comm_dtype]
        self.bucket_mb, self.comm_dtype = bucket_mb, comm_dtype
        self.reducer = GradBucketReducer(list(self.model.parameters()), bucket_bytes=int(bucket_mb * (1 << 20)),
                                         group_of=group_of, comm_dtype=wire, force_collectives=force_collectives)
        self.collective = self.reducer.collective
        # one-rank timing experiment (tools/exp_switches.py): take the step decision from the local loss although collectives are
        # on.  Never read from the environment; refused with more than one rank (ranks disagreeing on the decision mismatch
        # their collectives and hang).
        self._exp_local_decision = False
        # The collectives' hardware queue.  Four queues, four busy chains (main | camera weight gradients | two radar views).
        # "pg" (default since round 5) = the process group's own stream, "side" = in order on the camera's weight-gradient
        # stream (the chain with slack and the producer of most camera buckets; the round-3 default, chosen with no-op
        # one-rank collectives), "own" = a dedicated stream, "front" = the last view's stream.  Round 5 put a kernel with an
        # 8-rank ring all-reduce's footprint behind every bucket collective (32 workgroups, the bucket read and written twice:
        # tools/r05_comm_standin.sh, profiles/r05_comm_standin.txt): median step 26.1 / 26.1 ms and 0.09 ms exposed with "pg",
        # 26.0 / 27.6 ms and 0.29 ms exposed with "side" (its stream is joined by the main chain at every stage boundary, so a
        # slow collective there holds the data-gradient chain), 26.3 / 27.4 ms with "own"; plain step 25.4-25.6 ms.  Still one
        # rank: tools/scale.sh sweeps the three on a multi-GPU node.
        self.comm_placement = os.environ.get("DPFT_COMM_STREAM", "pg")
        self.optimizer = build_optimizer(name, self.model.parameters(), device=device, **opt)   # trainer.py:233
        # epoch loop (trainer.py:21-47,64-67): epochs, schedule, logging frequency (None | 'step' | 'epoch')
        self.epochs = int(train.get("epochs", 1))
        self.logging = train.get("logging")
        sched = dict(train.get("scheduler") or {"name": "ConstantLR", "factor": 1.0})
        from dpft_amd.training.scheduler import build_scheduler
        self.scheduler = build_scheduler(sched.pop("name"), **sched)(self.optimizer)             # trainer.py:236
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        overwritten = []
        for m in self.model.modules():
            if hasattr(m, "grad_direct"):
                m.grad_direct = self.reducer
                # the producer names exactly the parameters whose bucket views its backward OVERWRITES every step
                overwritten += list(m.overwritten_parameters())
        if self.device.type == "cuda" and os.environ.get("DPFT_ZERO_ALL_GRADS", "0") != "1":
            self.reducer.set_overwritten(overwritten)
        # Round 4 (opt-in, DPFT_EARLY_ADAMW=1): the optimizer steps a bucket's parameters as soon as the bucket's gradients
        # are final (behind its all-reduce on the communication stream; with one rank on the camera's weight-gradient
        # stream) instead of in one 0.44 ms launch behind the whole backward.  Same arithmetic, same per-parameter step
        # counts; buckets that are not complete when finish() runs are stepped by optimizer.step().  Measured on one rank:
        # the launch disappears from the end of the step and the backward grows by as much (27.33 vs 27.31-27.39 ms;
        # tools/step_timeline.py: optimizer 0.46 -> 0.01 ms, backward 18.8 -> 19.2 ms) -- the step is bound by total kernel
        # time, moving work between queues buys nothing.  Kept for N > 1, where the last buckets' updates can hide behind
        # the exposed tail of the exchange; off by default.
        self.early_adamw = (isinstance(self.optimizer, FusedAdamW) and self.device.type == "cuda"
                            and os.environ.get("DPFT_EARLY_ADAMW", "0") == "1")
        if self.early_adamw:
            self.optimizer.attach_segments([b["params"] for b in self.reducer.buckets])
            self.reducer.on_bucket_final = self.optimizer.step_segment

    def enable_graphs(self, sample_data: Dict[str, torch.Tensor]):
        """Replay the launch-bound decoder from hipGraphs (static shapes of ``sample_data``)."""
        if self.collective:                   # no collective (parameter broadcast) may still be in flight during a capture
            dist.barrier()
        torch.cuda.synchronize()
        self.reducer.reset()
        self.model.enable_fuser_graph(sample_data, grad_direct=self.reducer)
        self.optimizer.zero_grad(set_to_none=False)

    def train_step(self, data: Dict[str, torch.Tensor], labels: List[Dict[str, torch.Tensor]], with_metrics: bool = False):
        """One step of CentralizedTrainer.train_one_epoch (trainer.py:122-136).  ``with_metrics`` also evaluates the
        configured detection metrics on the step's outputs (two more launches) and returns them as a third value."""
        if not self.model.training:                        # (Module.train() walks ~1 000 modules: 2-4 ms of host time per call;
            self.model.train()                             # the epoch loop calls it once per epoch like trainer.py:113)
        if self.collective and self.reducer.comm_stream is None and self.comm_placement != "pg" and self.device.type == "cuda" \
                and dist.get_backend() == "nccl":
            if self.comm_placement == "own":
                # a stream of its own: never joined by a backward stage, unlike the weight-gradient stream; the
                # runtime maps it onto one of the 4 hardware queues, where it is in order with whatever else sits there
                self.reducer.comm_stream = torch.cuda.Stream(self.device)
            else:
                views = self.model.__dict__.get("_view_streams")
                if views:
                    first = self.model.backbones[self.model.inputs[0]]
                    self.reducer.comm_stream = views[-1] if self.comm_placement == "front" or first.side_stream is None \
                        else first.side_stream
        self.reducer.reset()                               # zero_grad (grads live in the buckets)
        if self.early_adamw and not self.collective and self.reducer.opt_stream is None:
            first = self.model.backbones[self.model.inputs[0]] if hasattr(self.model, "backbones") else None
            side = getattr(first, "side_stream", None)
            if side is not None:                           # (placed by the first multi-view forward: the second step on)
                self.reducer.opt_stream = side
                self.reducer.opt_from = torch.cuda.current_stream(self.device).cuda_stream
        g = self.model.__dict__.get("_graphed_fuser")
        if g is not None:
            g.clone_outputs = False                        # loss, metrics and the backward below are done with them in time
            if self.pace_host != "0" and g.pace_event is None:
                g.pace_event = torch.cuda.Event(enable_timing=self.pace_host == "auto")
        try:
            output = self.model(data)
        finally:
            if g is not None:
                g.clone_outputs = True
        if g is not None and g.last_inputs is not None and os.environ.get("DPFT_MANUAL_CHAIN", "1") != "0" \
                and hasattr(self.loss_fn, "backward_into"):
            # the buffers the decoder's backward graph reads its output gradients from: the loss call launches the criterion's
            # gradient kernel into them itself (one C call for the whole host window, Loss.forward_fused)
            go = g.static_grad_outputs
            self.loss_fn.__dict__["fused_grad_targets"] = (go[0], go[1], go[2], go[3])
        elif hasattr(self.loss_fn, "__dict__"):
            self.loss_fn.__dict__["fused_grad_targets"] = None
        loss, losses = self.loss_fn(output, labels)
        if g is not None and g.pace_event is not None and self.pace_host != "0":
            self._pace(g)
        gate = None
        if self._exp_local_decision and self.world > 1:
            raise RuntimeError("_exp_local_decision is a one-rank timing experiment switch")
        known = getattr(self.loss_fn, "last_has_targets", None)
        if self.collective and not self._exp_local_decision:
            # The global batch steps if ANY shard has a loss (MAX): a rank whose label shard is empty then runs the same
            # backward over a zero-valued loss, so it contributes zero gradients, issues its bucket collectives in the
            # same order and reports the same set of parameters-with-gradient as every other rank.
            if known is not None and isinstance(self.optimizer, FusedAdamW) and self.sync_free_decision:
                # No read-back: every rank ALWAYS runs its backward (over a zero-valued loss where its own shard has no target:
                # known on the host from the label dicts) and the any-rank flag stays on the device as the optimizer's gate --
                # in the rare step in which NO rank has a target the backward is wasted work and the gate keeps every
                # parameter and moment as it was, which is what skipping the step does.
                any_rank = (loss.detach() > 0).to(torch.float32).reshape(1)
                dist.all_reduce(any_rank, op=dist.ReduceOp.MAX)
                local, stepped, gate = bool(known), True, any_rank
            else:
                # [local, any-rank] are read back together: ONE host sync per step (trainer.py:131 has one).
                flag = (loss.detach() > 0).to(torch.int32).reshape(1).repeat(2)
                dist.all_reduce(flag[1:], op=dist.ReduceOp.MAX)
                local, stepped = (bool(v) for v in flag.tolist())
        else:
            # trainer.py:131 `if loss > 0`: the loss is exactly 0 when the batch has no target and positive otherwise (the focal term
            # alone) -- the host knows which from the label dicts, so it launches backward and optimizer WITHOUT reading the loss
            # back (the read-back was the second host sync inside the window in which the GPU waits for the host); the comparison
            # itself is done on the device by the optimizer launch (FusedAdamW.set_gate: nothing is updated unless loss > 0).
            if known is not None and isinstance(self.optimizer, FusedAdamW):
                stepped = local = bool(known)
                gate = loss.detach()
            else:
                stepped = local = float(loss.detach()) > 0          # (eager loss / torch optimizer: compared on the host)
        if stepped:
            if isinstance(self.optimizer, FusedAdamW):
                # BEFORE the backward: with DPFT_EARLY_ADAMW the optimizer steps buckets DURING it (on_bucket_final ->
                # step_segment); a gate installed only afterwards would leave those launches ungated -- a NaN / zero loss
                # would then update the early buckets and not the rest.  step() consumes the gate.
                self.optimizer.gate_required = gate is not None
                self.optimizer.set_gate(gate)
            if not local:
                loss = sum(v.sum() for v in output.values()) * 0.0
            if not (local and self._backward_without_engine(loss)):
                loss.backward()
            self.reducer.finish()
            if isinstance(self.optimizer, FusedAdamW):
                self.optimizer.set_active(self.reducer.seen_ids())
            self.optimizer.step()
        if hasattr(self.loss_fn, "__dict__"):
            # validation calls the same loss_fn: without this every eval batch would launch the criterion's gradient kernel
            # into the decoder graph's static gradient buffers
            self.loss_fn.__dict__["fused_grad_targets"] = None
        if with_metrics and self.eval_fn is not None:
            return loss.detach(), {k: v.detach() for k, v in losses.items()}, self.eval_fn(output, labels)
        return loss.detach(), {k: v.detach() for k, v in losses.items()}

    def _backward_without_engine(self, loss: torch.Tensor) -> bool:
        """Replayed decoder + fused loss: their two backward steps are launched directly (GraphedFuser.backward_from),
        autograd starts behind them.  False = the plain ``loss.backward()`` has to run."""
        g = self.model.__dict__.get("_graphed_fuser")
        if g is None or g.last_inputs is None or os.environ.get("DPFT_MANUAL_CHAIN", "1") == "0":
            return False
        if not hasattr(self.loss_fn, "backward_into") or self.loss_fn.__dict__.get("_last") is None:
            return False
        done = []

        def write(go):          # static_grad_outputs = (center, size, angle, class)
            done.append(self.loss_fn.backward_into(loss, go[0], go[1], go[2], go[3]))
        if self.loss_fn.__dict__["_last"][-1] is not loss:
            return False
        g.backward_from(write)
        return bool(done and done[0])

    # ------------------------------------------------------------------------------------------ epoch loop
    @classmethod
    def from_config(cls, model: torch.nn.Module, config: Dict[str, Any], **kwargs) -> "DataParallelTrainer":
        """``CentralizedTrainer.from_config`` (trainer.py:49-80) plus the module to shard: device from
        ``computing.device`` (rank-local: ``cuda`` -> ``cuda:LOCAL_RANK``)."""
        device = torch.device(config["computing"]["device"])
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
        return cls(model, config, device, **kwargs)

    def _dict_to(self, data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        return {k: (v.to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in data.items()}

    # Pacing of the host (train_step -> _pace): "auto" (default) | "1" on | "0" off  (DPFT_PACE_HOST).
    pace_host = os.environ.get("DPFT_PACE_HOST", "auto")
    pace_ratio = 0.8               # auto: pace when the host's own step period is below this fraction of the GPU's

    def _pace(self, g) -> None:
        """Pacing, not a data dependency.  With the assignments on the device nothing in the step makes the host wait.  On a
        GPU-bound step a host that is never held back costs 0-0.4 ms per step (kradar.json at batch 4: 24.3-24.9 ms unpaced,
        24.0-24.4 paced; bf16 at batch 8: 23.7 vs 23.3 -- profiles/r05_loss_window_ab.txt): it enqueues the whole backward while the
        GPU is still in the encoders' forward.  Paced, the host waits until the GPU has REACHED the decoder's forward graph (an
        event in front of it), with the matcher / assignment / criterion launches already queued behind that graph: the GPU
        never idles (the loss window stays ~50 us) and the backward is enqueued into nearly empty queues.  On a step whose launch
        work takes the host about as long as the kernels take the GPU (bf16 at batch 4: host 13-18 ms, GPU 15 ms) the same wait
        removes the lead that absorbs the host's jitter and costs 1-6 ms (15.3 -> 18.7-22.7 ms).  "auto" tells the two apart from
        four unpaced steps: the host's period between two pace points (it runs free: its own launch work) against the GPU's
        period between the two events; the observed ratios are 0.35-0.73 where pacing gains (fp32 batch 4 / 8, bf16 batch 8) and
        0.97-1.0 where it loses (bf16 batch 4): paced below pace_ratio = 0.8."""
        if self.pace_host == "1":
            g.pace_event.synchronize()
            return
        st = self.__dict__.setdefault("_pace_state", {"events": [], "host": [], "t": None, "decided": None})
        if st["decided"] is not None:
            if st["decided"]:
                g.pace_event.synchronize()
            return
        import time
        now = time.perf_counter()
        if st["t"] is not None:
            st["host"].append(now - st["t"])
        st["t"] = now
        st["events"].append(g.pace_event)                  # this step's (recorded) event; the next step records a fresh one
        if len(st["events"]) < 5:
            g.pace_event = torch.cuda.Event(enable_timing=True)
            return
        ev = st["events"]
        ev[-1].synchronize()                               # (once: the measurement needs the last event's time stamp)
        try:
            gpu_ms = min(ev[i].elapsed_time(ev[i + 1]) for i in range(1, 4))
            host_ms = min(st["host"][1:]) * 1e3
            st.update(decided=host_ms < self.pace_ratio * gpu_ms, host_ms=host_ms, gpu_ms=gpu_ms, events=[])
        except RuntimeError:                               # (an event of a step that did not replay the decoder graph)
            st.update(decided=False, host_ms=None, gpu_ms=None, events=[])
        g.pace_event = torch.cuda.Event()

    sync_free_decision = True      # multi-rank step decision without a read-back (train_step); False = the round-4 form

    def _check_matcher(self) -> None:
        """Deferred error of the on-device assignment (Loss.check_assignment_status): called where values are read back anyway."""
        chk = getattr(self.loss_fn, "check_assignment_status", None)
        if chk is not None and self.device.type == "cuda":
            chk(sync=True)

    def _rank_mean(self, scalars: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Mean over ranks of every scalar (ONE collective).  Every rank calls it with the same keys -- checked: a rank whose
        loader yielded nothing (tiny validation split, drop_last) would otherwise skip the collective the others enter and
        hang the job."""
        # the on-device matcher's deferred error rides in the same collective: a rank that raised alone would leave the others
        # waiting in the next all-reduce -- the MAX of the status words makes every rank raise here, together
        status_of = getattr(self.loss_fn, "assignment_status", None)
        code = status_of(sync=True) if (status_of is not None and self.device.type == "cuda") else 0
        if self.world > 1:
            n = torch.tensor([len(scalars), -len(scalars), code], dtype=torch.int64, device=self.device)
            dist.all_reduce(n, op=dist.ReduceOp.MAX)
            if int(n[0]) != -int(n[1]):
                raise RuntimeError(f"rank {self.rank}: {len(scalars)} epoch scalars, another rank has {int(n[0])} / {-int(n[1])}: "
                                   "a rank saw no batches (sampler / drop_last leave it an empty shard)")
            code = int(n[2])
        if code:
            raise ValueError(self.loss_fn.describe_assignment_status(code) + (" (on at least one rank)" if self.world > 1 else ""))
        if not scalars:
            return scalars
        keys = sorted(scalars)
        vec = torch.stack([torch.as_tensor(scalars[k], dtype=torch.float32, device=self.device).reshape(()) for k in keys])
        if self.world > 1:
            dist.all_reduce(vec, op=dist.ReduceOp.SUM)
            vec = vec / self.world
        return dict(zip(keys, vec.unbind()))

    def _log(self, writer, scalars: Dict[str, Any], step: int, prefix: str):
        if writer is not None:
            for name, v in scalars.items():
                writer.add_scalar(f"{prefix}/{name}", float(v), step)

    def train_one_epoch(self, epoch: int, data_loader: Iterable, writer=None) -> Dict[str, float]:
        """trainer.py:99-160 on this rank's shard: per step zero_grad -> forward -> loss -> (backward, all-reduce,
        optimizer) -> metrics.  Per-step values stay on the device; they are summed there and read back once per epoch
        (``logging == 'step'`` reads them every step, as the reference's writer does)."""
        self.model.train()
        self.loss_fn.train()
        sums: Dict[str, torch.Tensor] = {}
        n = 0
        for i, (data, labels) in enumerate(data_loader):
            step = i + epoch * len(data_loader)
            if self.logging == "step" and writer is not None:
                writer.add_scalar("train/learning_rate", self.optimizer.param_groups[0]["lr"], step)
            labels = [self._dict_to(l) for l in labels]
            data = self._dict_to(data)
            res = self.train_step(data, labels, with_metrics=True)
            loss, losses = res[0], res[1]
            metrics = res[2] if len(res) > 2 else {}
            scalars = {f"loss_{k}": v for k, v in losses.items()}
            scalars["loss"] = loss
            scalars.update(metrics)
            if self.logging == "step":
                self._log(writer if self.rank == 0 else None, self._rank_mean(scalars), step, "train")
            for k, v in scalars.items():
                v = torch.as_tensor(v, device=self.device).detach().float().reshape(())
                sums[k] = sums[k] + v if k in sums else v.clone()
            n += 1
        means = self._rank_mean({k: v / max(n, 1) for k, v in sums.items()})
        if self.logging == "epoch" and self.rank == 0 and writer is not None:
            self._log(writer, means, epoch, "train")
            writer.add_scalar("train/learning_rate", self.optimizer.param_groups[0]["lr"], epoch)
        return {k: float(v) for k, v in means.items()}

    @torch.no_grad()
    def validate_one_epoch(self, epoch: int, data_loader: Iterable, writer=None) -> Dict[str, float]:
        """trainer.py:162-213: eval-mode forward (the fused inference decoder), loss and metrics on this rank's shard,
        averaged over steps and ranks.  Returns ``{'loss': ...}`` like the reference (plus the other means)."""
        self.model.eval()
        self.loss_fn.eval()
        sums: Dict[str, torch.Tensor] = {}
        n = 0
        for i, (data, labels) in enumerate(data_loader):
            labels = [self._dict_to(l) for l in labels]
            data = self._dict_to(data)
            output = self.model(data)
            loss, losses = self.loss_fn(output, labels)
            metrics = self.eval_fn(output, labels) if self.eval_fn is not None else {}
            scalars = {f"loss_{k}": v for k, v in losses.items()}
            scalars["loss"] = loss
            scalars.update(metrics)
            if self.logging == "step":
                self._log(writer if self.rank == 0 else None, self._rank_mean(scalars), i + epoch * len(data_loader), "val")
            for k, v in scalars.items():
                v = torch.as_tensor(v, device=self.device).detach().float().reshape(())
                sums[k] = sums[k] + v if k in sums else v.clone()
            n += 1
        means = self._rank_mean({k: v / max(n, 1) for k, v in sums.items()})
        if self.logging == "epoch" and self.rank == 0 and writer is not None:
            self._log(writer, means, epoch, "val")
        return {k: float(v) for k, v in means.items()}

    def save_checkpoint(self, path: str) -> None:
        """``torch.save(model, path)`` (trainer.py:256-258) by rank 0; every rank returns after the file exists.  The
        module's process-local state (streams, graphs, plans, reducer links: ``__getstate__`` of DPRT / IMPFusion /
        BackboneBase / FPN) is not part of the pickle, so the live model keeps training afterwards."""
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
            chk = getattr(self.loss_fn, "check_assignment_status", None)
            if chk is not None:
                chk(sync=False)      # (a step with a non-finite cost matrix must not end up in a checkpoint unnoticed)
        if self.rank == 0:
            tmp = path + ".tmp"
            torch.save(self.model, tmp)
            os.replace(tmp, path)
        if self.world > 1:
            dist.barrier()

    def train(self, data_loader: Iterable, val_loader: Iterable = None, start_epoch: int = 0, timestamp: str = None,
              dst: str = None, sampler=None, continue_schedule: bool = False) -> List[str]:
        """trainer.py:215-263.  ``sampler``: an optional ``ShardedSampler`` whose ``set_epoch`` is called per epoch.
        Resuming (``start_epoch`` > 0): like the reference (trainer.py:233-239 builds a FRESH optimizer and scheduler for a
        resumed run) the learning-rate schedule starts over; ``continue_schedule=True`` fast-forwards it to where the
        interrupted run stopped instead.  Returns the checkpoint paths written."""
        if timestamp is None:
            timestamp = datetime.datetime.now().strftime("%Y%m%d-%H%M%S-%f")[:-3]
            if self.world > 1:                       # one directory for the job: rank 0's clock
                box = [timestamp]
                dist.broadcast_object_list(box, src=0)
                timestamp = box[0]
        assert dst is not None, "train() needs a destination directory for checkpoints"
        ckpt_dir = osp.join(dst, timestamp, "checkpoints")
        writer = None
        if self.rank == 0:
            os.makedirs(ckpt_dir, exist_ok=True)
            if self.logging is not None:
                writer = _ScalarLog(osp.join(dst, timestamp))
        if continue_schedule:
            for _ in range(start_epoch):             # opt-in: continue the schedule where the interrupted run stopped
                self.scheduler.step()
        written = []
        for epoch in range(start_epoch, self.epochs):
            if sampler is not None:
                sampler.set_epoch(epoch)
            self.last_train = self.train_one_epoch(epoch, data_loader, writer)
            if val_loader is not None:
                self.last_val = self.validate_one_epoch(epoch, val_loader, writer)
            self.scheduler.step()
            path = osp.join(ckpt_dir, f"{timestamp}_checkpoint_{str(epoch).zfill(4)}.pt")
            self.save_checkpoint(path)
            written.append(path)
        if writer is not None:
            writer.close()
        return written

    __call__ = train

    @torch.no_grad()
    def inference_time(self, data: Dict[str, torch.Tensor], warmup: int = 10, reps: int = 300):
        """mean / std forward latency in ms on one batch (evaluator.py:109-125)."""
        self.model.eval()
        for _ in range(warmup):
            self.model(data)
        starter, ender = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        times = []
        for _ in range(reps):
            starter.record()
            self.model(data)
            ender.record()
            torch.cuda.synchronize()
            times.append(starter.elapsed_time(ender))
        t = torch.tensor(times)
        return float(t.mean()), float(t.std())
