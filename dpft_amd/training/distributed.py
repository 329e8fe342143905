"""Data-parallel gradient reduction over RCCL (backend "nccl" on ROCm) / gloo.

The reference has no distributed code at all (single ``CentralizedTrainer``,
src/dprt/training/trainer.py:20,215); this is the MI355X-side addition: one process
per GPU, samples sharded across ranks, one exchange step = all-reduce(mean) of the gradients.

Design for xGMI (7 point-to-point links per GPU, ring collectives are per-link bound):
  * gradients live in flat fp32 buckets (default 25 MiB, SURVEY 5: large enough to run the links at their
    bandwidth, small enough that the first collective starts early in the backward; ``train.dp.bucket_mb`` /
    DPFT_BUCKET_MB) in REVERSE registration order, all carved out of ONE arena (one memset per step);
    ``param.grad`` are views into them -> one collective per bucket, no per-tensor launches;
  * optional bf16 wire format (``train.dp.comm_dtype = "bf16"`` / DPFT_COMM_DTYPE): a bucket is rounded into a bf16
    staging buffer, reduced in bf16 and widened back - half the bytes per link (179.8 MB instead of 359.6 MB per step);
  * ``exposed_ms()``: the part of the exchange that was NOT hidden behind the backward (time between the end of the
    rank's own backward work and the completion of the last collective);
  * a bucket's all-reduce is issued asynchronously the moment its last gradient is produced, so the
    exchange of layer4 overlaps the backward of layer3 ... ; the hand-scheduled backbone backward
    hands gradients over per bottleneck block through ``grad_sink`` instead of waiting for the end
    of its (single) autograd node;
  * parameters that never receive a gradient (the un-cloned template head) are
    tolerated: unfinished buckets are flushed in ``finish()`` with zeros for the missing grads.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist



class GradBucketReducer:
    # one-rank timing experiment (tools/exp_switches.py flips it; never read from the environment): buckets fire without their
    # collective -- no gradient exchange, so it is refused with more than one rank
    exp_skip_bucket_collectives = False
    # one-rank cost-model experiment (tools/exp_switches.py --standin-collective): called with the wire tensor right behind each
    # bucket's collective, on the stream the collective was enqueued on -- a stand-in for the CU / HBM footprint an N > 1
    # all-reduce kernel would have there.  None in the product.
    exp_after_collective = None

    def __init__(self, params: List[torch.nn.Parameter], bucket_bytes: int = 25 << 20, process_group=None,
                 average: bool = True, group_of: Dict[int, str] = None, comm_dtype: Optional[torch.dtype] = None,
                 force_collectives: bool = False):
        """``group_of`` (id(param) -> group key) keeps buckets from spanning groups: the three view encoders run
        (forward and backward) on their own HIP streams, and a bucket whose gradients all come from one stream
        can be reduced from that stream without joining the others.  ``force_collectives``: issue every bucket's
        all-reduce even in a one-rank group (the N>1 code path -- stream joins, RCCL's own stream, wire staging --
        exercised on a single GPU; the reduction over one rank is the identity)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.collective = self.world > 1 or (bool(force_collectives) and dist.is_initialized())
        if self.world > 1:
            # the one-rank timing switches of the round-3 forced-collectives breakdown, docs/history would let ranks diverge (no gradient exchange) or
            # hang (ranks disagreeing on the step decision mismatch their collectives): refused outside one-rank runs
            if GradBucketReducer.exp_skip_bucket_collectives:
                raise RuntimeError(f"exp_skip_bucket_collectives: one-rank timing experiment switches are not allowed with "
                                   f"world_size {self.world}")
        # RCCL averages inside the collective (ncclAvg): no separate division pass over the 360 MB of buckets
        self._avg_op = average and dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        # One rank (bench.py --force-collectives, tests): the average over one rank IS the sum, and RCCL implements a one-rank
        # AVG as a pre-multiply kernel over the whole bucket (oneRankReduce<FuncPreMulSum>: 20 launches, 0.5 ms of memory
        # passes per step that the all-reduce kernel of N > 1 ranks does not add) but a one-rank SUM as nothing -- so SUM
        # there.  DPFT_COLLECTIVE_OP=avg|sum forces either (A/B, the round-3 forced-collectives breakdown, docs/history).
        self.collective_op = "avg" if self._avg_op else "sum"
        forced_op = __import__("os").environ.get("DPFT_COLLECTIVE_OP")
        if self._avg_op and (forced_op == "sum" or (forced_op is None and self.world == 1)):
            self._avg_op = False
            self.collective_op = "sum"
        self.comm_stream = None      # optional torch stream the collectives are enqueued on (set by the trainer)
        # Round 4: called as on_bucket_final(bucket index) on the stream where a bucket's FINAL gradients are available
        # (behind its collective, or -- one rank -- behind its producers): the trainer steps the optimizer for that
        # bucket's parameters there, so the update overlaps the rest of the backward.  `opt_stream` (one rank, optional):
        # the stream to do that on instead of the firing stream (the camera's weight-gradient stream: off the critical chain)
        self.on_bucket_final = None
        self.opt_stream = None
        self.opt_from = None         # raw handle of the stream whose buckets move to opt_stream
        self._opt_streams = {}       # streams that carried such updates since reset(): joined by finish()
        self.average = average
        self.comm_dtype = comm_dtype if comm_dtype not in (None, torch.float32) else None
        self.params = [p for p in params if p.requires_grad]
        order = list(reversed(self.params))
        self.buckets: List[dict] = []
        plan, cur, cur_bytes, cur_group = [], [], 0, None
        for p in order:
            nbytes = p.numel() * p.element_size()
            grp = group_of.get(id(p)) if group_of else None
            if cur and (cur_bytes + nbytes > bucket_bytes or grp != cur_group):
                plan.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
            cur_group = grp
        if cur:
            plan.append(cur)
        # one arena for all buckets (64-element aligned starts): reset() is a single memset
        starts, total = [], 0
        for ps in plan:
            starts.append(total)
            total += (sum(p.numel() for p in ps) + 63) // 64 * 64
        p0 = self.params[0]
        self.arena = torch.zeros(total, dtype=p0.dtype, device=p0.device)
        for ps, st in zip(plan, starts):
            self._add_bucket(ps, self.arena[st:st + sum(p.numel() for p in ps)])
        self._exposed = None
        self._overwritten = set()                # ids of parameters reset() does not clear (set_overwritten)
        self._zero_spans = [(0, total)]          # arena spans reset() clears (see set_overwritten)
        self._spans = {id(p): (st + off, st + off + p.numel())
                       for ps, st in zip(plan, starts)
                       for p, off in zip(ps, [sum(q.numel() for q in ps[:i]) for i in range(len(ps))])}
        self._index: Dict[int, tuple] = {}
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._index[id(p)] = bi
        self._bucket_index = {id(b["flat"]): bi for bi, b in enumerate(self.buckets)}
        self._in_finish = False
        self._trace = None               # trace_begin(): per-bucket ready / start / end marks of the next step's exchange
        self._hooks = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]
        self._pending: List = []
        self.reset()

    def _add_bucket(self, params, flat):
        views, off = {}, 0
        for p in params:
            seg = flat[off:off + p.numel()]
            # keep the parameter's physical layout (conv weights are channels_last) so that copies are linear
            if p.dim() == 4 and p.permute(0, 2, 3, 1).is_contiguous():
                K, C, kh, kw = p.shape
                v = seg.view(K, kh, kw, C).permute(0, 3, 1, 2)
            else:
                v = seg.view(p.shape)
            views[id(p)] = v
            off += p.numel()
        stage = torch.empty(flat.numel(), dtype=self.comm_dtype, device=flat.device) if self.comm_dtype else None
        self.buckets.append(dict(params=params, flat=flat, views=views, ready=0, fired=False, seen=set(), streams={},
                                 stage=stage))

    # ------------------------------------------------------------------------------------------
    def set_overwritten(self, params) -> None:
        """Parameters whose gradient producer OVERWRITES the bucket view every step (the native backbone plans, the FPN's
        direct hand-over) need no clearing: reset() then zeroes only the spans of the others (the decoder's gradients are
        added into the buckets, parameters without a gradient must read as zero) -- ~5 MB instead of 360 MB per step."""
        params = list(params)
        self._overwritten = {id(p) for p in params if id(p) in self._spans}
        skip = sorted(self._spans[id(p)] for p in params if id(p) in self._spans)
        spans, cur = [], 0
        for a, b in skip:
            if a > cur:
                spans.append((cur, a))
            cur = max(cur, b)
        if cur < self.arena.numel():
            spans.append((cur, self.arena.numel()))
        # merge spans separated by small gaps (alignment padding) and keep the launch count low
        merged = []
        for a, b in spans:
            if merged and a - merged[-1][1] <= 4096:
                merged[-1] = (merged[-1][0], b)
            else:
                merged.append((a, b))
        self._zero_spans = merged

    def clear(self, params) -> None:
        """Zero the bucket views of these parameters now (a producer that was announced as overwriting falls back to
        accumulation for this step)."""
        for p in params:
            bi = self._index.get(id(p))
            if bi is not None:
                self.buckets[bi]["views"][id(p)].zero_()

    def reset(self):
        """Call before every backward: zero the buckets and point ``param.grad`` at the bucket views."""
        if self.arena.is_cuda:      # all spans in one launch (dpft_memops)
            from dpft_amd.hip import ops
            ops.memops([(self.arena[a:b], None) for a, b in self._zero_spans])
        elif len(self._zero_spans) == 1 and self._zero_spans[0] == (0, self.arena.numel()):
            self.arena.zero_()
        else:
            for a, b in self._zero_spans:
                self.arena[a:b].zero_()
        for b in self.buckets:
            b["ready"], b["fired"] = 0, False
            b["seen"].clear()
            b["streams"].clear()
            for p in b["params"]:
                p.grad = b["views"][id(p)]
        self._pending = []
        self._opt_streams = {}

    def grad_sink(self, p: torch.nn.Parameter, g: torch.Tensor):
        """Direct hand-over from a hand-scheduled backward (bypasses autograd accumulation)."""
        bi = self._index.get(id(p))
        if bi is None:
            return False
        self.buckets[bi]["views"][id(p)].add_(g)
        self._mark(bi, p)
        return True

    def grad_buffer(self, p: torch.nn.Parameter):
        """The bucket view a native backward writes this parameter's gradient into (None if unknown).
        Conv weights get a view that is physically [K][kh][kw][C] contiguous."""
        bi = self._index.get(id(p))
        return None if bi is None else self.buckets[bi]["views"][id(p)]

    def mark_ready(self, p: torch.nn.Parameter):
        bi = self._index.get(id(p))
        if bi is not None:
            self._mark(bi, p)

    def mark_ready_many(self, params):
        """``mark_ready`` for a whole list at once (a replayed decoder graph hands over ~250 gradients): the per-bucket
        bookkeeping and the current-stream lookup happen once per bucket instead of once per parameter -- this runs on
        the host right after the step's only device sync, while the GPU has nothing queued."""
        touched = {}
        for p in params:
            bi = self._index.get(id(p))
            if bi is None:
                continue
            b = self.buckets[bi]
            if id(p) in b["seen"]:
                continue
            b["seen"].add(id(p))
            b["ready"] += 1
            touched[bi] = b
        for b in touched.values():
            if b["flat"].is_cuda:
                st = torch.cuda.current_stream(b["flat"].device)
                b["streams"][st.cuda_stream] = st
            if b["ready"] == len(b["params"]) and not b["fired"]:
                self._fire(b)

    def _hook(self, p):
        bi = self._index[id(p)]
        b = self.buckets[bi]
        if p.grad is not b["views"][id(p)]:                  # autograd replaced the view: copy back
            b["views"][id(p)].copy_(p.grad)
            p.grad = b["views"][id(p)]
        self._mark(bi, p)

    def _mark(self, bi: int, p):
        b = self.buckets[bi]
        if id(p) in b["seen"]:
            return
        b["seen"].add(id(p))
        b["ready"] += 1
        if b["flat"].is_cuda:
            # the view encoders (and therefore their backward) run on separate HIP streams: remember which
            # streams produced gradients of this bucket so that its collective is ordered after all of them
            st = torch.cuda.current_stream(b["flat"].device)
            b["streams"][st.cuda_stream] = st
        if b["ready"] == len(b["params"]) and not b["fired"]:
            self._fire(b)

    def _final(self, b, st=None):
        """The bucket's gradients are final on stream ``st`` (None: the current one)."""
        if self.on_bucket_final is None or self._in_finish:
            return
        bi = self._bucket_index[id(b["flat"])]
        if st is None or not b["flat"].is_cuda:
            self.on_bucket_final(bi)
            return
        with torch.cuda.stream(st):
            self.on_bucket_final(bi)
        self._opt_streams[st.cuda_stream] = st

    def _fire(self, b):
        b["fired"] = True
        comm = self.comm_stream if b["flat"].is_cuda else None
        if not self.collective:
            # one rank, no exchange: the gradients are final as soon as every producer stream has passed this point
            if self.on_bucket_final is not None and not self._in_finish and b["flat"].is_cuda:
                cur = torch.cuda.current_stream(b["flat"].device)
                # (only buckets finished on `opt_from` -- the main stream, the critical chain -- move; a view's own stream
                # is idle once its backward is through)
                tgt = self.opt_stream if (self.opt_stream is not None and cur.cuda_stream == self.opt_from) else cur
                for sid, st in b["streams"].items():
                    if sid != tgt.cuda_stream:
                        tgt.wait_stream(st)
                if cur.cuda_stream != tgt.cuda_stream and cur.cuda_stream not in b["streams"]:
                    tgt.wait_stream(cur)
                self._final(b, tgt)
            return
        if b["flat"].is_cuda and self.collective and comm is None:     # single process: finish() joins the streams
            # The process group's stream orders itself behind the CURRENT stream only, so the current stream has to wait for
            # the other producers of this bucket.  That stalls a compute chain at every bucket boundary (the main stream
            # waits for the weight-gradient stream it is supposed to run ahead of) -- the reason a comm stream of our own
            # is the default on the GPU: there only the communication stream waits (below).
            cur = torch.cuda.current_stream(b["flat"].device)
            for sid, st in b["streams"].items():
                if sid != cur.cuda_stream:          # tail of that stream is after its last contribution
                    cur.wait_stream(st)
        if self.collective and not GradBucketReducer.exp_skip_bucket_collectives:
            op = dist.ReduceOp.SUM
            if self._avg_op:
                op = dist.ReduceOp.AVG
            if comm is None:
                # the process group's own stream (async_op=True): wherever the runtime put it among the hardware queues
                if op == dist.ReduceOp.SUM and self.average and self.world > 1:
                    b["flat"].div_(self.world)
                wire = b["flat"]
                if b["stage"] is not None:
                    wire = b["stage"]
                    wire.copy_(b["flat"])                             # round to the wire dtype (RNE)
                ready = self._trace_mark() if self._trace is not None else None
                work = dist.all_reduce(wire, op=op, group=self.group, async_op=True)
                self._pending.append((work, b))
                if ready is not None:
                    tr = self._trace
                    if tr["cuda"]:
                        with torch.cuda.stream(tr["helper"]):      # the helper stream waits for the collective, nothing else does
                            work.wait()
                            end = self._trace_mark(tr["helper"])
                    else:
                        work.wait()
                        end = self._trace_mark()
                    tr["rows"].append((self._bucket_index[id(b["flat"])], wire.numel() * wire.element_size(), len(b["params"]), ready, ready, end))
                if GradBucketReducer.exp_after_collective is not None:
                    GradBucketReducer.exp_after_collective(wire)
            else:
                # a stream of OUR choosing (= a hardware queue of our choosing, DataParallelTrainer): a synchronous
                # collective is enqueued on the current stream, so the exchange of a bucket -- scaling, wire rounding,
                # all-reduce, widening -- is one in-order sequence there, fenced by two events
                # the COMMUNICATION stream waits for every producer of the bucket; no compute stream waits for another
                cur = torch.cuda.current_stream(b["flat"].device)
                ready = self._trace_mark(cur) if self._trace is not None else None
                if cur.cuda_stream != comm.cuda_stream:
                    comm.wait_stream(cur)
                for sid, st in b["streams"].items():
                    if sid != comm.cuda_stream and sid != cur.cuda_stream:
                        comm.wait_stream(st)
                with torch.cuda.stream(comm):
                    if op == dist.ReduceOp.SUM and self.average and self.world > 1:
                        b["flat"].div_(self.world)
                    wire = b["flat"]
                    if b["stage"] is not None:
                        wire = b["stage"]
                        wire.copy_(b["flat"])
                    start = self._trace_mark(comm) if ready is not None else None
                    dist.all_reduce(wire, op=op, group=self.group, async_op=False)
                    if ready is not None:
                        self._trace["rows"].append((self._bucket_index[id(b["flat"])], wire.numel() * wire.element_size(),
                                                    len(b["params"]), ready, start, self._trace_mark(comm)))
                    if GradBucketReducer.exp_after_collective is not None:
                        GradBucketReducer.exp_after_collective(wire)
                    if b["stage"] is not None:
                        b["flat"].copy_(b["stage"])
                    self._final(b)                                    # (on the communication stream, behind the collective)
                    self._pending.append((comm.record_event(), None))

    # ---- per-bucket timeline of one step's exchange (bench.py, N > 1: the line explains itself the day a node exists) ----
    def trace_begin(self) -> None:
        """Mark the buckets of the step that starts now: `ready` (the bucket's last gradient is written: recorded on the firing
        stream), `start` / `end` of its collective (comm stream of our own: events around the call on that stream; the process
        group's stream: `start` = the issue point, `end` = an event on a helper stream that waits for the work).  ~3 event
        records per bucket; off unless called."""
        cuda = self.arena.is_cuda
        t0 = None
        if cuda:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record()
        else:
            import time
            t0 = time.perf_counter()
        self._trace = {"t0": t0, "rows": [], "cuda": cuda, "helper": torch.cuda.Stream(self.arena.device) if cuda else None}

    def _trace_mark(self, stream=None):
        tr = self._trace
        if tr["cuda"]:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream if stream is not None else torch.cuda.current_stream(self.arena.device))
            return ev
        import time
        return time.perf_counter()

    def trace_report(self):
        """-> rows {bucket, mb, params, ready_ms, start_ms, end_ms} relative to trace_begin() (synchronises); ends the trace."""
        tr, self._trace = self._trace, None
        if tr is None:
            return None
        if tr["cuda"]:
            torch.cuda.synchronize(self.arena.device)
        rel = (lambda e: tr["t0"].elapsed_time(e)) if tr["cuda"] else (lambda t: (t - tr["t0"]) * 1e3)
        rows = []
        for bi, nbytes, n_params, ready, start, end in tr["rows"]:
            rows.append({"bucket": bi, "bytes": int(nbytes), "mb": round(nbytes / 2 ** 20, 2), "params": n_params, "ready_ms": round(rel(ready), 3),
                         "start_ms": round(rel(start), 3), "end_ms": round(rel(end), 3)})
        return rows

    def seen_ids(self):
        """ids of the parameters that received a gradient since the last reset()."""
        out = set()
        for b in self.buckets:
            out |= b["seen"]
        return out

    def finish(self):
        """Flush buckets whose parameters did not all receive gradients, then wait for every collective."""
        self._in_finish = True           # buckets flushed here (parameters without a gradient) are left to optimizer.step()
        try:
            for b in self.buckets:
                if not b["fired"]:
                    # a parameter announced as overwritten whose producer did not run this step (a backward that was not
                    # reached, a partially used view) still holds the PREVIOUS step's gradient: reset() skipped it
                    for p in b["params"]:
                        if id(p) in self._overwritten and id(p) not in b["seen"]:
                            b["views"][id(p)].zero_()
                    self._fire(b)
        finally:
            self._in_finish = False
        if self.arena.is_cuda and self._opt_streams:
            cur = torch.cuda.current_stream(self.arena.device)
            for sid, st in self._opt_streams.items():      # early optimizer launches: whatever follows sees the new weights
                if sid != cur.cuda_stream:
                    cur.wait_stream(st)
        if self.arena.is_cuda:
            # Gradients are written straight into the buckets by kernels on several streams (view encoders, their weight-
            # gradient streams, replayed graphs) and no AccumulateGrad node runs for them, so the autograd engine has no
            # leaf stream to join at the end of backward(): whatever follows finish() on the current stream (the
            # optimizer) is ordered here behind every stream that produced a gradient.
            cur = torch.cuda.current_stream(self.arena.device)
            joined = set()
            for b in self.buckets:
                for sid, st in b["streams"].items():
                    if sid != cur.cuda_stream and sid not in joined:
                        cur.wait_stream(st)
                        joined.add(sid)
        if not self._pending:
            self._exposed = 0.0
            return
        cuda = self.arena.is_cuda
        if cuda:                                   # t0: the rank's own backward work is done on the current stream
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
        else:
            import time
            t0 = time.perf_counter()
        for w, b in self._pending:
            w.wait()                               # cuda: the current stream waits for the collective's stream / event
            if b is not None and b["stage"] is not None:
                b["flat"].copy_(b["stage"])
        if cuda:
            t1.record()
            self._exposed = (t0, t1)
        else:
            self._exposed = (time.perf_counter() - t0) * 1e3
        self._pending = []

    def exposed_ms(self) -> float:
        """Exposed (non-overlapped) exchange time of the last ``finish()`` in ms; synchronises on its end event."""
        e = self._exposed
        if isinstance(e, tuple):
            e[1].synchronize()
            e = self._exposed = e[0].elapsed_time(e[1])
        return float(e or 0.0)

    def remove(self):
        for h in self._hooks:
            h.remove()


def broadcast_module(module: torch.nn.Module, src: int = 0, group=None):
    """One-time parameter/buffer broadcast from rank 0 (identical replicas before step 0)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=group)
