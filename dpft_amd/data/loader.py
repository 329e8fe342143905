"""Data loading around the model for one-process-per-GPU training.

* ``listed_collating`` -- the reference's collate contract (src/dprt/datasets/loader.py:10-34): inputs are stacked into a
  dict of batched tensors, targets stay a list of per-sample dicts.
* ``ShardedSampler`` -- per-rank index shards of an epoch-seeded permutation (the reference is single-device and uses
  ``shuffle=`` of one DataLoader, loader.py:37-44); every rank draws the same permutation and takes a strided slice.
* ``PrefetchLoader`` -- wraps any iterable of (inputs, targets) host batches: a worker thread stages them in pinned
  memory, uploads them on its own HIP stream, runs the ``GpuPreprocessor`` there and hands over (batch, labels) with an
  event the consumer's stream waits on, so upload + preprocessing of batch i+1 overlap the training step of batch i.
"""
from __future__ import annotations

import queue
import threading
from typing import Any, Dict, Iterable, Iterator, List, Optional, Tuple

import torch
from torch.utils.data import DataLoader, Dataset, Sampler, default_collate


def listed_collating(data: List[Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]]):
    inputs, targets = zip(*data)
    return default_collate(list(inputs)), list(targets)


class ShardedSampler(Sampler[int]):
    def __init__(self, n: int, rank: int = 0, world: int = 1, shuffle: bool = True, seed: int = 0,
                 drop_last: bool = True):
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world of {world}")
        self.n, self.rank, self.world, self.shuffle, self.seed, self.drop_last = n, rank, world, shuffle, seed, drop_last
        self.epoch = 0
        self.per_rank = n // world if drop_last else (n + world - 1) // world

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def __len__(self) -> int:
        return self.per_rank

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        total = self.per_rank * self.world
        if total > len(order):                       # pad by wrapping around (drop_last = False)
            order += order[: total - len(order)]
        return iter(order[self.rank:total:self.world])


class BlockShardedSampler(Sampler[int]):
    """Evaluation shards: rank r takes the CONTIGUOUS block [start, start + len) of the split, in order, nothing dropped and
    nothing repeated (the last rank may get fewer) -- sample k keeps the index, and with it the export file name, it has in
    a one-process run (dpft_amd/evaluation/evaluator.py)."""

    def __init__(self, n: int, rank: int = 0, world: int = 1):
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world of {world}")
        per = (n + world - 1) // world
        self.start = min(n, rank * per)
        self.stop = min(n, self.start + per)

    def __len__(self) -> int:
        return self.stop - self.start

    def __iter__(self) -> Iterator[int]:
        return iter(range(self.start, self.stop))


def _pin(obj):
    if isinstance(obj, torch.Tensor):
        return obj.pin_memory() if not obj.is_pinned() and torch.cuda.is_available() else obj
    if isinstance(obj, dict):
        return {k: _pin(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_pin(v) for v in obj)
    return obj


class _PinnedRing:
    """Pinned staging for the batches in flight: ONE page-locked arena per slot, allocated once (grow-only) -- every tensor of
    a batch (inputs and label dicts: ~30 of them) is copied into its slot's arena and uploaded from there.  Round 4:
    ``tensor.pin_memory()`` per tensor page-locks a fresh allocation each time and cost 34 ms per batch of four raw K-Radar
    samples (14 MB) -- the loader delivered 118 samples/s to a model that trains at 146 (tools/probes/loader_alone.py); the
    copies into a standing arena take ~2 ms.  A slot is reused only after the upload that read it has completed (event)."""
    ALIGN = 256

    def __init__(self, slots: int):
        self.arena = [None] * slots
        self.event = [None] * slots
        self.i = 0

    @classmethod
    def _bytes(cls, obj) -> int:
        if isinstance(obj, torch.Tensor):
            return (obj.numel() * obj.element_size() + cls.ALIGN - 1) // cls.ALIGN * cls.ALIGN
        if isinstance(obj, dict):
            return sum(cls._bytes(v) for v in obj.values())
        if isinstance(obj, (list, tuple)):
            return sum(cls._bytes(v) for v in obj)
        return 0

    def _place(self, obj, arena, off):
        if isinstance(obj, torch.Tensor):
            n = obj.numel() * obj.element_size()
            if n == 0 or obj.is_pinned():
                return obj, off
            view = arena[off[0]:off[0] + n].view(obj.dtype).view(obj.shape)
            view.copy_(obj)
            off[0] += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            return view, off
        if isinstance(obj, dict):
            return {k: self._place(v, arena, off)[0] for k, v in obj.items()}, off
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._place(v, arena, off)[0] for v in obj), off
        return obj, off

    def stage(self, *objs):
        """-> (slot, pinned copies of ``objs`` with the same structure).  Call ``done(slot, event)`` after the upload."""
        slot = self.i % len(self.arena)
        self.i += 1
        if self.event[slot] is not None:
            self.event[slot].synchronize()
        need = sum(self._bytes(o) for o in objs)
        if self.arena[slot] is None or self.arena[slot].numel() < need:
            self.arena[slot] = torch.empty(int(need * 1.25) + 4096, dtype=torch.uint8).pin_memory()
        off = [0]
        return slot, tuple(self._place(o, self.arena[slot], off)[0] for o in objs)

    def done(self, slot: int, event) -> None:
        self.event[slot] = event


class SlotCollate:
    """Collate of the reference's contract (inputs stacked into a dict of batched tensors, targets a list of dicts) that
    writes the stacked inputs into a SHARED, page-locked slot instead of a fresh shared-memory segment per tensor.

    Why (round 4, tools/probes/loader_alone.py): a worker's batch reaches the main process as ~12 new shm segments; the
    first touch of their 14 MB there (the copy into pinned staging) page-faults its way through at 0.6 GB/s -- 23 ms per
    batch of four raw K-Radar samples, more than a training step leaves (27 ms) once the GIL is shared with it.  Here the
    workers do the stacking copy themselves, in parallel, into a ring of slots that the main process mapped and registered
    with the HIP runtime ONCE (hipHostRegister): the main process only enqueues the upload.

    Slot of a batch = (worker id, k-th batch of that worker mod ``per_worker``).  A worker runs at most ``prefetch``
    batches ahead of the consumer, so with ``per_worker = prefetch + 2`` a slot is rewritten two full rounds of all workers
    after its batch was handed over -- long after its upload was enqueued and completed.  Batches whose shapes do not fit
    the slot layout (first sample's shapes x batch size) fall back to the plain collate."""
    KEY = "__slot__"
    ALIGN = 256

    def __init__(self, sample_inputs: Dict[str, torch.Tensor], batch_size: int, n_workers: int, prefetch: int = 2):
        self.batch_size, self.n_workers, self.per_worker = batch_size, max(1, n_workers), prefetch + 2
        self.layout, off = [], 0
        for k, v in sample_inputs.items():
            n = batch_size * v.numel() * v.element_size()
            self.layout.append((k, v.dtype, (batch_size,) + tuple(v.shape), off, n))
            off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.slot_bytes = max(off, self.ALIGN)
        self.ring = torch.empty((self.n_workers * self.per_worker, self.slot_bytes), dtype=torch.uint8).share_memory_()
        self.ring.zero_()                 # touch every page in the creating process before it is registered
        self.count = 0
        self.registered = False

    def register(self) -> bool:
        """Page-lock the ring for the HIP runtime (main process, once).  False: the runtime refused; uploads still work."""
        if not self.registered and torch.cuda.is_available():
            try:
                rc = torch.cuda.cudart().cudaHostRegister(self.ring.data_ptr(), self.ring.numel(), 0)
                self.registered = int(rc) == 0
            except Exception:
                self.registered = False
        return self.registered

    def views(self, slot: int) -> Dict[str, torch.Tensor]:
        buf = self.ring[slot]
        return {k: buf[off:off + n].view(dt).view(shape) for k, dt, shape, off, n in self.layout}

    def __call__(self, data):
        inputs, targets = zip(*data)
        fits = len(inputs) == self.batch_size and all(
            list(i.keys()) == [k for k, *_ in self.layout] and
            all(i[k].dtype == dt and tuple(i[k].shape) == shape[1:] for k, dt, shape, _, _ in self.layout) for i in inputs)
        if not fits:
            return default_collate(list(inputs)), list(targets)
        info = torch.utils.data.get_worker_info()
        wid = info.id if info is not None else 0
        slot = wid * self.per_worker + self.count % self.per_worker
        self.count += 1
        views = self.views(slot)
        for k, *_ in self.layout:
            torch.stack([i[k] for i in inputs], out=views[k])
        return {self.KEY: torch.tensor(slot)}, list(targets)


def _to_device(obj, device):
    if isinstance(obj, torch.Tensor):
        return obj.to(device, non_blocking=True)
    if isinstance(obj, dict):
        return {k: _to_device(v, device) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_device(v, device) for v in obj)
    return obj


class PrefetchLoader:
    _END = object()

    def __init__(self, source: Iterable, device, preprocessor=None, depth: int = 2, slots: "SlotCollate" = None):
        self.source, self.device, self.preprocessor, self.depth = source, torch.device(device), preprocessor, depth
        self.slots = slots                  # the workers' shared, page-locked batch slots (SlotCollate) or None

    def __len__(self):
        return len(self.source)

    @property
    def sampler(self):
        """The wrapped DataLoader's sampler (DataParallelEvaluator reads ``sampler.start`` of an evaluation block)."""
        return getattr(self.source, "sampler", None)

    def _producer(self, q: "queue.Queue", stop: threading.Event):
        try:
            up = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
            ring = _PinnedRing(self.depth + 2) if up is not None else None      # queue depth + the one being filled + the one consumed
            # Slot reuse: a worker writes batch n + per_worker into the slot of its batch n.  The DataLoader hands a
            # worker its next task when the main process TAKES a batch from it, batches arrive round-robin over the workers, and
            # a worker runs `prefetch` tasks ahead: taking global batch i dispatches the task that overwrites the slot of batch
            # i - (per_worker - prefetch) * workers.  Its upload must have completed by then -- waited for here (an event that
            # old has long fired unless the upload stream is stalled; with one worker the distance is two batches).
            upload_done: Dict[int, "torch.cuda.Event"] = {}
            reuse = None
            if self.slots is not None:
                reuse = max(1, self.slots.per_worker - 2) * self.slots.n_workers
            it, i = iter(self.source), -1
            while True:
                i += 1
                if reuse is not None:
                    old_ev = upload_done.pop(i - reuse, None)
                    if old_ev is not None:
                        old_ev.synchronize()
                try:
                    inputs, targets = next(it)
                except StopIteration:
                    break
                if stop.is_set():
                    break
                in_slot = self.slots is not None and isinstance(inputs, dict) and SlotCollate.KEY in inputs
                if in_slot:                                     # the stacked inputs already sit in a page-locked shared slot
                    inputs = self.slots.views(int(inputs[SlotCollate.KEY]))
                if up is None:
                    q.put(({k: v.clone() for k, v in inputs.items()} if in_slot else inputs, targets, None, None))
                    continue
                if in_slot:
                    slot, (host_t,) = ring.stage(targets)
                    host = (inputs, host_t)
                else:
                    slot, host = ring.stage(inputs, targets)    # pinned copies in the slot's standing arena
                with torch.cuda.stream(up):
                    batch = _to_device(host[0], self.device)
                    labels = _to_device(host[1], self.device)
                    if self.preprocessor is not None:
                        batch = self.preprocessor(batch)
                    ev = torch.cuda.Event()
                    ev.record(up)
                ring.done(slot, ev)
                if in_slot:
                    upload_done[i] = ev
                q.put((batch, labels, ev, host))
        except BaseException as e:          # surface loader errors in the consumer thread
            q.put(e)
        finally:
            q.put(self._END)

    def __iter__(self):
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        th = threading.Thread(target=self._producer, args=(q, stop), daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is self._END:
                    break
                if isinstance(item, BaseException):
                    raise item
                batch, labels, ev, _host = item
                if ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)
                    for t in list(batch.values()) + [v for l in labels for v in l.values() if isinstance(v, torch.Tensor)]:
                        t.record_stream(torch.cuda.current_stream(self.device))
                yield batch, labels
        finally:
            stop.set()
            while th.is_alive():             # drain so that the producer can observe `stop`
                try:
                    q.get(timeout=0.1)
                except queue.Empty:
                    pass
            th.join()


def load_listed(dataset: Dataset, config: Dict[str, Any], device="cpu", rank: int = 0, world: int = 1,
                preprocessor=None, seed: int = 0) -> Tuple[PrefetchLoader, ShardedSampler]:
    """The reference's ``load_listed`` (loader.py:37-44) for one rank of a data-parallel job."""
    sampler = ShardedSampler(len(dataset), rank, world, shuffle=config["train"].get("shuffle", True), seed=seed)
    workers = config.get("computing", {}).get("workers", 0)
    slots = None
    import os
    if torch.device(device).type == "cuda" and workers > 0 and len(dataset) > 0 and os.environ.get("DPFT_LOADER_SLOTS", "1") != "0":
        # the workers stack their batches straight into shared page-locked slots (SlotCollate); layout from the first sample
        # (workers x 4 slots of one batch each in /dev/shm, page-locked: ~450 MB per rank for 16 workers on raw K-Radar frames.  A
        # container with a small /dev/shm cannot hold that: fall back to the plain collate + the pinned staging ring)
        try:
            slots = SlotCollate(dataset[0][0], config["train"]["batch_size"], workers)
            slots.register()
        except (RuntimeError, OSError, MemoryError) as e:
            import warnings
            warnings.warn(f"shared batch slots unavailable ({e}); using the pinned staging ring (slower loader)")
            slots = None
    dl = DataLoader(dataset, batch_size=config["train"]["batch_size"], sampler=sampler, num_workers=workers,
                    collate_fn=slots if slots is not None else listed_collating, drop_last=True, persistent_workers=False)
    return PrefetchLoader(dl, device, preprocessor, slots=slots), sampler



def load_listed_eval(dataset: Dataset, config: Dict[str, Any], device="cpu", rank: int = 0, world: int = 1,
                     preprocessor=None) -> Tuple[PrefetchLoader, BlockShardedSampler]:
    """``load_listed`` for EVALUATION over several ranks: rank r reads the contiguous block of the split a
    ``BlockShardedSampler`` gives it, in order, nothing shuffled, nothing dropped (``drop_last=False``) -- every sample is
    scored and exported exactly once and keeps the file index it has in a one-process run.  The returned loader exposes
    ``.sampler`` (``.start`` = the block's first global index, which ``DataParallelEvaluator.evaluate_one_epoch`` numbers the
    export files from)."""
    sampler = BlockShardedSampler(len(dataset), rank, world)
    workers = config.get("computing", {}).get("workers", 0)
    dl = DataLoader(dataset, batch_size=config["train"]["batch_size"], sampler=sampler, num_workers=workers,
                    collate_fn=listed_collating, drop_last=False, persistent_workers=False)
    return PrefetchLoader(dl, device, preprocessor), sampler
