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


def _pin(obj):
    if isinstance(obj, torch.Tensor):
        return obj.pin_memory() if not obj.is_pinned() and torch.cuda.is_available() else obj
    if isinstance(obj, dict):
        return {k: _pin(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_pin(v) for v in obj)
    return obj


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

    def __init__(self, source: Iterable, device, preprocessor=None, depth: int = 2):
        self.source, self.device, self.preprocessor, self.depth = source, torch.device(device), preprocessor, depth

    def __len__(self):
        return len(self.source)

    def _producer(self, q: "queue.Queue", stop: threading.Event):
        try:
            up = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
            for inputs, targets in self.source:
                if stop.is_set():
                    break
                if up is None:
                    q.put((inputs, targets, None, None))
                    continue
                host = (_pin(inputs), _pin(targets))            # keep the pinned staging buffers alive until consumed
                with torch.cuda.stream(up):
                    batch = _to_device(host[0], self.device)
                    labels = _to_device(host[1], self.device)
                    if self.preprocessor is not None:
                        batch = self.preprocessor(batch)
                    ev = torch.cuda.Event()
                    ev.record(up)
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
    dl = DataLoader(dataset, batch_size=config["train"]["batch_size"], sampler=sampler,
                    num_workers=config.get("computing", {}).get("workers", 0), collate_fn=listed_collating,
                    drop_last=True, persistent_workers=False)
    return PrefetchLoader(dl, device, preprocessor), sampler
