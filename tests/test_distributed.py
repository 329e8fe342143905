"""world_size-2 gloo test of the bucketed gradient reducer (runs on CPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Flatten(),
                                       torch.nn.Linear(8 * 6 * 6, 10), torch.nn.ReLU(), torch.nn.Linear(10, 4))
        self.net[0].weight.data = self.net[0].weight.data.contiguous(memory_format=torch.channels_last)
        self.unused = torch.nn.Linear(3, 3)          # never receives a gradient (like the template head)

    def forward(self, x):
        return self.net(x)


def _make_model():
    torch.manual_seed(0)
    return _Net()


def _worker(rank, world, port, q, wire=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dpft_amd.training.distributed import GradBucketReducer, broadcast_module
    model = _make_model()
    if rank != 0:                              # replicas start different; broadcast must fix that
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    broadcast_module(model)
    red = GradBucketReducer(list(model.parameters()), bucket_bytes=2048,     # several small buckets
                            comm_dtype=torch.bfloat16 if wire == "bf16" else None)
    assert len(red.buckets) > 2
    assert sum(b["flat"].numel() for b in red.buckets) <= red.arena.numel()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 3, 6, 6, generator=g)
    y = torch.randn(4, 4, generator=g)
    xs, ys = x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2]
    for step in range(2):                      # second pass checks reset()
        red.reset()
        if step == 1:
            red.trace_begin()                  # the per-bucket timeline bench.py prints for N > 1
        loss = ((model(xs) - ys) ** 2).sum(1).mean()       # per-sample mean => averaging = global-batch grad
        loss.backward()
        # one gradient is delivered through the direct sink path as the hand-scheduled backward does
        red.finish()
        assert red.exposed_ms() >= 0.0
    rows = red.trace_report()
    assert red.trace_report() is None          # the trace ended with its report
    assert len(rows) == len(red.buckets) and sorted(r["bucket"] for r in rows) == list(range(len(red.buckets)))
    assert all(0.0 <= r["ready_ms"] <= r["start_ms"] <= r["end_ms"] and r["bytes"] > 0 and r["params"] >= 1 for r in rows), rows
    grads = {n: p.grad.clone() for n, p in model.named_parameters()}
    q.put((rank, {k: v.numpy() for k, v in grads.items()}))
    dist.destroy_process_group()


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_bucketed_allreduce_equals_global_batch_gradient(wire):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, wire)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = _make_model()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 3, 6, 6, generator=g)
    y = torch.randn(4, 4, generator=g)
    ((model(x) - y) ** 2).sum(1).mean().backward()
    for n, p in model.named_parameters():
        for r in range(world):
            got = torch.from_numpy(res[r][n])
            if p.grad is None:
                assert float(got.abs().max()) == 0.0
            elif wire == "bf16":                 # each rank's contribution and the sum are rounded to 8 mantissa bits
                torch.testing.assert_close(got, p.grad, rtol=2e-2, atol=2e-2 * float(p.grad.abs().max()))
                assert torch.equal(got, torch.from_numpy(res[0][n]))          # replicas stay bit-identical
            else:
                torch.testing.assert_close(got, p.grad, rtol=1e-5, atol=1e-6)


def test_reducer_sink_and_single_process():
    from dpft_amd.training.distributed import GradBucketReducer
    model = _make_model()
    red = GradBucketReducer(list(model.parameters()), bucket_bytes=1 << 20)
    red.reset()
    w = model.net[0].weight
    g = torch.ones_like(w)
    assert red.grad_sink(w, g)
    red.finish()
    assert torch.equal(w.grad, g) and w.grad.permute(0, 2, 3, 1).is_contiguous()
    assert not red.grad_sink(torch.nn.Parameter(torch.zeros(1)), torch.zeros(1))


def test_reducer_clears_only_parameters_that_accumulate():
    """GradBucketReducer.set_overwritten: reset() leaves the bucket views of producers that overwrite their gradients
    alone (or clears them too where a small one sits between cleared spans) and zeroes everything else; clear() zeroes
    the views of given parameters on demand."""
    import torch
    from dpft_amd.training.distributed import GradBucketReducer
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(n)) for n in (10, 20000, 64, 7, 30000, 5)]
    r = GradBucketReducer(ps, bucket_bytes=1 << 20)
    r.set_overwritten([ps[1], ps[4]])
    r.arena.fill_(1.0)
    r.reset()
    view = lambda p: r.buckets[r._index[id(p)]]["views"][id(p)]
    for i in (0, 2, 3, 5):
        assert float(view(ps[i]).abs().sum()) == 0.0, i           # accumulating parameters: cleared
    for i in (1, 4):
        assert float(view(ps[i]).sum()) == ps[i].numel(), i        # overwritten parameters: untouched
    assert all(p.grad is view(p) for p in ps)
    r.clear([ps[1]])
    assert float(view(ps[1]).abs().sum()) == 0.0 and float(view(ps[4]).sum()) == ps[4].numel()
    r.set_overwritten([])                                          # back to the full clear
    r.arena.fill_(1.0)
    r.reset()
    assert float(r.arena.abs().sum()) == 0.0
