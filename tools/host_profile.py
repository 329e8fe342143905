"""Where the HOST's time per training step goes (cProfile over N steps; DTYPE=bf16 BATCH=4 is the host-bound configuration).
usage: DTYPE=bf16 BATCH=4 python tools/host_profile.py > gpurun_out/host_profile.txt"""
import cProfile, io, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer

steps = int(os.environ.get("STEPS", "30"))
cfg = load_config("kradar")
B = int(os.environ.get("BATCH", "4"))
if os.environ.get("DTYPE", "f32") == "bf16":
    cfg["computing"]["conv_compute"] = "bf16"
torch.manual_seed(0)
dev = torch.device("cuda", 0)
tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
data = make_batch(cfg["model"]["inputs"], B, device=dev)
labels = make_labels(B, device=dev)
tr.enable_graphs(data)
for _ in range(10):
    tr.train_step(data, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.train_step(data, labels)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host ms/step {(t1 - t0) / steps * 1e3:.2f}   wall ms/step {(t2 - t0) / steps * 1e3:.2f}   pacing {tr.__dict__.get('_pace_state')}")
# ---- wall time inside the backward's Python pieces (they run on the autograd engine's thread: invisible to cProfile below)
import collections, functools
acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)


def timed(owner, name, label=None):
    fn = getattr(owner, name)
    label = label or f"{getattr(owner, '__name__', type(owner).__name__)}.{name}"
    raw = fn.__func__ if hasattr(fn, "__func__") and not isinstance(owner, type) else fn

    @functools.wraps(raw)
    def wrap(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t
            cnt[label] += 1
    return wrap


import dpft_amd.models.backbones.resnet as R
import dpft_amd.models.necks.fpn as F
import dpft_amd.models.fusers.graphed as G
from dpft_amd.hip import ops as O
R._BodyFn.backward = staticmethod(timed(R._BodyFn, "backward", "_BodyFn.backward"))
R._BodyFn.forward = staticmethod(timed(R._BodyFn, "forward", "_BodyFn.forward"))
F._FPNFn.backward = staticmethod(timed(F._FPNFn, "backward", "_FPNFn.backward"))
F._FPNFn.forward = staticmethod(timed(F._FPNFn, "forward", "_FPNFn.forward"))
red = tr.reducer
for nme in ("mark_ready_many", "finish", "reset"):
    setattr(red, nme, timed(red, nme, "reducer." + nme))
tr.optimizer.step = timed(tr.optimizer, "step", "optimizer.step")
tr.loss_fn.forward_fused = timed(tr.loss_fn, "forward_fused", "loss.forward_fused")
g = tr.model.__dict__.get("_graphed_fuser")
g.backward_from = timed(g, "backward_from", "graphed.backward_from(incl. engine)")
tr._backward_without_engine = timed(tr, "_backward_without_engine", "trainer._backward_without_engine")
mdl = tr.model
mdl.forward = timed(mdl, "forward", "model.forward")
t0 = time.perf_counter()
for _ in range(steps):
    tr.train_step(data, labels)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"instrumented host ms/step {(t1 - t0) / steps * 1e3:.2f}")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:46s} {v / steps * 1e3:8.3f} ms/step   {cnt[k] / steps:6.1f} calls/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    tr.train_step(data, labels)
pr.disable()
torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
