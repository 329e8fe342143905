"""GPU time of the phases of the training step on the launch stream (events around model forward, loss, backward + exchange,
optimizer), untraced.   STEPS=40 python tools/phase_times.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer

steps = int(os.environ.get("STEPS", "40"))
cfg = load_config(os.environ.get("CONFIG", "kradar"))
B = int(os.environ.get("BATCH", "4"))
torch.manual_seed(0)
dev = torch.device("cuda", 0)
tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
data = make_batch(cfg["model"]["inputs"], B, device=dev)
labels = make_labels(B, device=dev)
tr.enable_graphs(data)
for _ in range(5):
    tr.train_step(data, labels)
torch.cuda.synchronize()
marks = []


def mark(tag):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((tag, e))


def wrap(obj, name, tag):
    f = getattr(obj, name)

    def g(*a, **k):
        r = f(*a, **k)
        mark(tag)
        return r
    setattr(obj, name, g)


model_call = tr.model.forward
tr.model.forward = lambda *a, **k: (lambda r: (mark("forward (encoders, necks, decoder)"), r)[1])(model_call(*a, **k))
loss_call = tr.loss_fn.forward
tr.loss_fn.forward = lambda *a, **k: (lambda r: (mark("loss"), r)[1])(loss_call(*a, **k))
wrap(tr.reducer, "finish", "backward")
wrap(tr.optimizer, "step", "optimizer")
tot = {}
t0 = time.perf_counter()
mark("start")
for _ in range(steps):
    tr.train_step(data, labels)
torch.cuda.synchronize()
for (ta, ea), (tb, eb) in zip(marks[:-1], marks[1:]):
    tot[tb] = tot.get(tb, 0.0) + ea.elapsed_time(eb)
print(f"ms/step {(time.perf_counter() - t0) / steps * 1e3:.2f}  (events on the launch stream, one sync at the end)")
for k, v in tot.items():
    print(f"  {k:40s} {v / steps:7.3f} ms")
print(f"  {'sum':40s} {sum(tot.values()) / steps:7.3f} ms")
