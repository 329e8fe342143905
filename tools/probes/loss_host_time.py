"""Where the host time between the matcher's read-back and the return of Loss.forward goes (perf_counter marks around the pieces)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer
from dpft_amd.training import loss as L
from dpft_amd.hip.lib import lib
cfg = load_config("kradar"); torch.manual_seed(0); dev = torch.device("cuda", 0)
tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
data = make_batch(cfg["model"]["inputs"], 4, device=dev); labels = make_labels(4, device=dev)
tr.enable_graphs(data)
marks = []
def mark(name): marks.append((name, time.perf_counter()))
orig_to_host = tr.loss_fn._to_host
def to_host(t):
    mark("to_host_enter"); r = orig_to_host(t); mark("to_host_done"); return r
tr.loss_fn._to_host = to_host
dll = lib.load()
orig_c = dll.dpft_assign_loss_f32
class W:
    def __call__(self, *a):
        mark("c_enter"); r = orig_c(*a); mark("c_done"); return r
lib.__dict__.setdefault("_over", {})
import types
orig_getattr = type(lib).__getattr__
def ga(self, name):
    if name == "dpft_assign_loss_f32": return W()
    return orig_getattr(self, name)
type(lib).__getattr__ = ga
orig_apply = L._SetLossFn.apply
def ap(*a):
    mark("apply_enter"); r = orig_apply(*a); mark("apply_done"); return r
L._SetLossFn.apply = staticmethod(ap)
acc = {}
for it in range(30):
    marks.clear()
    loss, _ = tr.train_step(data, labels)
    mark("train_step_returned")
    if it >= 5:
        t0 = dict(marks).get("to_host_done")
        for (n, t) in marks:
            acc.setdefault(n, []).append((t - t0) * 1e6)
for n, v in acc.items():
    v.sort(); print(f"{n:24s} median {v[len(v)//2]:9.1f} us")
