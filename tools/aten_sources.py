"""Which Python call sites still launch ATen kernels in a training step (copy_ / fill_ / zero_ / sum / add ...)?
A TorchDispatchMode over 3 steps after warm-up records every ATen op that reaches the device together with the innermost
dpft_amd frames of the Python stack at that moment (ops issued by autograd's C++ nodes show the frame that started the backward)."""
import os, sys, collections, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.utils._python_dispatch import TorchDispatchMode
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer

cfg = load_config("kradar")
torch.manual_seed(0)
dev = torch.device("cuda", 0)
tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
data = make_batch(cfg["model"]["inputs"], 4, device=dev)
labels = make_labels(4, device=dev)
SPY_CAPTURE = os.environ.get("CAPTURE") == "1"      # list what the decoder's forward / backward graphs record instead
if not SPY_CAPTURE:
    tr.enable_graphs(data)
for _ in range(0 if SPY_CAPTURE else 6):
    tr.train_step(data, labels)
torch.cuda.synchronize()
N = 3
VIEWS = ("view", "reshape", "alias", "detach", "t", "transpose", "permute", "select", "slice", "unsqueeze", "squeeze", "expand",
         "as_strided", "_unsafe_view", "unbind", "split", "narrow", "empty", "empty_like", "empty_strided", "new_empty", "size", "stride",
         "is_pinned", "_local_scalar_dense", "record_stream", "set_", "lift_fresh", "_reshape_alias", "unflatten", "flatten", "movedim",
         "new_empty_strided", "sym_size", "sym_stride", "sym_numel", "chunk", "split_with_sizes", "unsafe_split", "view_as", "resize_")
agg = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in VIEWS:
            ts = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
            if any(t.is_cuda for t in ts) or (not ts and "cuda" in str((kwargs or {}).get("device", ""))):
                frames = [f for f in traceback.extract_stack() if "/dpft_amd/" in f.filename]
                site = " <- ".join(f"{f.filename.split('/dpft_amd/')[-1]}:{f.lineno} {f.name}" for f in frames[-2:][::-1]) or "(outside dpft_amd)"
                shape = "x".join(str(s) for s in ts[0].shape) if ts else ""
                agg[(name, site, shape)] += 1
        return func(*args, **(kwargs or {}))


if SPY_CAPTURE:
    from dpft_amd.models.fusers import graphed
    orig = graphed.GraphedFuser.__init__
    def spied(self, model, sample_batch, warmup=3, grad_direct=None):      # warm-up outside the spy: only the captures count
        orig(self, model, sample_batch, warmup=warmup, grad_direct=grad_direct)
    N = 1
    import torch.cuda.graphs as G
    real_enter, real_exit = G.graph.__enter__, G.graph.__exit__
    spy = Spy()
    def enter(self):
        r = real_enter(self); spy.__enter__(); return r
    def exit_(self, *a):
        spy.__exit__(*a); return real_exit(self, *a)
    G.graph.__enter__, G.graph.__exit__ = enter, exit_
    tr.enable_graphs(data)
    torch.cuda.synchronize()
else:
    with Spy():
        for _ in range(N):
            tr.train_step(data, labels)
        torch.cuda.synchronize()
tot = sum(agg.values()) / N
print(f"{tot:.1f} device-touching ATen ops per step")
for (name, site, shape), n in sorted(agg.items(), key=lambda kv: -kv[1])[:int(os.environ.get("TOP", "90"))]:
    print(f"{n / N:6.1f}  {name:20s} {shape:16s} {site[:200]}")
