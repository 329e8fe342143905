"""Which Python call sites still launch ATen kernels in a training step (copy_ / fill_ / zero_ / sum / add ...)?
torch.profiler with stacks over 3 steps after warm-up; prints op, count per step, innermost dpft_amd frames."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
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
tr.enable_graphs(data)
for _ in range(6):
    tr.train_step(data, labels)
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    for _ in range(N):
        tr.train_step(data, labels)
    torch.cuda.synchronize()
LAUNCHING = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::sum", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::div_",
             "aten::cat", "aten::stack", "aten::index", "aten::gt", "aten::repeat", "aten::clone", "aten::_foreach_add_", "aten::dot",
             "aten::mean", "aten::sub", "aten::neg", "aten::where", "aten::_to_copy", "aten::contiguous", "aten::select_backward")
agg = collections.Counter()
for ev in prof.events():
    if ev.name not in LAUNCHING:
        continue
    frames = [f for f in (ev.stack or []) if "dpft_amd" in f or "bench.py" in f or "tools/" in f]
    site = " <- ".join(f.split("/root/repo/")[-1].strip() for f in frames[:2]) or "(no dpft_amd frame: autograd engine / torch internals)"
    agg[(ev.name, site)] += 1
for (name, site), n in sorted(agg.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{n / N:6.1f}  {name:22s} {site[:230]}")
