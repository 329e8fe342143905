"""Which python lines launch the remaining ATen / runtime kernels of a training step (torch.profiler, with_stack)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer
from torch.profiler import profile, ProfilerActivity

cfg = load_config("kradar")
torch.manual_seed(0)
dev = torch.device("cuda", 0)
tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
data = make_batch(cfg["model"]["inputs"], 4, device=dev)
labels = make_labels(4, device=dev)
tr.enable_graphs(data)
for _ in range(3):
    tr.train_step(data, labels)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.train_step(data, labels)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if not e.name.startswith("aten::") or e.name in ("aten::empty", "aten::view", "aten::as_strided", "aten::empty_like", "aten::empty_strided"):
        continue
    if e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue
    par = e.cpu_parent.name if e.cpu_parent is not None else "-"
    st = [s for s in (e.stack or []) if "site-packages" not in s and "dist-packages" not in s][:2]
    cnt[(e.name, str(e.input_shapes)[:60], par[:50], " <- ".join(x[-60:] for x in st))] += 1
for (n, sh, par, s), c in cnt.most_common(70):
    print(f"{c:4d} {n:16s} {sh:60s} | {par} | {s}")
