import os, sys, collections, torch
sys.path.insert(0, "/root/repo")
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer
from torch.profiler import profile, ProfilerActivity
cfg = load_config("kradar"); torch.manual_seed(0); dev = torch.device("cuda", 0)
tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
data = make_batch(cfg["model"]["inputs"], 4, device=dev); labels = make_labels(4, device=dev)
for _ in range(3): tr.train_step(data, labels)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_step(data, labels); torch.cuda.synchronize()
cnt = collections.Counter()
LAUNCH = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::sum", "aten::add", "aten::add_", "aten::mul", "aten::cat", "aten::stack",
          "aten::clone", "aten::contiguous", "aten::zeros", "aten::_foreach_add_", "aten::div", "aten::sub", "aten::index", "aten::gather",
          "aten::argmax", "aten::gt", "aten::where", "aten::select_backward", "aten::slice_backward", "aten::index_put_", "aten::neg", "aten::exp", "aten::sigmoid")
for e in prof.events():
    if e.name not in LAUNCH: continue
    if e.cpu_parent is not None and e.cpu_parent.name in LAUNCH + ("aten::zeros_like", "aten::to", "aten::_to_copy", "aten::empty_like"): continue
    chain = []; p_ = e.cpu_parent
    while p_ is not None and len(chain) < 4:
        if not p_.name.startswith("aten::"): chain.append(p_.name[:40])
        p_ = p_.cpu_parent
    cnt[(e.name, str(e.input_shapes)[:50], " < ".join(chain))] += 1
for (n, sh, ch), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{c:4d} {n:16s} {sh:50s} | {ch}")
