"""Time of the K-Radar export of one batch (B=4, N=400, kradar thresholds): the selection launch, the whole export()
(two launches + 4 device->host copies + text files on a tmpfs) and the oracle's per-(threshold, sample) torch-CPU path."""
import os, sys, time, json, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.evaluation.exporters import build
from dpft_amd.synthetic import make_labels
from oracle import export_oracle as EO
cfg = load_config("kradar")
ex = build("kradar", cfg)
g = torch.Generator().manual_seed(0)
B, N = 4, 400
out = {"class": torch.randn(B, N, 2, generator=g), "size": 1 + torch.rand(B, N, 3, generator=g) * 4,
       "center": torch.stack((torch.rand(B, N, generator=g) * 72, -6 + torch.rand(B, N, generator=g) * 12,
                              -1 + torch.rand(B, N, generator=g) * 3), -1)}
yaw = (torch.rand(B, N, generator=g) * 2 - 1) * 3.1
out["angle"] = torch.stack((torch.sin(yaw), torch.cos(yaw)), -1)
labels = make_labels(B)
for b, l in enumerate(labels):
    l["description"] = torch.tensor([b, b % 2, b])
dout, dlab = {k: v.cuda() for k, v in out.items()}, [{k: v.cuda() for k, v in l.items()} for l in labels]
for _ in range(3): ex.select(dout["class"], dout["center"], dout["size"], dout["angle"], ex.conf_thrs)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(100): ex.select(dout["class"], dout["center"], dout["size"], dout["angle"], ex.conf_thrs)
e1.record(); torch.cuda.synchronize()
sel_us = e0.elapsed_time(e1) / 100 * 1e3
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
    ex(dout, dlab, 0, d); torch.cuda.synchronize()
    a = time.perf_counter()
    for i in range(10): ex(dout, dlab, 4 * (i + 1), d)
    torch.cuda.synchronize(); whole = (time.perf_counter() - a) / 10
a = time.perf_counter()
for i in range(3): EO.export_tree(out, labels, 0, categories=cfg["data"].get("categories"))
cpu = (time.perf_counter() - a) / 3
n_obj = int(sum(c.sum() for c in [ex.select(dout["class"], dout["center"], dout["size"], dout["angle"], ex.conf_thrs)[1]]))
print(json.dumps({"batch": B, "queries": N, "thresholds": len(ex.conf_thrs), "objects_written": n_obj,
                  "select_launch_us_incl_host": sel_us, "export_ms_per_batch": whole * 1e3,
                  "oracle_torch_cpu_ms_per_batch_no_files": cpu * 1e3, "host_threads": torch.get_num_threads()}))
