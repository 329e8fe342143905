"""GPU-side length of the loss window of the REAL training step (DataParallelTrainer.train_step): from the end of the matcher's
cost kernel to the launch of the decoder's backward graph, with DPFT_LSAP_C = 1 (C assignment solver, the default) | 0 (scipy per
sample), alternating in one process on one box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer
cfg = load_config("kradar"); torch.manual_seed(0); dev = torch.device("cuda", 0)
tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
data = make_batch(cfg["model"]["inputs"], 4, device=dev); labels = make_labels(4, device=dev)
tr.enable_graphs(data)
g = tr.model.__dict__["_graphed_fuser"]
ev = {"cost": torch.cuda.Event(enable_timing=True), "bwd": torch.cuda.Event(enable_timing=True)}
orig_to_host = tr.loss_fn._to_host
def to_host(t):
    ev["cost"].record()
    return orig_to_host(t)
tr.loss_fn._to_host = to_host
orig_replay = g.bwd_graph.replay
def rp():
    ev["bwd"].record()
    orig_replay()
g.bwd_graph.replay = rp
for _ in range(8):
    tr.train_step(data, labels)
res = {}
for rnd in range(4):
    for lsap in ("0", "1"):
        for direct in ("-",):
            os.environ["DPFT_LSAP_C"] = lsap
            win, step = [], []
            for _ in range(30):
                ev["cost"], ev["bwd"] = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); tr.train_step(data, labels); e1.record()
                torch.cuda.synchronize()
                win.append(ev["cost"].elapsed_time(ev["bwd"]) * 1e3); step.append(e0.elapsed_time(e1))
            win.sort(); step.sort()
            res.setdefault((lsap, direct), []).append((win[15], step[15]))
for k, v in res.items():
    print(f"C solver={k[0]}: window median us {[round(a) for a, _ in v]}  step median ms {[round(b, 3) for _, b in v]}")
