"""Host time inside the window that the GPU spends waiting for the host: from the matcher's D2H sync to the launch of
the backward graph (monkeypatched timestamps, no extra syncs)."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer
from dpft_amd.training import loss as L
from dpft_amd.models.fusers import graphed as G
cfg = load_config("kradar"); torch.manual_seed(0); dev = torch.device("cuda", 0)
tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
data = make_batch(cfg["model"]["inputs"], 4, device=dev); labels = make_labels(4, device=dev)
tr.enable_graphs(data)
marks = {}
orig_lsa = L.linear_sum_assignment
def lsa(c):
    marks.setdefault("sync_done", time.perf_counter())
    t0 = time.perf_counter(); r = orig_lsa(c); marks["lsa"] = marks.get("lsa", 0.0) + time.perf_counter() - t0
    return r
L.linear_sum_assignment = lsa
orig_bf = G.GraphedFuser.backward_from
def bf(self, write):
    marks["backward_from_enter"] = time.perf_counter()
    orig_replay = self.bwd_graph.replay
    def rp():
        ev["bwd_start"][cur[0]].record()
        marks["bwd_graph_replay_call"] = time.perf_counter()
        orig_replay()
        marks["bwd_graph_replay_returned"] = time.perf_counter()
    self.bwd_graph.replay = rp
    try:
        return orig_bf(self, write)
    finally:
        self.bwd_graph.replay = orig_replay
G.GraphedFuser.backward_from = bf
# GPU-side view of the same window: events behind the forward (decoder graph end), behind the matcher's cost kernel (recorded when
# the read-back is entered) and in front of the backward graph
ev = {k: [torch.cuda.Event(enable_timing=True) for _ in range(25)] for k in ("fwd_end", "cost", "bwd_start")}
cur = [0]
orig_to_host = tr.loss_fn._to_host
def to_host(t):
    ev["cost"][cur[0]].record()
    r = orig_to_host(t)
    marks["sync_done"] = time.perf_counter()      # the matcher's read-back has landed: the host window starts
    return r
tr.loss_fn._to_host = to_host
acc = {}
for it in range(25):
    cur[0] = it
    marks.clear()
    t_a = time.perf_counter()
    tr.model.train(); tr.reducer.reset()
    out = tr.model(data)
    ev["fwd_end"][it].record()
    loss, losses = tr.loss_fn(out, labels)
    marks["loss_fn_done"] = time.perf_counter()
    ok = bool(loss > 0)
    marks["gt_sync_done"] = time.perf_counter()
    if not tr._backward_without_engine(loss):
        loss.backward()
    marks["backward_returned"] = time.perf_counter()
    tr.reducer.finish(); tr.optimizer.set_active(tr.reducer.seen_ids()); tr.optimizer.step()
    if it >= 5:
        s = marks["sync_done"]
        for k in ("lsa",): acc[k] = acc.get(k, 0) + marks.get(k, 0.0)
        for k in ("loss_fn_done", "gt_sync_done", "backward_from_enter", "bwd_graph_replay_call", "bwd_graph_replay_returned", "backward_returned"):
            acc[k] = acc.get(k, 0) + marks[k] - s
torch.cuda.synchronize()
n = 20
print("per step, host time from the matcher's sync (us):")
print(f"  scipy assignments (4 samples; 0 = the C solver is in use) {acc['lsa'] / n * 1e6:7.0f}")
for k in ("loss_fn_done", "gt_sync_done", "backward_from_enter", "bwd_graph_replay_call", "bwd_graph_replay_returned", "backward_returned"):
    print(f"  -> {k:22s} {acc[k] / n * 1e6:7.0f}")
g1 = sorted(ev["fwd_end"][i].elapsed_time(ev["cost"][i]) * 1e3 for i in range(5, 25))
g2 = sorted(ev["cost"][i].elapsed_time(ev["bwd_start"][i]) * 1e3 for i in range(5, 25))
print(f"GPU side (median, us): forward end -> matcher's cost kernel done {g1[10]:.0f}   |   cost done -> backward graph start {g2[10]:.0f}")
