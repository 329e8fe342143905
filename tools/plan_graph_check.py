"""Losses of N trainer steps on a fixed small batch -- run once with DPFT_PLAN_GRAPHS=0 and once with =1 and compare."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer
dev = torch.device("cuda", 0)
cfg = load_config("kradar")
cfg["model"]["fuser"]["dropout"] = 0.0
cfg["model"]["backbones"]["camera_mono"]["name"] = "ResNet50"
torch.manual_seed(3)
tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
shapes = {"camera_mono": (128, 224, 3), "radar_bev": (128, 43, 6), "radar_front": (37, 107, 6)}
data = make_batch(cfg["model"]["inputs"], 2, seed=7, shapes=shapes, device=dev)
labels = make_labels(2, seed=3, device=dev)
if os.environ.get("FUSER_GRAPH", "0") == "1":
    tr.enable_graphs(data)
out = []
for step in range(int(os.environ.get("STEPS", "8"))):
    loss, _ = tr.train_step(data, labels)
    g = torch.cat([b["flat"] for b in tr.reducer.buckets]).double()
    out.append((float(loss), float(g.norm()), float(g.abs().sum())))
for i, o in enumerate(out):
    print(f"step {i}: loss {o[0]:.6f} |g| {o[1]:.6f} sum|g| {o[2]:.4f}")
m = tr.model
print("graphed plans:", {i: [p.graphed for p in m.backbones[i]._plans.values()] for i in m.inputs})
