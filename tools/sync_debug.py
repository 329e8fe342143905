"""List the host<->device synchronisation points of one training step and the host-side duration of a step."""
import os, sys, time, torch, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for _ in range(3):
    tr.train_step(data, labels)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    tr.train_step(data, labels)
torch.cuda.set_sync_debug_mode("default")
for x in w:
    print("SYNC:", str(x.message)[:100], "|", x.filename.split("/")[-1], x.lineno)
torch.cuda.synchronize()
# host-side time of a step (time until train_step returns) vs wall per step
hs = []
t0 = time.perf_counter()
for _ in range(10):
    a = time.perf_counter()
    tr.train_step(data, labels)
    hs.append(time.perf_counter() - a)
torch.cuda.synchronize()
print("host ms per step", [round(h * 1e3, 1) for h in hs], "wall ms/step", (time.perf_counter() - t0) / 10 * 1e3)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.train_step(data, labels)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
