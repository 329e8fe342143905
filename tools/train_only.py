"""N plain training steps (no latency / decoder / cpu legs) -- the target of per-step rocprof breakdowns."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer

steps = int(os.environ.get("STEPS", "20"))
cfg = load_config("kradar")
B = int(os.environ.get("BATCH", "4"))
if os.environ.get("DTYPE", "f32") == "bf16":          # configs[4]: bf16 operands / activations, fp32 accumulation
    cfg["computing"]["conv_compute"] = "bf16"
torch.manual_seed(0)
dev = torch.device("cuda", 0)
tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
data = make_batch(cfg["model"]["inputs"], B, device=dev)
labels = make_labels(B, device=dev)
if os.environ.get("SERIAL") == "1":        # the serialized step bench.py brackets: one view stream, no wgrad side stream
    from dpft_amd.hip.lib import lib
    lib.call("dpft_profile_serialize", 1)
    tr.model.concurrent_views = False
if os.environ.get("GRAPHS", "1") == "1":
    tr.enable_graphs(data)
for _ in range(3):
    tr.train_step(data, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.train_step(data, labels)
torch.cuda.synchronize()
print(f"ms/step {(time.perf_counter() - t0) / steps * 1e3:.2f}  (steps {steps})")
