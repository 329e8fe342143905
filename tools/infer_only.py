"""N eval forwards of kradar.json at BATCH (default 1): target of per-kernel rocprof breakdowns of the inference path."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch

cfg = load_config("kradar")
B = int(os.environ.get("BATCH", "1"))
reps = int(os.environ.get("REPS", "50"))
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = build("dprt", cfg).to(dev).eval()
data = make_batch(cfg["model"]["inputs"], B, device=dev)
with torch.no_grad():
    for _ in range(10):
        model(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model(data)
    torch.cuda.synchronize()
print(f"fwd ms/batch {(time.perf_counter() - t0) / reps * 1e3:.3f}  batch {B}")
