"""Host time vs GPU time of an eval forward at batch 1 / 4 (is the forward launch-bound, does the plan replay help?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch
cfg = load_config("kradar")
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = build("dprt", cfg).to(dev).eval()
for B in (1, 4):
    data = make_batch(cfg["model"]["inputs"], B, device=dev)
    with torch.no_grad():
        for _ in range(6):
            model(data)
        torch.cuda.synchronize()
        host, tot = [], []
        for _ in range(30):
            t0 = time.perf_counter()
            model(data)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
    host.sort(); tot.sort()
    print(f"B={B}: host issue {host[len(host)//2]:.2f} ms, forward incl. sync {tot[len(tot)//2]:.2f} ms  (DPFT_EVAL_GRAPHS={os.environ.get('DPFT_EVAL_GRAPHS', '1')})")
