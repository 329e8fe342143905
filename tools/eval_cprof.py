import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, "/root/repo")
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch
cfg = load_config("kradar"); torch.manual_seed(0); dev = torch.device("cuda", 0)
m = build("dprt", cfg).to(dev).eval()
data = make_batch(cfg["model"]["inputs"], 4, device=dev)
with torch.no_grad():
    for _ in range(5): m(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m(data); torch.cuda.synchronize()
    print("wall ms/forward", (time.perf_counter() - t0) / 20 * 1e3)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20):
        m(data); torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
