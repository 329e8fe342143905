"""Where the loader's time goes without a training loop next to it: the DataLoader alone (worker processes -> collated host
batches), + pinning, + upload and device transforms (PrefetchLoader).  FILES=1: from a K-Radar tree on disk.  WORKERS, N."""
import copy, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.data import GpuPreprocessor, KRadarFolderDataset, SyntheticRawDataset
from dpft_amd.data.loader import load_listed, _pin
import loader_rate

cfg = copy.deepcopy(load_config("kradar"))
cfg["computing"]["workers"] = int(os.environ.get("WORKERS", "16"))
N = int(os.environ.get("N", "40"))
B = cfg["train"]["batch_size"]
if os.environ.get("FILES") == "1":
    ds = KRadarFolderDataset(loader_rate.write_tree((N + 10) * B), camera="M", radar="BF", num_classes=2, fov=cfg["data"]["fov"], image_size=512)
else:
    ds = SyntheticRawDataset((N + 10) * B, seed=3)
pre = GpuPreprocessor.from_config(cfg)
loader, _ = load_listed(ds, cfg, device="cuda:0", preprocessor=pre, seed=1)
for name, it in (("DataLoader alone (host batches)", iter(loader.source)), ("+ pin_memory", iter(loader.source)), ("PrefetchLoader (pin + upload + device transforms)", iter(loader))):
    for _ in range(6):
        x = next(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N - 8):
        x = next(it)
        if name.startswith("+"):
            x = (_pin(x[0]), _pin(x[1]))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (N - 8)
    print(f"{name:52s} {dt * 1e3:7.2f} ms per batch of {B} = {B / dt:7.1f} samples/s", flush=True)
    del it
t0 = time.perf_counter()
for i in range(8):
    ds[i]
print(f"one sample in-process: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms")

# --- breakdown of the producer's steps on host batches ----------------------------------------------------------------------
from dpft_amd.data.loader import _PinnedRing, _to_device
ring = _PinnedRing(4)
it = iter(loader.source)
up = torch.cuda.Stream()
acc = {"next(DataLoader)": 0.0, "stage into the pinned arena": 0.0, "upload (enqueue)": 0.0, "device transforms (enqueue)": 0.0, "sync": 0.0}
n = 0
for _ in range(N - 4):
    t0 = time.perf_counter(); x = next(it); t1 = time.perf_counter()
    slot, host = ring.stage(x[0], x[1]); t2 = time.perf_counter()
    with torch.cuda.stream(up):
        b = _to_device(host[0], torch.device("cuda:0")); l = _to_device(host[1], torch.device("cuda:0")); t3 = time.perf_counter()
        b = pre(b); t4 = time.perf_counter()
        ev = torch.cuda.Event(); ev.record(up)
    ring.done(slot, ev)
    up.synchronize(); t5 = time.perf_counter()
    if _ >= 4:
        n += 1
        for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[k] += v
print({k: round(v / n * 1e3, 2) for k, v in acc.items()}, "ms per batch")
