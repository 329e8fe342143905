"""Throughput / roofline of the device-side input transforms and of the prefetch loader."""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.data import GpuPreprocessor, SyntheticRawDataset, load_listed
from dpft_amd.data.preprocess import resize_bilinear, scale_clip

dev = torch.device("cuda", 0)
B = 4
out = {}
def timed(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for name, dt in (("u8", torch.uint8), ("f32", torch.float32)):
    x = torch.randint(0, 256, (B, 720, 1280, 3), dtype=torch.uint8, device=dev).to(dt)
    t = timed(lambda: resize_bilinear(x, (512, 910)))
    byts = x.numel() * x.element_size() + B * 512 * 910 * 3 * 4
    out[f"resize_{name}"] = {"us_per_batch4": t * 1e6, "GB/s": byts / t / 1e9, "frac_of_8TBs": byts / t / 8e12}
r = torch.rand(B, 256, 107, 6, device=dev) * 200
t = timed(lambda: scale_clip(r))
out["radar_scale_clip"] = {"us_per_batch4": t * 1e6, "GB/s": 2 * r.numel() * 4 / t / 1e9}
# loader end to end (synthetic frames are generated on the host: this is a HOST-side number)
cfg = {"train": {"batch_size": B, "shuffle": True}, "computing": {"workers": int(os.environ.get("WORKERS", "8"))}}
ds = SyntheticRawDataset(96, seed=0)
loader, sampler = load_listed(ds, cfg, device=dev, preprocessor=GpuPreprocessor(512))
n = 0
t0 = None
for i, (batch, labels) in enumerate(loader):
    if i == 4:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    if i >= 4:
        n += batch["camera_mono"].shape[0]
torch.cuda.synchronize()
out["loader_samples_per_s"] = n / (time.perf_counter() - t0)
out["loader_workers"] = cfg["computing"]["workers"]
print(json.dumps(out))
