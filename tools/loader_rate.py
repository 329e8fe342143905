"""Loader-in-the-loop training rate: the 16-worker DataLoader of the reference's config
(config/kradar.json: workers 16; src/dprt/datasets/loader.py:37-44) -> listed collate -> pinned staging + upload on its own
stream -> device-side resize / radar scaling (GpuPreprocessor) -> DataParallelTrainer.train_step, against the same steps on
ONE resident batch.  Samples are RAW-sized: 720 x 1280 x 3 uint8 camera frames, 256 x 107 x 6 / 37 x 107 x 6 fp32 dB maps
(SyntheticRawDataset; generating a frame costs the worker about what a memcpy of a decoded frame would).
What this does NOT contain is the JPEG decode of the real dataset (dataset.py:120-139 -> torchvision.io.read_image, ~5-8 ms per
1280 x 720 frame and core: 16 workers sustain ~2-3 k frames/s, above the 8-GPU step rate) -- there is no image codec in
this image.

    python tools/loader_rate.py            -> one JSON line
"""
import copy, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.data import GpuPreprocessor, SyntheticRawDataset
from dpft_amd.data.loader import load_listed
from dpft_amd.models import build
from dpft_amd.training.trainer import DataParallelTrainer


def main():
    workers = int(os.environ.get("WORKERS", "16"))
    steps = int(os.environ.get("STEPS", "60"))
    cfg = copy.deepcopy(load_config("kradar"))
    cfg["computing"]["workers"] = workers
    B = cfg["train"]["batch_size"]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
    ds = SyntheticRawDataset((steps + 12) * B, seed=3)
    pre = GpuPreprocessor.from_config(cfg)
    loader, sampler = load_listed(ds, cfg, device=dev, preprocessor=pre, seed=1)
    it = iter(loader)
    batch, labels = next(it)
    tr.enable_graphs(batch)
    for _ in range(8):                                   # warm-up on loader batches (graph capture of the plans included)
        batch, labels = next(it)
        tr.train_step(batch, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for batch, labels in it:
        tr.train_step(batch, labels)
        n += 1
        if n == steps:
            break
    torch.cuda.synchronize()
    t_loader = (time.perf_counter() - t0) / n
    for _ in range(3):
        tr.train_step(batch, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.train_step(batch, labels)
    torch.cuda.synchronize()
    t_res = (time.perf_counter() - t0) / n
    print(json.dumps({"workers": workers, "steps": n, "batch": B,
                      "loader_in_the_loop_samples_per_s": B / t_loader, "loader_ms_per_step": 1e3 * t_loader,
                      "resident_batch_samples_per_s": B / t_res, "resident_ms_per_step": 1e3 * t_res,
                      "loader_over_resident": t_res / t_loader,
                      "raw_sample": "720x1280x3 u8 + 256x107x6 f32 + 37x107x6 f32 per sample; resize to 512x910 and radar scaling on "
                                    "the device (upload stream)", "not_included": "JPEG decode (no codec in the image)"}))


if __name__ == "__main__":
    main()
