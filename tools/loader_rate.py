"""Loader-in-the-loop training rate: the 16-worker DataLoader of the reference's config
(config/kradar.json: workers 16; src/dprt/datasets/loader.py:37-44) -> listed collate -> pinned staging + upload on its own
stream -> device-side resize / radar scaling (GpuPreprocessor) -> DataParallelTrainer.train_step, against the same steps on
ONE resident batch.  Samples are RAW-sized: 720 x 1280 x 3 uint8 camera frames, 256 x 107 x 6 / 37 x 107 x 6 fp32 dB maps
(SyntheticRawDataset; generating a frame costs the worker about what a memcpy of a decoded frame would).
FILES=1 (round 4): the same measurement on FILES -- a pre-processed K-Radar tree of raw-sized samples written to a temp dir
(720 x 1280 JPEG frames, ra.npy / ea.npy maps, calibration, labels) and read by KRadarFolderDataset: JPEG decode (Pillow;
the reference: torchvision.io.read_image) and np.load are then inside the workers.

    python tools/loader_rate.py            -> one JSON line
    FILES=1 python tools/loader_rate.py    -> the same from files on disk
"""
import copy, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.data import GpuPreprocessor, KRadarFolderDataset, SyntheticRawDataset
from dpft_amd.data.loader import load_listed
from dpft_amd.models import build
from dpft_amd.training.trainer import DataParallelTrainer


def write_tree(n, root=None):
    """n raw-sized samples in the reference's folder layout (16 distinct frames, hard-linked / re-saved round robin)."""
    import tempfile
    import numpy as np
    from PIL import Image
    root = root or tempfile.mkdtemp(prefix="kradar_tree_")
    rng = np.random.default_rng(0)
    base = []
    for i in range(16):
        # low-pass noise: a frame that compresses like a photograph (pure noise would make the JPEG 3x larger and slower)
        small = rng.integers(0, 256, size=(90, 160, 3), dtype=np.uint8)
        base.append(np.asarray(Image.fromarray(small).resize((1280, 720), Image.BICUBIC)))
    for i in range(n):
        d = os.path.join(root, "train", f"{i // 64 + 1}", f"{i % 64:05d}")
        os.makedirs(d)
        Image.fromarray(base[i % 16]).save(os.path.join(d, "mono.jpg"), quality=90)
        np.save(os.path.join(d, "mono_info.npy"), np.eye(4))
        np.save(os.path.join(d, "ra.npy"), (60.0 + 180.0 * rng.random(size=(256, 107, 6))).astype(np.float32))
        np.save(os.path.join(d, "ra_info.npy"), np.eye(4))
        np.save(os.path.join(d, "ea.npy"), (60.0 + 180.0 * rng.random(size=(37, 107, 6))).astype(np.float32))
        np.save(os.path.join(d, "ea_info.npy"), np.eye(4))
        k = int(rng.integers(1, 6))
        boxes = np.concatenate([rng.uniform(5, 60, (k, 1)), rng.uniform(-5, 5, (k, 1)), rng.uniform(-1, 3, (k, 1)),
                                rng.uniform(-3, 3, (k, 1)), rng.uniform(1, 5, (k, 3)), np.zeros((k, 1)), np.arange(k)[:, None]], axis=1)
        np.save(os.path.join(d, "labels.npy"), boxes)
        np.save(os.path.join(d, "description.npy"), np.zeros(6))
    return root


def main():
    workers = int(os.environ.get("WORKERS", "16"))
    steps = int(os.environ.get("STEPS", "60"))
    cfg = copy.deepcopy(load_config("kradar"))
    cfg["computing"]["workers"] = workers
    B = cfg["train"]["batch_size"]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
    files = os.environ.get("FILES") == "1"
    if files:
        ds = KRadarFolderDataset(write_tree((steps + 12) * B), camera="M", radar="BF", num_classes=cfg["data"].get("num_classes", 2),
                                 fov=cfg["data"].get("fov"), image_size=cfg["data"].get("image_size", 512))
    else:
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
                                    "the device (upload stream)",
                      "source": ("files on disk: KRadarFolderDataset, JPEG decode (Pillow) + np.load inside the workers" if files
                                 else "SyntheticRawDataset (generated in the workers)"),
                      "not_included": None if files else "JPEG decode / file reads (FILES=1 measures them)"}))


if __name__ == "__main__":
    main()
