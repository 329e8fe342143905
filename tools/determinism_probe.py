"""Run-to-run spread of the training trajectory (two identically seeded trainers in one process, 6 steps): the only sources should
be the fp32 atomics of the training decoder's pyramid gradients.  usage: python tools/determinism_probe.py [reps]"""
import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer

SHAPES = {"camera_mono": (96, 160, 3), "radar_bev": (128, 43, 6), "radar_front": (37, 107, 6)}
cfg = copy.deepcopy(load_config("kradar"))
cfg["model"]["backbones"]["camera_mono"]["name"] = "ResNet50"
cfg["model"]["fuser"]["dropout"] = 0.0
dev = torch.device("cuda", 0)
batch = make_batch(cfg["model"]["inputs"], 2, seed=4, shapes=SHAPES, device=dev)
labels = make_labels(2, seed=4, device=dev)
sync_each = os.environ.get("SYNC_EACH", "1") == "1"


def run():
    torch.manual_seed(21)
    tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
    tr.enable_graphs(batch)
    out = []
    for _ in range(6):
        l = tr.train_step(batch, labels)[0]
        out.append(float(l) if sync_each else l)
    torch.cuda.synchronize()
    return [float(v) for v in out]


worst = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    a, b = run(), run()
    rel = [abs(x - y) / max(abs(y), 1.0) for x, y in zip(a, b)]
    worst.append(max(rel))
    print(" ".join(f"{r:.1e}" for r in rel))
print("max over reps", f"{max(worst):.2e}", "median", f"{sorted(worst)[len(worst) // 2]:.2e}")
