"""Sanity: the full training path (fused decoder with dropout, HIP loss, side-stream wgrad, fused AdamW, graphs) must be
able to overfit one fixed synthetic batch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer
cfg = load_config("kradar")
if os.environ.get("COMPUTE"):      # fp32 | bf16 (mixed-precision mode of the conv GEMMs)
    cfg["computing"]["conv_compute"] = os.environ["COMPUTE"]
torch.manual_seed(int(os.environ.get("SEED", "0")))
dev = torch.device("cuda", 0)
tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
data = make_batch(cfg["model"]["inputs"], 4, device=dev)
labels = make_labels(4, device=dev)
if os.environ.get("EAGER", "0") == "1":          # torch-op decoder, heads and loss (autograd) instead of the fused kernels
    from dpft_amd.models.fusers import mpfusion
    mpfusion.MPFusion.use_fused_train = False
    mpfusion.IMPFusion.use_fused_train = False
    tr.loss_fn.use_fused = False
else:
    tr.enable_graphs(data)
hist = []
EVERY = int(os.environ.get("EVERY", "25"))
for i in range(int(os.environ.get("STEPS", "150"))):
    out = tr.train_step(data, labels, with_metrics=(i % EVERY == 0))
    if i % EVERY == 0:
        hist.append((i, round(float(out[0]), 3), {k: round(float(v), 3) for k, v in out[2].items()}))
hist.append(("last", round(float(out[0]), 3)))
print(hist)
# diagnostics: what the predictions look like at the end (eval mode, same batch)
if os.environ.get("DIAG", "1") == "1":
    print({k: round(float(v), 4) for k, v in out[1].items()})
    tr.model.train()
    with torch.no_grad():
        ot = tr.model(data)
    print("train-mode argmax histogram", [int((ot["class"].argmax(-1) == c).sum()) for c in range(ot["class"].shape[-1])],
          "metrics", {k: round(float(v), 3) for k, v in tr.eval_fn(ot, labels).items()})
    tr.model.eval()
    with torch.no_grad():
        o = tr.model(data)
    print("eval-mode metrics", {k: round(float(v), 3) for k, v in tr.eval_fn(o, labels).items()})
    cls = o["class"].float()
    lab = cls.argmax(-1)
    print("argmax histogram", [int((lab == c).sum()) for c in range(cls.shape[-1])])
    print("class logits min/max", [(round(float(cls[..., c].min()), 3), round(float(cls[..., c].max()), 3)) for c in range(cls.shape[-1])])
    for b in range(len(labels)):
        d = (o["center"][b][:, None, :] - labels[b]["gt_center"][None]).norm(dim=-1)   # (N, M)
        best, idx = d.min(0)
        print(b, "closest pred distance per gt", [round(float(x), 2) for x in best],
              "its logits", [[round(float(v), 2) for v in cls[b, int(i)]] for i in idx])
    print("predicted size of the query closest to the first gt of each sample",
          [[round(float(v), 2) for v in o["size"][b, int((o["center"][b] - labels[b]["gt_center"][0]).norm(dim=-1).argmin())]]
           for b in range(len(labels))], "gt", [[round(float(v), 2) for v in labels[b]["gt_size"][0]] for b in range(len(labels))])
