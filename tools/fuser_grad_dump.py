"""Per-tensor gradient comparison of the decoder / FPN / head parameters at full size (kradar.json, batch 4, dropout 0):
HIP path vs the fp32 oracle vs the fp64 oracle -- the data behind tests/test_gpu_model.py::test_full_size_train_step_matches_oracle's
per-tensor gates.  Writes gpurun_out/fuser_grads.pt (hip / fp32 / fp64 gradients of every non-backbone parameter) + a table."""
import copy, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_model import _build, state_dict_f64, rel_l2          # noqa: E402
from dpft_amd.configs import load_config                           # noqa: E402
from dpft_amd.synthetic import make_batch, make_labels             # noqa: E402
from dpft_amd.training.loss import build_loss                      # noqa: E402
from oracle import dprt_oracle as O                                # noqa: E402

g = torch.Generator().manual_seed(int(os.environ.get("SEED", "41")))
cfg = copy.deepcopy(load_config("kradar")); cfg["model"]["fuser"]["dropout"] = 0.0
model = _build(cfg, g)
sd64 = state_dict_f64(model)
batch = make_batch(cfg["model"]["inputs"], 4, seed=9); labels = make_labels(4, seed=9)
w = cfg["train"]["loss_weights"]
torch.set_num_threads(min(64, os.cpu_count() or 1))
res = {}
for name, dtype in (("f32", torch.float32), ("f64", torch.float64)):
    sd = {k: (v.to(dtype).clone().requires_grad_(True) if v.is_floating_point() and "running" not in k
              else (v.to(dtype) if v.is_floating_point() else v)) for k, v in sd64.items()}
    b = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in batch.items()}
    lab = [{k: (v.to(dtype) if v.is_floating_point() else v) for k, v in l.items()} for l in labels]
    out = O.dprt_forward(sd, cfg, b, train=True)
    loss, _ = O.loss_forward(out, lab, w)
    loss.backward()
    res[name] = {k: v.grad for k, v in sd.items() if v.is_floating_point() and v.grad is not None and not k.startswith("backbones")}
model = model.to("cuda").train()
loss_fn = build_loss(cfg["train"])
dev_labels = [{k: v.to("cuda") for k, v in l.items()} for l in labels]
runs = []
for rep in range(2):
    model.zero_grad(set_to_none=True)
    out = model({k: v.to("cuda") for k, v in batch.items()})
    loss, _ = loss_fn(out, dev_labels)
    loss.backward()
    runs.append({n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None and not n.startswith("backbones")})
hip = runs[0]
rows = []
for n in res["f64"]:
    if n not in hip: continue
    g64, g32, gh = res["f64"][n], res["f32"][n], hip[n]
    rows.append((rel_l2(gh, g64) / max(rel_l2(g32, g64), 1e-9), n, float(g64.norm()), rel_l2(gh, g64), rel_l2(g32, g64), rel_l2(runs[1][n], gh)))
rows.sort(reverse=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "fuser_grads.txt"), "w") as f:
    for r in rows:
        line = f"x{r[0]:8.1f} {r[1]:90s} |g64| {r[2]:.3e} hip {r[3]:.2e} fp32 {r[4]:.2e} hip-rerun {r[5]:.2e}"
        print(line); f.write(line + "\n")
torch.save({"hip": hip, "hip2": runs[1], "f32": res["f32"], "f64": res["f64"]}, os.path.join(ROOT, "gpurun_out", "fuser_grads.pt"))
