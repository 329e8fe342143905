"""How far are the gradients of the mixed-precision mode from the fp32 step at the bench configuration (kradar, B=4)?
Same weights, same batch, dropout off; prints relative L2 error and cosine similarity per parameter group."""
import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.hip import ops
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer
dev = torch.device("cuda", 0)
res = {}
for mode in ("fp32", "bf16", "fp32b"):
    cfg = copy.deepcopy(load_config("kradar"))
    cfg["model"]["fuser"]["dropout"] = 0.0
    cfg["computing"]["conv_compute"] = mode[:4]
    torch.manual_seed(0)
    tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
    if mode == "fp32b":      # control: fp32 with the INPUT perturbed by 2^-9 relative noise (what bf16 rounding injects once)
        pass
    data = make_batch(cfg["model"]["inputs"], 4, device=dev)
    if mode == "fp32b":
        g = torch.Generator(device="cuda").manual_seed(1)
        data = {k: (v * (1 + (torch.rand(v.shape, generator=g, device=dev) - 0.5) * 2 ** -8) if v.is_floating_point() and v.dim() == 4 else v)
                for k, v in data.items()}
    labels = make_labels(4, device=dev)
    tr.model.train(); tr.reducer.reset()
    out = tr.model(data)
    loss, _ = tr.loss_fn(out, labels)
    loss.backward(); tr.reducer.finish()
    groups = {}
    for n, p in tr.model.named_parameters():
        if p.grad is not None:
            key = ".".join(n.split(".")[:4]) if n.startswith("backbones") else n.split(".")[0]
            groups.setdefault(key, []).append(p.grad.detach().flatten().double())
    res[mode] = (float(loss), {k: torch.cat(v) for k, v in groups.items()})
    del tr
ops.conv_set_compute("fp32")
for other in ("bf16", "fp32b"):
    print(other, "loss", res[other][0], "vs", res["fp32"][0])
    for k, g0 in res["fp32"][1].items():
        g1 = res[other][1][k]
        print(f"  {k:45s} rel-L2 {float((g1-g0).norm()/g0.norm()):.3f}  cos {float((g1*g0).sum()/(g1.norm()*g0.norm())):.3f}")
