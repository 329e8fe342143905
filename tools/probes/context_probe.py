"""Are the reduced radar-front train step's results a function of what ran earlier in the process?  Step A (fresh), then a
full-size 3-view eval forward at batch 4, then step B on a freshly built identical model: outputs and gradients compared bit
for bit, the first differing tensors named."""
import os, sys, copy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as T
from dpft_amd.synthetic import make_batch
DEV = "cuda"


def step(hook_backbone=False):
    cfg = T.view_config("kradar_radar_front", dropout=0.0)
    g = torch.Generator().manual_seed(32)
    model = T._build(cfg, g).to(DEV).train()
    batch = make_batch(cfg["model"]["inputs"], 6, seed=7, shapes=T.SHAPES)
    feats = {}
    bb = model.backbones["radar_front"]
    orig = bb.forward
    def fwd(x):
        out = orig(x)
        for k, v in out.items():
            feats[f"stage{k}"] = v.detach().clone()
        return out
    bb.forward = fwd
    out = model({k: v.to(DEV) for k, v in batch.items()})
    cots = {k: torch.randn(out[k].shape, generator=g).to(DEV) for k in out}
    sum((out[k] * cots[k]).sum() for k in out).backward()
    torch.cuda.synchronize()
    res = {f"out.{k}": v.detach().clone() for k, v in out.items()}
    res.update(feats)
    res.update({f"grad.{n}": p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    return res


a = step()
a2 = step()
same = [k for k in a if torch.equal(a[k], a2[k])]
print(f"fresh vs fresh: {len(same)} of {len(a)} tensors bit-equal;", "first differing:", [k for k in a if not torch.equal(a[k], a2[k])][:6])
if os.environ.get("MIDDLE", "eval") == "eval":
    from dpft_amd.configs import load_config
    from dpft_amd.models import build
    cfg = load_config("kradar")
    torch.manual_seed(0)
    big = build("dprt", cfg).to(DEV).eval()
    with torch.no_grad():
        big(make_batch(cfg["model"]["inputs"], 4, seed=1, device=DEV))
    torch.cuda.synchronize()
    del big
b = step()
diff = [(k, float((a[k].double() - b[k].double()).norm() / (a[k].double().norm() + 1e-30))) for k in a if not torch.equal(a[k], b[k])]
print(f"fresh vs after-full-size-eval: {len(a) - len(diff)} of {len(a)} bit-equal")
for k, e in diff[:12]:
    print("   ", k, f"{e:.3e}")
print("   largest:", sorted(diff, key=lambda t: -t[1])[:6])
