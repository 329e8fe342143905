import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer
cfg = copy.deepcopy(load_config("kradar")); cfg["model"]["fuser"]["dropout"] = 0.0
cfg["model"]["backbones"]["camera_mono"]["name"] = "ResNet50"
SH = {"camera_mono": (96, 160, 3), "radar_bev": (128, 43, 6), "radar_front": (37, 107, 6)}
batch = make_batch(cfg["model"]["inputs"], 2, seed=9, shapes=SH, device="cuda")
labels = make_labels(2, seed=9, device="cuda")
def grads(graphs, use_reducer):
    torch.manual_seed(0)
    model = build("dprt", cfg)
    if use_reducer:
        tr = DataParallelTrainer(model, cfg, torch.device("cuda")); model = tr.model
    else:
        model = model.cuda(); tr = None
    from dpft_amd.training.loss import build_loss
    loss_fn = build_loss(cfg["train"])
    if graphs: model.enable_fuser_graph(batch)
    model.train()
    out_g = None
    for it in range(3):
        if tr: tr.reducer.reset()
        else: model.zero_grad(set_to_none=True)
        out = model(batch)
        loss, _ = loss_fn(out, labels)
        loss.backward()
        if tr: tr.reducer.finish()
        torch.cuda.synchronize()
        out_g = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    return out_g
ref = grads(False, False)
for name, (g, r) in {"graph,no reducer": (True, False), "eager,reducer": (False, True), "graph,reducer": (True, True)}.items():
    got = grads(g, r)
    bad = sorted(((float((got[k] - ref[k]).norm() / (ref[k].norm() + 1e-12)), k) for k in ref if k in got), reverse=True)[:4]
    print(name, [(round(e, 6), k[-60:]) for e, k in bad])
print("---- detail")
got = grads(True, False)
import itertools
keys320 = [k for k in ref if ref[k].shape == (320,)]
for k in keys320:
    e = float((got[k] - ref[k]).norm())
    if e > 1e-3:
        d = got[k] - ref[k]
        print(k[-70:], "abs err", round(e, 4), "ref norm", float(ref[k].norm()), "diff[:8]", [round(float(x), 4) for x in d[:8]])
        # is the difference equal to another tensor?
        for k2 in keys320:
            for nm, t in (("ref", ref[k2]), ("got", got[k2])):
                if float((d - t).norm()) < 1e-3 * float(t.norm() + 1e-9) and float(t.norm()) > 0:
                    print("    diff == ", nm, k2[-70:])
# absolute errors over all
errs = sorted(((float((got[k] - ref[k]).abs().max()), k) for k in ref), reverse=True)[:8]
print([(round(e, 5), k[-50:]) for e, k in errs])
