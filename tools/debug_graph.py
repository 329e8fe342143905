import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch
cfg = copy.deepcopy(load_config("kradar")); cfg["model"]["fuser"]["dropout"] = 0.0
cfg["model"]["backbones"]["camera_mono"]["name"] = "ResNet50"
SH = {"camera_mono": (96, 160, 3), "radar_bev": (128, 43, 6), "radar_front": (37, 107, 6)}
torch.manual_seed(0)
m = build("dprt", cfg).cuda().train()
batch = make_batch(cfg["model"]["inputs"], 2, seed=9, shapes=SH, device="cuda")
def run():
    m.zero_grad(set_to_none=True)
    out = m(batch)
    sum((v * (i + 1)).sum() for i, v in enumerate(out.values())).backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
g_e = run(); g_e2 = run()
m.enable_fuser_graph(batch)
gs = [run() for _ in range(3)]
def worst(a, b):
    e = sorted(((float((a[k] - b[k]).norm() / (b[k].norm() + 1e-12)), k, float(b[k].norm()), float(a[k].norm())) for k in b if "fuser" in k), reverse=True)
    return e[:3]
print("eager vs eager  :", worst(g_e2, g_e))
for i, g in enumerate(gs):
    print(f"graph{i} vs eager:", worst(g, g_e))
print("graph1 vs graph2:", worst(gs[1], gs[2]))
