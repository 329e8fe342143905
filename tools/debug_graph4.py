import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch
cfg = copy.deepcopy(load_config("kradar")); cfg["model"]["fuser"]["dropout"] = 0.0
cfg["model"]["backbones"]["camera_mono"]["name"] = "ResNet50"
SH = {"camera_mono": (96, 160, 3), "radar_bev": (128, 43, 6), "radar_front": (37, 107, 6)}
batch = make_batch(cfg["model"]["inputs"], 2, seed=9, shapes=SH, device="cuda")
torch.manual_seed(0)
model = build("dprt", cfg).cuda().train()
model.enable_fuser_graph(batch)
g = model.__dict__["_graphed_fuser"]
names = {id(p): n for n, p in g.flat.named_parameters()}
items = []
for i, t in enumerate(g.static_grad_inputs):
    if t is None: continue
    nm = f"level{i}" if i < g.n_levels else names[id(g.params[i - g.n_levels])]
    items.append((t.data_ptr(), t.data_ptr() + t.numel() * 4, nm, t.untyped_storage().data_ptr(), t.untyped_storage().nbytes()))
items.sort()
print("n grads", len(items))
for a, b in zip(items, items[1:]):
    if b[0] < a[1]:
        print("OVERLAP", a[2], "<->", b[2])
for it in items:
    if "sampling_offsets.bias" in it[2]:
        print(it[2][-60:], hex(it[0]), it[1] - it[0], "storage", hex(it[3]), it[4])
