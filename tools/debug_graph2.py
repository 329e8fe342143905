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
KEY = "fuser.mpfusion.fusion1.ml_fusion_layers.ms_deform_attn0.ms_deform_attn.sampling_offsets.bias"
def go(graphs, sync):
    torch.manual_seed(0)
    tr = DataParallelTrainer(build("dprt", cfg), cfg, torch.device("cuda"))
    if graphs: tr.enable_graphs(batch)
    tr.model.train()
    res = []
    for _ in range(3):
        tr.reducer.reset()
        out = tr.model(batch)
        loss, _ = tr.loss_fn(out, labels)
        loss.backward()
        if sync: torch.cuda.synchronize()
        tr.reducer.finish()
        g = dict(tr.model.named_parameters())[KEY].grad
        res.append((float(g.norm()), float(loss)))
    return res
print("eager        ", go(False, False))
print("graph nosync ", go(True, False))
print("graph sync   ", go(True, True))
