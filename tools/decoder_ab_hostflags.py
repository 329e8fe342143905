"""Decoder forward time with host-side transformation flags (works with library builds before and after the device-flag
form): the A/B of tools/ab_cmd.sh for changes to decoder.hip."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch
cfg = load_config("kradar"); torch.manual_seed(0); dev = torch.device("cuda", 0)
m = build("dprt", cfg).to(dev).eval()
data = make_batch(cfg["model"]["inputs"], 4, device=dev)
with torch.no_grad():
    feats = m._encode_views(data)
    proj = m._get_projetions(m.inputs, data); shp = [data[f"{i}_shape"][:, :2] for i in m.inputs]
    vb = [feats[i] for i in m.inputs]; c0 = m.querent(data)
    flags = m.fuser.transformation_flags(proj)
    m.fuser(batch=vb, shape=shp, projection=proj, out=c0, has_transformation=flags)
    fd = m.fuser.__dict__["_fused_decoder"]
    fd.prepare(vb, shp, proj, c0, flags)
    ts = []
    for rnd in range(3):
        for _ in range(10): fd.launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(300): fd.launch()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 300 * 1e3)
    print("host flags", " ".join(f"{t:.1f}" for t in ts))
