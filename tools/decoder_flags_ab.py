import os, sys, torch
sys.path.insert(0, "/root/repo")
import bench
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch
cfg = load_config("kradar"); torch.manual_seed(0); dev = torch.device("cuda", 0)
m = build("dprt", cfg).to(dev).eval()
data = make_batch(cfg["model"]["inputs"], 4, device=dev)
with torch.no_grad():
    run, feats = bench.decoder_runner(m, data)
    fd = m.fuser.__dict__["_fused_decoder"]
    proj = m._get_projetions(m.inputs, data); shp = [data[f"{i}_shape"][:, :2] for i in m.inputs]
    vb = [feats[i] for i in m.inputs]; c0 = m.querent(data)
    def t(reps=300):
        for _ in range(10): fd.launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fd.launch()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    for rnd in range(3):
        fd.prepare(vb, shp, proj, c0, None); a = t()
        fd.prepare(vb, shp, proj, c0, m.fuser.transformation_flags(proj)); b = t()
        print(f"device flags {a:.1f} us   host flags {b:.1f} us")
    # same 9 launches replayed from a hipGraph: how much of the forward is dispatch / boundary latency?
    fd.prepare(vb, shp, proj, c0, None)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fd.launch()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fd.launch()
    torch.cuda.synchronize()
    for _ in range(10): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(300): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"graph replay {e0.elapsed_time(e1) / 300 * 1e3:.1f} us")
