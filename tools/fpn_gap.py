"""GPU idle time between the end of the decoder's backward graph and the first kernel of the (camera) FPN backward, without a
profiler: an event behind the graph launch, one at the entry of the first _FPNFn.backward of the step.  If the host is in
time the two events are processed back to back (elapsed ~ 0); a late host shows up as elapsed time."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.models.necks import fpn as fpn_mod
from dpft_amd.models.fusers import graphed as graphed_mod
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer

cfg = load_config("kradar")
torch.manual_seed(0)
dev = torch.device("cuda", 0)
tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
data = make_batch(cfg["model"]["inputs"], 4, device=dev)
labels = make_labels(4, device=dev)
tr.enable_graphs(data)
rec = {"dec": None, "fpn": None, "host_dec": 0.0, "host_fpn": 0.0, "pairs": []}
orig_ab = torch.autograd.backward
def ab(*a, **k):          # GraphedFuser.backward_from starts autograd on the pyramids right behind the graph launch
    if rec["dec"] is None:
        e = torch.cuda.Event(enable_timing=True); e.record(); rec["dec"] = e; rec["host_dec"] = time.perf_counter()
    return orig_ab(*a, **k)
graphed_mod.torch.autograd.backward = ab
orig_bw = fpn_mod._FPNFn.backward
def bw(ctx, *douts):
    if rec["fpn"] is None and torch.cuda.current_stream().cuda_stream == 0:
        e = torch.cuda.Event(enable_timing=True); e.record(); rec["fpn"] = e; rec["host_fpn"] = time.perf_counter()
    return orig_bw(ctx, *douts)
fpn_mod._FPNFn.backward = staticmethod(bw)
for it in range(16):
    rec["dec"] = rec["fpn"] = None
    tr.train_step(data, labels)
    torch.cuda.synchronize()
    if it >= 6 and rec["dec"] is not None and rec["fpn"] is not None:
        rec["pairs"].append((rec["dec"].elapsed_time(rec["fpn"]) * 1e3, (rec["host_fpn"] - rec["host_dec"]) * 1e6))
print("decoder-backward end -> camera FPN backward entry: gpu us", [round(a) for a, _ in rec["pairs"]])
print("host time between the two points (us):", [round(b) for _, b in rec["pairs"]])
