"""GPU-side and host-side timeline of un-serialized training steps: CUDA events recorded at the phase boundaries on
the main stream (no syncs added) + host clocks at the same points.  gpu_ms = time the main stream spent between two
boundaries; host_ms = time the host needed to ISSUE the phase.  host >> gpu for a phase right after a sync = the GPU
starves there."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.optimizer import FusedAdamW
from dpft_amd.training.trainer import DataParallelTrainer

cfg = load_config("kradar")
torch.manual_seed(0)
dev = torch.device("cuda", 0)
tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
data = make_batch(cfg["model"]["inputs"], 4, device=dev)
labels = make_labels(4, device=dev)
if os.environ.get("GRAPHS", "1") == "1":
    tr.enable_graphs(data)
for _ in range(5):
    tr.train_step(data, labels)
torch.cuda.synchronize()
names = ["reset", "encoders + fpn + emb fwd", "fuser fwd", "loss (incl. Hungarian sync)", "backward", "finish", "optimizer"]
N = 10
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)] for _ in range(N)]
host = [[0.0] * (len(names) + 1) for _ in range(N)]
m = tr.model
wall0 = time.perf_counter()
for it in range(N):
    k = 0
    def mark():
        global k
        ev[it][k].record(); host[it][k] = time.perf_counter(); k += 1
    m.train()
    mark()
    tr.reducer.reset(); mark()
    feats = m._encode_views(data); mark()      # encoders + FPN + embedding of all views
    out = m.querent(data)
    g = m.__dict__.get("_graphed_fuser")
    shp = {i: data[f"{i}_shape"] for i in m.inputs}
    if g is not None:
        out = g(feats, shp, m._get_projetions(m.inputs, data), out)
    else:
        out = m.fuser(batch=[feats[i] for i in m.inputs], shape=[data[f"{i}_shape"][:, :2] for i in m.inputs],
                      projection=m._get_projetions(m.inputs, data), out=out)
    mark()
    loss, _ = tr.loss_fn(out, labels); mark()
    loss.backward(); mark()
    tr.reducer.finish(); mark()
    if isinstance(tr.optimizer, FusedAdamW):
        tr.optimizer.set_active(tr.reducer.seen_ids())
    tr.optimizer.step(); mark()
torch.cuda.synchronize()
wall = (time.perf_counter() - wall0) / N * 1e3
print(f"wall {wall:.2f} ms/step")
print(f"{'phase':34s} {'gpu ms':>8s} {'host ms':>8s}")
for j, n in enumerate(names):
    g_ = sum(ev[it][j].elapsed_time(ev[it][j + 1]) for it in range(2, N)) / (N - 2)
    h_ = sum(host[it][j + 1] - host[it][j] for it in range(2, N)) / (N - 2) * 1e3
    print(f"{n:34s} {g_:8.2f} {h_:8.2f}")
gs = sum(ev[it][0].elapsed_time(ev[it][-1]) for it in range(2, N)) / (N - 2)
print(f"{'step (event 0 -> last)':34s} {gs:8.2f}")
