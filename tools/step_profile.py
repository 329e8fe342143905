"""Wall-clock phase breakdown of one training step (each phase closed by a device sync)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer

cfg = load_config("kradar")
torch.manual_seed(0)
dev = torch.device("cuda", 0)
tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
data = make_batch(cfg["model"]["inputs"], 4, device=dev)
labels = make_labels(4, device=dev)
if os.environ.get("GRAPHS", "1") == "1":
    tr.enable_graphs(data)
for _ in range(3):
    tr.train_step(data, labels)
torch.cuda.synchronize()

def t():
    torch.cuda.synchronize()
    return time.perf_counter()

acc = {}
for it in range(5):
    tr.model.train()
    t0 = t(); tr.reducer.reset(); t1 = t()
    # forward pieces
    m = tr.model
    feats = {}
    ta = t()
    for i in m.inputs:
        feats[i] = m.backbones[i](data[i])
    tb = t()
    for i in m.inputs:
        f = m._add_raw_data(feats[i], data[i]); feats[i] = m.embeddings[i](m.necks[i](f))
    tc = t()
    out = m.querent(data)
    g = m.__dict__.get("_graphed_fuser")
    if g is not None:
        out = g(feats, {i: data[f"{i}_shape"] for i in m.inputs}, m._get_projetions(m.inputs, data), out)
    else:
        out = m.fuser(batch=[feats[i] for i in m.inputs], shape=[data[f"{i}_shape"][:, :2] for i in m.inputs],
                      projection=m._get_projetions(m.inputs, data), out=out)
    td = t()
    loss, _ = tr.loss_fn(out, labels); te = t()
    loss.backward(); tf = t()
    tr.reducer.finish(); tr.optimizer.step(); tg = t()
    for k, v in (("reset", t1 - t0), ("backbones fwd", tb - ta), ("fpn+emb fwd", tc - tb), ("fuser fwd", td - tc),
                 ("loss", te - td), ("backward", tf - te), ("optimizer", tg - tf), ("total", tg - t0)):
        acc[k] = acc.get(k, 0.0) + v
for k, v in acc.items():
    print(f"{k:16s} {1e3 * v / 5:8.2f} ms")
