"""Which hardware queue should each concurrent chain of the training step sit on?  One process, one model: the five
movable chains (bev view, front view, camera / bev / front weight-gradient streams) are re-assigned to the four queue
classes (0 = the main stream's, 1..3 = dpft_stream_set) and a few steps are timed per assignment.  Assignments that
only permute classes 1..3 are measured once.   python tools/stream_search.py [steps]  -> gpurun_out/r03/stream_search.txt"""
import itertools, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
from dpft_amd.configs import load_config
from dpft_amd.hip.lib import stream_set
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
cfg = load_config("kradar")
torch.manual_seed(42)
tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
data = make_batch(cfg["model"]["inputs"], 4, seed=42, device=dev)
labels = make_labels(4, seed=42, device=dev)
tr.enable_graphs(data)
for _ in range(3):
    tr.train_step(data, labels)
streams, distinct = stream_set(dev, 3)
print("distinct hardware queues besides the main one:", distinct, flush=True)
main = torch.cuda.current_stream()
cls = [main] + list(streams)
m = tr.model
names = ["bev_view", "front_view", "cam_side", "bev_side", "front_side"]


def canon(a):      # relabel classes 1..3 in order of first appearance
    mp, nxt, out = {0: 0}, 1, []
    for v in a:
        if v not in mp:
            mp[v] = nxt
            nxt += 1
        out.append(mp[v])
    return tuple(out)


def measure(a):
    m.__dict__["_view_streams"] = [cls[a[0]], cls[a[1]]]
    for inp, k in zip(m.inputs, (a[2], a[3], a[4])):
        m.backbones[inp].side_stream = cls[k]
    for _ in range(2):
        tr.train_step(data, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(data, labels)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


seen, res = set(), []
for a in itertools.product(range(4), repeat=5):
    c = canon(a)
    if c in seen or max(c) > distinct:
        continue
    seen.add(c)
    res.append((measure(c), c))
    print(f"{res[-1][0]:7.2f} ms  " + " ".join(f"{n}={v}" for n, v in zip(names, c)), flush=True)
res.sort()
os.makedirs("gpurun_out/r03", exist_ok=True)
with open("gpurun_out/r03/stream_search.txt", "w") as f:
    f.write("ms_per_step " + " ".join(names) + "   (0 = main stream's hardware queue)\n")
    for t, c in res:
        f.write(f"{t:7.2f} " + " ".join(str(v) for v in c) + "\n")
print("best:", res[:8])
# re-measure the best few and the reference placements with more steps
steps = 3 * steps
for t, c in res[:6] + [r for r in res if r[1] in ((1, 2, 3, 1, 2), (1, 2, 2, 1, 3))]:
    print(f"recheck {measure(c):7.2f} ms (was {t:.2f}) {c}", flush=True)
