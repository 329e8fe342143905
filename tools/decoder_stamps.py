"""Phase timeline inside the inference decoder's kernels (DPFT_DEC_DBG=1024): per-phase start / end over all blocks."""
import os, sys
os.environ["DPFT_DEC_DBG"] = "1024"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import decoder_runner
from dpft_amd.configs import load_config
from dpft_amd.hip.lib import lib
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch

cfg = load_config("kradar")
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = build("dprt", cfg).to(dev)
data = make_batch(cfg["model"]["inputs"], 4, device=dev)
run, _ = decoder_runner(model, data)
for _ in range(5):
    run()
torch.cuda.synchronize()
buf = np.zeros(2 * 2048 * 8, dtype=np.uint64)
lib.call("dpft_debug_decoder_stamps", buf.ctypes.data)
st = buf.reshape(2, 2048, 8).astype(np.float64) * 0.01      # us
# the LAST launches: scores_head = final heads-only launch overwrote head slots; scores from the last full one
def summarize(name, arr, nslot):
    ok = arr[:, 0] > 0
    a = arr[ok]
    if not len(a):
        print(name, "no stamps"); return
    t0 = a[:, 0].min()
    print(f"{name}: {len(a)} blocks, first block start = 0")
    for s in range(nslot):
        v = a[:, s][a[:, s] > 0] - t0
        if len(v):
            print(f"   slot {s}: min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():7.2f} us")
summarize("scores blocks (kernel 0, ids < 1024)", st[0, :1024], 5)
summarize("head blocks   (kernel 0, ids >= 1024)", st[0, 1024:], 4)
summarize("xattn blocks  (kernel 1)", st[1], 6)
nb = int((st[1][:, 0] > 0).sum())
per = nb // 3
t0 = st[1][:nb, 0].min()
for v in range(3):
    a = st[1][v * per:(v + 1) * per]
    print(f"  view {v}: " + "  ".join(f"s{s}: med {np.median(a[:, s] - t0):6.2f} max {(a[:, s] - t0).max():6.2f}" for s in range(6)))
# per-CU load: blocks are placed round-robin, so block b and b + 256 ... share a CU (approximately)
end = st[1][:nb, 5] - t0
print("  end-time histogram (us):", np.histogram(end, bins=8)[0].tolist(), [round(float(x), 1) for x in np.histogram(end, bins=8)[1]])
