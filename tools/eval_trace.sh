# kernel timeline of eval forwards (the fwd ms/frame protocol): per-queue busy time, gaps, categories.  bash tools/eval_trace.sh
cd /tmp && export TMPDIR=/tmp
cat > /tmp/eval_only.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch
cfg = load_config("kradar"); torch.manual_seed(0); dev = torch.device("cuda", 0)
m = build("dprt", cfg).to(dev).eval()
data = make_batch(cfg["model"]["inputs"], 4, device=dev)
with torch.no_grad():
    for _ in range(5): m(data)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); m(data); t1 = time.perf_counter(); torch.cuda.synchronize(); ts.append((t1 - t0, time.perf_counter() - t0))
print("host issue ms %.2f  wall ms %.2f" % (1e3 * sum(a for a, _ in ts) / 20, 1e3 * sum(b for _, b in ts) / 20))
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/proft6 -- python /tmp/eval_only.py 2>/dev/null | tail -1
f=$(find /tmp/proft6 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last forward = kernels after the second-to-last decoder_scores_head (final launch of a forward) ... use last 1/25th by xattn count
idx = [i for i, r in enumerate(rows) if "decoder_xattn_kernel" in r["Kernel_Name"]]
end = idx[-1] + 2
prev_end = idx[-5] + 2          # 4 xattn launches per forward
step = rows[prev_end:end]
t0 = int(step[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in step)
print(f"one eval forward: {len(step)} kernels, span {(t1-t0)/1e6:.2f} ms")
key = "Stream_Id" if "Stream_Id" in step[0] else "Queue_Id"
per = collections.defaultdict(list)
for r in step: per[r[key]].append(r)
for k, v in sorted(per.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in v)
    s, e = min(int(r["Start_Timestamp"]) for r in v), max(int(r["End_Timestamp"]) for r in v)
    print(f"  {key} {k}: {len(v)} kernels busy {busy/1e6:.2f} ms window {(s-t0)/1e6:.2f}..{(e-t0)/1e6:.2f}")
cat = collections.defaultdict(lambda: [0, 0])
for r in step:
    n = r["Kernel_Name"].split("(")[0]
    c = "conv" if ("igemm" in n or "conv16" in n) else "bn/pool" if "bn_" in n else "decoder" if "decoder_" in n else "aten/rt" if ("at::" in n or "rocclr" in n) else n[-30:]
    cat[c][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cat[c][1] += 1
for c, (t, n) in sorted(cat.items(), key=lambda kv: -kv[1][0])[:12]: print(f"  {c:32s} {t/1e6:6.2f} ms {n:4d}")
PY
