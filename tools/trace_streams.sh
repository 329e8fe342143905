cd /tmp && export TMPDIR=/tmp
STEPS=2 timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/proft3 -- python /root/repo/tools/train_only.py </dev/null > /tmp/proft3.log 2>&1
f=$(find /tmp/proft3 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "adamw_kernel" in n]
lo, hi = idx[-2] + 1, idx[-1] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in step)
print("step span ms", (t1 - t0) / 1e6)
key = "Stream_Id" if "Stream_Id" in step[0] else "Queue_Id"
per = collections.defaultdict(lambda: [0, 0, 10**30, 0])
for r in step:
    k = r[key]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    p = per[k]; p[0] += e - s; p[1] += 1; p[2] = min(p[2], s); p[3] = max(p[3], e)
for k, p in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print(f"{key} {k}: busy {p[0]/1e6:.2f} ms, {p[1]} kernels, active window {(p[2]-t0)/1e6:.2f}..{(p[3]-t0)/1e6:.2f} ms")
# timeline in 2 ms buckets: number of distinct streams busy & busy fraction of the top stream
PY
