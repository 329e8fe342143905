# Timeline of the serial decoder / loss section of a training step (between the encoders' forward and backward).
cd /tmp && export TMPDIR=/tmp
STEPS=3 timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/proft7 -- python /root/repo/tools/train_only.py </dev/null > /tmp/proft7.log 2>&1
f=$(find /tmp/proft7 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re, collections, os
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
step = rows[idx[-2] + 1: idx[-1] + 1]
def short(n):
    n = n.split("(")[0].replace("void ", "")
    n = re.sub(r"at::native::(\(anonymous namespace\)::)?", "aten:", n)
    return re.sub(r"<.*", "", n)[:40]
first = next(i for i, r in enumerate(step) if "hd_train_fwd" in r["Kernel_Name"] or "sa_train_fwd" in r["Kernel_Name"])
last = max(i for i, r in enumerate(step) if "sa_train_bwd" in r["Kernel_Name"])
sec = step[first - int(os.environ.get("BEFORE", "30")):last + int(os.environ.get("AFTER", "12"))]
t0 = int(sec[0]["Start_Timestamp"])
agg = collections.OrderedDict()
prev_end = None
tot_busy = 0
for r in sec:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = short(r["Kernel_Name"])
    gap = 0 if prev_end is None else max(0, s - prev_end)
    a = agg.setdefault(n, [0, 0, 0]); a[0] += 1; a[1] += e - s; a[2] += gap
    prev_end = max(prev_end or 0, e); tot_busy += e - s
span = (max(int(r["End_Timestamp"]) for r in sec) - t0) / 1e3
print(f"serial section: {len(sec)} kernels, span {span:.0f} us, kernel time {tot_busy/1e3:.0f} us")
import os
if os.environ.get("SEQ") == "1":      # the launches in order: start (us), duration, idle before, stream
    pe = None
    for r in sec:
        s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"   {(s_ - t0)/1e3:8.1f} {(e_ - s_)/1e3:7.1f} us  gap {0 if pe is None else (s_ - pe)/1e3:6.1f}  q{r.get('Queue_Id', '?')}  {short(r['Kernel_Name'])}")
        pe = max(pe or 0, e_)
for n, (c, t, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"  {n:40s} x{c:3d}  kernel {t/1e3:7.1f} us  idle before {g/1e3:7.1f} us")
PY
