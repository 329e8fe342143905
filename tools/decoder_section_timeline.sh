# Timeline (start offset, duration, gap to the previous kernel) of the serial decoder / loss section of one plain training step.
cd /tmp && export TMPDIR=/tmp
STEPS=2 timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/proft7 -- python /root/repo/tools/train_only.py </dev/null > /tmp/proft7.log 2>&1
f=$(find /tmp/proft7 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
step = rows[idx[-2] + 1: idx[-1] + 1]
def short(n):
    n = n.split("(")[0].replace("void ", "")
    n = re.sub(r"at::native::(\(anonymous namespace\)::)?", "aten:", n)
    return re.sub(r"<.*", "", n)[:40]
dec = [i for i, r in enumerate(step) if any(p in r["Kernel_Name"] for p in ("sa_train", "xf_train", "hd_train", "set_loss", "match_cost"))]
lo, hi = dec[0] - 18, dec[-1] + 14
t0 = int(step[lo]["Start_Timestamp"]); prev_end = t0
for r in step[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?')}  {short(r['Kernel_Name'])}")
    prev_end = max(prev_end, e)
PY
