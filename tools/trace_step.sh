cd /tmp && export TMPDIR=/tmp
STEPS=2 timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/proft2 -- python /root/repo/tools/train_only.py </dev/null > /tmp/proft2.log 2>&1
f=$(find /tmp/proft2 -name "*kernel_trace.csv" | head -1)
mkdir -p /root/repo/gpurun_out
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last step = after the last adamw kernel but one
idx = [i for i, n in enumerate(names) if "adamw_kernel" in n]
lo, hi = idx[-2] + 1, idx[-1] + 1
def short(n):
    n = re.sub(r"at::native::|\(anonymous namespace\)::|void ", "", n)
    return n[:90]
out = []
prev, cnt = None, 0
for r in rows[lo:hi]:
    s = short(r["Kernel_Name"]) + " q" + r.get("Queue_Id", "?")
    if s == prev:
        cnt += 1
    else:
        if prev is not None:
            out.append(f"{cnt:4d} x {prev}")
        prev, cnt = s, 1
out.append(f"{cnt:4d} x {prev}")
open("/root/repo/gpurun_out/step_sequence.txt", "w").write("\n".join(out))
print(len(rows[lo:hi]), "kernels in the step;", len(out), "runs")
PY
