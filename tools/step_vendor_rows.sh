# Vendor / ATen kernels INSIDE one steady-state training step (between two optimizer launches of a kernel trace): the
# whole-run statistics also count the set-up copies and fills (parameter flattening, bucket and arena initialisation).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/proft6
STEPS=4 timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/proft6 -- python /root/repo/tools/train_only.py </dev/null > /tmp/proft6.log 2>&1
f=$(find /tmp/proft6 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
steps = [rows[idx[k] + 1: idx[k + 1] + 1] for k in range(len(idx) - 3, len(idx) - 1)]
for step in steps:
    agg = collections.OrderedDict()
    tot_n = tot_t = 0
    for r in step:
        n = r["Kernel_Name"]
        if "dpft::" in n: continue
        key = n.split("(")[0][:130]
        if "elementwise" in n or "reduce_kernel" in n or "multi_tensor" in n:
            key = n[:200]
        a = agg.setdefault(key, [0, 0.0])
        dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a[0] += 1; a[1] += dt; tot_n += 1; tot_t += dt
    print(f"== one step: {len(step)} kernels, {tot_n} vendor / ATen launches, {tot_t:.1f} us")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{c:4d} {t:8.1f} us  {k}")
PY
