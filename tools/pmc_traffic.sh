# HBM traffic of the conv kernels from PMC counters (separate passes, MI355X_MICROARCH.md "HBM" section)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  STEPS=2 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python /root/repo/tools/train_only.py </dev/null > /tmp/pmc_$c.log 2>&1
done
mkdir -p /root/repo/gpurun_out
python - <<'PY'
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counter file for", c); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") != c: continue
        n = r["Kernel_Name"]
        key = "igemm" if "igemm" in n else "wgrad" if "wgrad_" in n else "splitk" if "splitk" in n else None
        if key is None: continue
        agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
    out[c] = {k: {"sum": v[0], "launches": v[1]} for k, v in agg.items()}
json.dump(out, open("/root/repo/gpurun_out/pmc_conv_traffic_raw.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
head -3 $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) | cut -c1-400
