# Round-2 evidence run (one gpurun call): serialized + plain step kernel traces, decoder kernel trace + PMC traffic.
# usage: bash tools/r02_profile.sh <tag>      -> gpurun_out/<tag>_*
TAG=${1:-r02}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (1) serialized steps: every conv kernel alone on the device (dpft_profile_serialize), no event brackets
SERIAL=1 STEPS=10 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_serial -- python /root/repo/tools/train_only.py </dev/null > $OUT/${TAG}_serial.log 2>&1
f=$(find /tmp/p_serial -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${TAG}_serialized_step_kernel_stats.csv
# (2) plain steps (concurrent view streams + side-stream weight gradients)
STEPS=20 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_plain -- python /root/repo/tools/train_only.py </dev/null > $OUT/${TAG}_plain.log 2>&1
f=$(find /tmp/p_plain -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${TAG}_train_step_kernel_stats.csv
# (3) inference decoder: kernel trace, then FETCH_SIZE / WRITE_SIZE in separate passes
REPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dec -- python /root/repo/tools/decoder_only.py </dev/null > $OUT/${TAG}_decoder.log 2>&1
f=$(find /tmp/p_dec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "Name\|decoder_\|pack_" "$f" > $OUT/${TAG}_decoder_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  REPS=5 timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/p_dec_$c -- python /root/repo/tools/decoder_only.py </dev/null > /tmp/p_dec_$c.log 2>&1
done
python - <<PY
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/p_dec_{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        out[c] = "no counter file"; continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") != c: continue
        n = r["Kernel_Name"].split("(")[0]
        if "decoder_" not in n: continue
        agg[n][0] += float(r["Counter_Value"]); agg[n][1] += 1
    out[c] = {k: {"sum": v[0], "launches": v[1], "per_launch": v[0] / max(v[1], 1)} for k, v in agg.items()}
json.dump(out, open("$OUT/${TAG}_decoder_pmc_raw.json", "w"), indent=1)
print(json.dumps(out)[:2000])
PY
tail -2 $OUT/${TAG}_serial.log $OUT/${TAG}_plain.log $OUT/${TAG}_decoder.log
