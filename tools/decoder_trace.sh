# per-dispatch timeline of the last decoder forward: bash tools/decoder_trace.sh <tag>
TAG=${1:-dec}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_dect
REPS=${REPS:-6} timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_dect -- python /root/repo/tools/decoder_only.py </dev/null > $OUT/${TAG}_dtrace.log 2>&1
f=$(find /tmp/p_dect -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/${TAG}_decoder_timeline.txt
import csv, sys, re
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "decoder_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = 9
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = None
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void dpft::", "").replace("dpft::", "")
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{name:32s} start {(s - t0) / 1e3:8.2f} us  dur {(e - s) / 1e3:7.2f} us  gap {gap:6.2f} us  grid {r.get('Grid_Size','?')} wg {r.get('Workgroup_Size','?')} lds {r.get('LDS_Block_Size','?')} vgpr {r.get('VGPR_Count','?')}")
    prev_end = e
print("forward span", (int(last[-1]["End_Timestamp"]) - t0) / 1e3, "us")
PY
