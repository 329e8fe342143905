# per-kernel A/B of two library builds inside training steps: bash tools/kernel_ab.sh '<grep pattern>'
cd /tmp; export TMPDIR=/tmp
cp /root/repo/dpft_amd/libdpft_hip.so /tmp/lib_keep.so
for v in old new; do cp /root/repo/tools/ab/lib_$v.so /root/repo/dpft_amd/libdpft_hip.so; rm -rf /tmp/p_x
STEPS=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_x -- python /root/repo/tools/train_only.py > /dev/null 2>&1
f=$(find /tmp/p_x -name "*kernel_stats.csv" | head -1); echo "== $v"
python - "$f" "$1" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(f'  {r["Name"].split("(")[0][:50]:50s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:8.1f} us')
PY
done
cp /tmp/lib_keep.so /root/repo/dpft_amd/libdpft_hip.so
