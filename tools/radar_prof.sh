# correctness + per-kernel time of the radar projection: bash tools/radar_prof.sh
cd /root/repo
timeout 300 python -m pytest tests/test_data_pipeline.py -m gpu -x -q -k radar 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_radar -- python /root/repo/tools/radar_bench.py 2>/dev/null | tail -1
f=$(find /tmp/p_radar -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "radar" in r["Name"]:
        print(f'{r["Name"].split("(")[0]:60s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
