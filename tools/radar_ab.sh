# radar projection: merged doppler-major fold (default) vs the two separate fold launches, per-kernel times
cd /root/repo
timeout 300 python -m pytest tests/test_data_pipeline.py -m gpu -x -q -k radar 2>&1 | tail -2
for v in 1 0 1 0; do
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p_radar
  DPFT_RADAR_FOLD_BOTH=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_radar -- python /root/repo/tools/radar_bench.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('both=$v', {k: round(d[k],1) for k in ('gpu_ms_per_cube','achieved_GBs_algorithmic','torch_sum_one_read_us','torch_sum_GBs')} | {'gpu_us_per_cube': round(d['gpu_ms_per_cube']*1e3,1)})"
  f=$(find /tmp/p_radar -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "radar" in r["Name"]:
        print(f'   {r["Name"].split("(")[0]:50s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
  cd /root/repo
done
