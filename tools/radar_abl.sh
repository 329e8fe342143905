# per-kernel time of the radar projection for the ablation builds tools/ab/lib_v{0,1,2,4,7}.so (RADAR_ABL bits: 1 no median, 2 no variance, 4 no log)
cd /root/repo; cp dpft_amd/libdpft_hip.so /tmp/lib_keep.so
for v in 0 1 2 4 7; do
  cp tools/ab/lib_v$v.so dpft_amd/libdpft_hip.so
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p_radar
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_radar -- python /root/repo/tools/radar_bench.py 2>/dev/null | tail -1 | cut -c1-120
  f=$(find /tmp/p_radar -name "*kernel_stats.csv" | head -1)
  python - "$f" $v <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "radar" in r["Name"]:
        print(f'v{sys.argv[2]} {r["Name"].split("(")[0]:50s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
  cd /root/repo
done
cp /tmp/lib_keep.so dpft_amd/libdpft_hip.so
