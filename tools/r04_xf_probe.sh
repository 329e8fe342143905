# xf_train_bwd / xf_scatter_small: where the time goes (round 4).  Per-kernel averages from rocprofv3 over tools/train_only.py.
OUT=/root/repo/gpurun_out/r04c; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for tag in ${TAGS:-scatter scatter_noldsatomics scatter_noflush scatter_norows}; do
  case $tag in default) E="";; noatomics) E="DPFT_XF_EXP=1";; scatter) E="DPFT_XF_SCATTER=1";; scatter_noldsatomics) E="DPFT_XF_SCATTER=1 DPFT_XF_EXP=2";;
    scatter_noflush) E="DPFT_XF_SCATTER=1 DPFT_XF_EXP=4";; scatter_norows) E="DPFT_XF_SCATTER=1 DPFT_XF_EXP=8";; esac
  env $E STEPS=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xf_$tag -- python /root/repo/tools/train_only.py </dev/null > $OUT/xf_$tag.log 2>&1
  f=$(find /tmp/xf_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("xf_", "sa_train", "hd_train", "rows_outer", "scatter_small")):
        print(f"  {n.split('(')[0][-48:]:48s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:7.1f} us  min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f}")
PY
done > $OUT/xf_probe.txt 2>&1
cat $OUT/xf_probe.txt
