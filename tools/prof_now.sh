OUT=/root/repo/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
STEPS=20 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_plain -- python /root/repo/tools/train_only.py </dev/null > $OUT/now_plain.log 2>&1
f=$(find /tmp/p_plain -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/now_train_step_kernel_stats.csv
python /root/repo/tools/roofline_from_rocprof.py $OUT/now_train_step_kernel_stats.csv 23
grep ms/step $OUT/now_plain.log
