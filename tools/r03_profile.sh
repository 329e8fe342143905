# Round-3 evidence run (one gpurun call).  usage: bash tools/r03_profile.sh [quick]  -> gpurun_out/r03p/*  (copy into profiles/)
OUT=/root/repo/gpurun_out/r03p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (1) serialized steps: every conv kernel alone on the device (dpft_profile_serialize), no event brackets
SERIAL=1 STEPS=10 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_serial -- python /root/repo/tools/train_only.py </dev/null > $OUT/serial.log 2>&1
f=$(find /tmp/p_serial -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/r03_serialized_step_kernel_stats.csv
python /root/repo/tools/roofline_from_rocprof.py $OUT/r03_serialized_step_kernel_stats.csv 13 > $OUT/r03_roofline_from_rocprof.json
# (2) plain steps (concurrent view streams + side-stream weight gradients)
STEPS=20 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_plain -- python /root/repo/tools/train_only.py </dev/null > $OUT/plain.log 2>&1
f=$(find /tmp/p_plain -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/r03_train_step_kernel_stats.csv
python /root/repo/tools/roofline_from_rocprof.py $OUT/r03_train_step_kernel_stats.csv 23 > $OUT/r03_roofline_from_rocprof_plain_steps.json
tail -2 $OUT/serial.log; grep "ms/step" $OUT/plain.log; cat $OUT/r03_roofline_from_rocprof.json
[ "$1" = "quick" ] && exit 0
