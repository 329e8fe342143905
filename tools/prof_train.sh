cd /tmp && export TMPDIR=/tmp
STEPS=20 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft -- python /root/repo/tools/train_only.py </dev/null > /tmp/proft.log 2>&1
f=$(find /tmp/proft -name "*kernel_stats.csv" | head -1)
mkdir -p /root/repo/gpurun_out
if [ -n "$f" ]; then cp "$f" /root/repo/gpurun_out/train_kernel_stats.csv; fi
grep "ms/step" /tmp/proft.log
