cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python /root/repo/bench.py --steps 3 --warmup 2 --no-cpu-baseline </dev/null > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
mkdir -p /root/repo/gpurun_out
if [ -n "$f" ]; then grep -i "decoder\|Name" "$f" > /root/repo/gpurun_out/decoder_stats.csv; cp "$f" /root/repo/gpurun_out/kernel_stats_v3.csv; fi
tail -2 /tmp/prof.log | cut -c1-300
