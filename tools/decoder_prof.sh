# decoder-only kernel stats: bash tools/decoder_prof.sh <tag>
TAG=${1:-dec}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dec -- python /root/repo/tools/decoder_only.py </dev/null > $OUT/${TAG}_decoder.log 2>&1
f=$(find /tmp/p_dec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "Name\|decoder_\|pack_" "$f" | cut -c1-200 > $OUT/${TAG}_decoder_kernel_stats.csv
grep decoder_fwd_us $OUT/${TAG}_decoder.log; cat $OUT/${TAG}_decoder_kernel_stats.csv | cut -d, -f1-4
REPS=50 python /root/repo/tools/decoder_only.py 2>&1 | tail -1
