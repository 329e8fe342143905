cd /tmp && export TMPDIR=/tmp
DPFT_CONV_COMPUTE=bf16 SERIAL=1 STEPS=10 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bf -- python /root/repo/tools/train_only.py </dev/null > /root/repo/gpurun_out/r02_bf16_serial.log 2>&1
f=$(find /tmp/p_bf -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" /root/repo/gpurun_out/r02_bf16_serialized_kernel_stats.csv
python /root/repo/tools/roofline_from_rocprof.py /root/repo/gpurun_out/r02_bf16_serialized_kernel_stats.csv 13 | head -16
DPFT_CONV_COMPUTE=bf16 STEPS=20 python /root/repo/tools/train_only.py | tail -1
