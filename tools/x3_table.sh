cd /root/repo
export PYTHONUNBUFFERED=1
DPFT_CONV_TABLE=gpurun_out/table_f32.txt python bench.py --steps 10 --warmup 5 --no-cpu-baseline --latency-reps 5 2>/dev/null | tail -n 1 > gpurun_out/tb_f32.json
DPFT_CONV_TABLE=gpurun_out/table_x3.txt python bench.py --steps 10 --warmup 5 --no-cpu-baseline --latency-reps 5 --dtype f32x3 2>/dev/null | tail -n 1 > gpurun_out/tb_x3.json
