cd /root/repo
export PYTHONUNBUFFERED=1
DPFT_CONV_TABLE=gpurun_out/table_f32.txt python bench.py --steps 10 --warmup 5 --no-cpu-baseline --latency-reps 5 2>/dev/null | tail -n 1 > gpurun_out/tb_f32.json
DPFT_CONV_TABLE=gpurun_out/table_x3.txt python bench.py --steps 10 --warmup 5 --no-cpu-baseline --latency-reps 5 --dtype f32x3 2>/dev/null | tail -n 1 > gpurun_out/tb_x3.json

python bench.py --steps 40 --warmup 10 --no-cpu-baseline --latency-reps 30 --dtype f32x3 2>/dev/null | tail -n 1 > gpurun_out/step_x3_3.json
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --latency-reps 30 2>/dev/null | tail -n 1 > gpurun_out/step_f32_3.json
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --latency-reps 30 --dtype f32x3 2>/dev/null | tail -n 1 > gpurun_out/step_x3_4.json
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --latency-reps 30 2>/dev/null | tail -n 1 > gpurun_out/step_f32_4.json
