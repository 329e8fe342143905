# same-box A/B of bf16 activation storage: bash tools/ab_bf16.sh [batch]
B=${1:-4}
for round in 1 2 3; do for a in 0 1; do
DPFT_ACT16=$a DPFT_CONV_TABLE=/root/repo/gpurun_out/r02_conv_table_bf16_act$a.txt timeout 300 python /root/repo/bench.py --dtype bf16 --batch $B --steps ${STEPS:-40} --warmup 10 --no-cpu-baseline --latency-reps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('act16=$a', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,1) for k,v in d['roofline']['per_kind_tflops'].items()})"
done; done
