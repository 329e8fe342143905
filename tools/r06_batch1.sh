# round 6, batch 1: new parity gates, bench line with the pipe-correct fields + secondary legs, stream-priority A/B, BN-finalize A/B
cd /root/repo; export PYTHONUNBUFFERED=1
O=gpurun_out/r6b; mkdir -p $O
python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -q -s -m gpu -k "full_size_train_step or halfspace" > $O/parity.txt 2>&1; tail -3 $O/parity.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
q() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --latency-reps 5 2>/dev/null | grep "^{" | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'mean', round(d['ms_per_step'],2), 'median', round(d['step_ms_median'],2), 'min', round(d['step_ms_min'],2))"; }
for i in 1 2 3; do
  q "prio_normal_$i"; DPFT_STREAM_PRIORITY=low q "prio_low_$i"
done > $O/prio_ab.txt 2>&1
cat $O/prio_ab.txt
for i in 1 2; do
  q "bnfuse_off_$i"; DPFT_BN_FINAL_FUSE=2 q "bnfuse_slab_$i"
done > $O/bnfuse_ab.txt 2>&1
cat $O/bnfuse_ab.txt
