#!/bin/bash
mkdir -p gpurun_out
{
for r in 1 2 3; do for v in none 3 2 1 0; do
  echo "== round $r pace2=$v"
  unset DPFT_EXP_PACE2; if [ $v != none ]; then export DPFT_EXP_PACE2=$v; fi
  timeout 900 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --latency-reps 5 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('value','ms_per_step','loss_window_us')})"
done; done
} > gpurun_out/pace2_ab.txt 2>&1
