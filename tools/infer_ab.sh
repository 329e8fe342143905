#!/bin/bash
{
for r in 1 2 3; do for v in 0 4 2; do for b in 1 4; do
  echo "narrow=$v batch=$b $(DPFT_X3_NARROW=$v BATCH=$b REPS=200 python tools/infer_only.py 2>/dev/null | tail -1)"
done; done; done
for v in 0 4 2; do
  echo "== train narrow=$v"; DPFT_X3_NARROW=$v timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --latency-reps 5 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step','step_ms_median')})"
done
} > gpurun_out/infer_ab.txt 2>&1
