#!/bin/bash
# Host pacing A/B (profiles/r05_loss_window_ab.txt): DPFT_PACE_HOST=auto|0|1 on the three single-GPU configurations, three rounds.
# usage (one gpurun call): bash tools/pace_ab.sh  -> gpurun_out/pace_ab.txt
mkdir -p gpurun_out
{
for cfg in "f32 4" "bf16 4" "bf16 8"; do set -- $cfg; dt=$1; b=$2
for r in 1 2 3; do for v in auto 0 1; do
  echo "== $dt batch $b round $r PACE=$v"
  DPFT_PACE_HOST=$v timeout 900 python bench.py --dtype $dt --batch $b --steps 40 --warmup 5 --no-cpu-baseline --latency-reps 5 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('value','ms_per_step','loss_window_us','host_pacing')})"
done; done; done
} > gpurun_out/pace_ab.txt 2>&1
