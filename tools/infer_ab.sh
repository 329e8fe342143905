#!/bin/bash
# Inference A/Bs of round 5 (one gpurun call)  -> gpurun_out/infer_ab.txt
#  DPFT_BNACT_FIXUP=0|1  K-split convs: conv + reduction + elementwise pass | BatchNorm / residual / ReLU epilogue inside the launch (default)
#  DPFT_X3_NARROW=0|2|4  split kernels: 128 x 128 tiles always | 128 x 64 tiles where the K split would be deeper than 2 (default) / 4
#  DPFT_TILE_EFF=a,b,c,rows  weights of the 128x128 / 128x64 / 64x64 candidates for problems of <= rows row tiles (sweep: within noise)
{
for r in 1 2 3; do
  for v in 1 0; do for b in 1 4; do
    echo "bnact_fixup=$v batch=$b $(DPFT_BNACT_FIXUP=$v BATCH=$b REPS=200 python tools/infer_only.py 2>/dev/null | tail -1)"
  done; done
  for v in 0 4 2; do for b in 1 4; do
    echo "x3_narrow=$v batch=$b $(DPFT_X3_NARROW=$v BATCH=$b REPS=200 python tools/infer_only.py 2>/dev/null | tail -1)"
  done; done
done
for v in 0 2; do
  echo "== train x3_narrow=$v"; DPFT_X3_NARROW=$v timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --latency-reps 5 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step','step_ms_median')})"
done
} > gpurun_out/infer_ab.txt 2>&1
