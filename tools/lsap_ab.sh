#!/bin/bash
mkdir -p gpurun_out
{
for r in 1 2 3; do for v in none predec predecb loss; do
  echo "== round $r $v"
  export DPFT_LSAP_DEV=1; export DPFT_EXP_SYNC=$v; unset DPFT_EXP_SYNC_AT
  if [ $v = predecb ]; then export DPFT_EXP_SYNC=predec; export DPFT_EXP_SYNC_AT=before; fi
  timeout 900 python bench.py --steps 60 --warmup 10 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('value','ms_per_step','loss_window_us')})"
done; done
} > gpurun_out/lsap_ab.txt 2>&1
