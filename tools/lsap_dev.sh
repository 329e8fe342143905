#!/bin/bash
# device assignments: parity tests, trainer tests, bench with and without (tools/lsap_dev.sh -> gpurun_out/lsap_dev.txt)
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_lsap.py -x -q 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_trainer.py -x -q 2>&1 | tail -5
for v in 1 0; do
  echo "== DPFT_LSAP_DEV=$v"
  DPFT_LSAP_DEV=$v timeout 900 python bench.py --steps 40 --warmup 10 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','loss_window_us','loss_window_assignments','loss')})"
done
echo "== forced collectives"
for v in 1 0; do
DPFT_LSAP_DEV=$v timeout 900 python bench.py --steps 40 --warmup 10 --force-collectives 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','loss')})"
done
} > gpurun_out/lsap_dev.txt 2>&1
