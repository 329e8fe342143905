#!/bin/bash
{
for r in 1 2; do for v in "" "1,1,0.8,16" "1,1.1,0.8,16" "1,0.92,0.6,16" "0.8,1,0.9,16" "1,0.92,1.0,16"; do
  echo "eff=[$v] batch=1 $(DPFT_TILE_EFF=$v BATCH=1 REPS=300 python tools/infer_only.py 2>/dev/null | tail -1)"
done; done
} > gpurun_out/infer_ab.txt 2>&1
