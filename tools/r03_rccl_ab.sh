#!/bin/bash
# Plain N=1 step vs the same step with every bucket forced through a one-rank RCCL communicator, and the stream-budget
# variants (shared radar view stream / shared weight-gradient side stream).  Writes gpurun_out/r03_rccl_ab/*.json
set -u
OUT=gpurun_out/r03_rccl_ab; mkdir -p $OUT
run() { name=$1; shift; env "$@" python bench.py --gpus 1 --steps 30 --warmup 8 --latency-reps 20 --no-cpu-baseline $EXTRA > $OUT/$name.json 2> $OUT/$name.err; tail -c 300 $OUT/$name.err; python - <<PY
import json
try:
    l=json.loads([x for x in open("$OUT/$name.json") if x.startswith("{")][-1])
    print("$name", round(l["ms_per_step"],2), "ms", round(l["value"],1), "samples/s exposed", round(l["exposed_allreduce_ms"],3), "forced", l.get("collectives_forced"))
except Exception as e: print("$name FAILED", e)
PY
}
EXTRA="" run plain X=1
EXTRA="--force-collectives" run forced X=1
EXTRA="--force-collectives" run forced_sharedside DPFT_SHARED_WGRAD_STREAM=1
EXTRA="--force-collectives" run forced_sharedview DPFT_SHARED_VIEW_STREAM=1
EXTRA="--force-collectives" run forced_sharedboth DPFT_SHARED_WGRAD_STREAM=1 DPFT_SHARED_VIEW_STREAM=1
EXTRA="" run plain_sharedboth DPFT_SHARED_WGRAD_STREAM=1 DPFT_SHARED_VIEW_STREAM=1
EXTRA="--force-collectives --comm-dtype bf16" run forced_bf16wire X=1
EXTRA="" run plain2 X=1
