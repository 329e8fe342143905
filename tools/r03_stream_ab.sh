#!/bin/bash
# Stream placement A/B: probed hardware queues (default) vs pool streams, plain and with forced RCCL collectives.
set -u
OUT=gpurun_out/r03_stream_ab; mkdir -p $OUT
run() { name=$1; shift; env "$@" python bench.py --gpus 1 --steps 30 --warmup 8 --latency-reps 20 --no-cpu-baseline $EXTRA > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    l=json.loads([x for x in open("$OUT/$name.json") if x.startswith("{")][-1])
    print("$name", round(l["ms_per_step"],2), "ms", round(l["value"],1), "samples/s exposed", round(l["exposed_allreduce_ms"],3), "fwd/frame", round(l["fwd_ms_per_frame"],3))
except Exception as e: print("$name FAILED", e); print(open("$OUT/$name.err").read()[-600:])
PY
}
for rep in 1 2; do
EXTRA="" run probe_$rep DPFT_STREAM_PLACEMENT=probe
EXTRA="" run pool_$rep DPFT_STREAM_PLACEMENT=pool
EXTRA="--force-collectives" run probe_forced_$rep DPFT_STREAM_PLACEMENT=probe
EXTRA="--force-collectives" run pool_forced_$rep DPFT_STREAM_PLACEMENT=pool
done
