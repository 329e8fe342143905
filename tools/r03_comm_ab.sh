#!/bin/bash
set -u
OUT=gpurun_out/r03_comm_ab; mkdir -p $OUT
run() { name=$1; shift; env "$@" python bench.py --gpus 1 --steps 30 --warmup 8 --latency-reps 20 --no-cpu-baseline $EXTRA > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    l=json.loads([x for x in open("$OUT/$name.json") if x.startswith("{")][-1])
    print("$name", round(l["ms_per_step"],2), "ms", round(l["value"],1), "samples/s exposed", round(l["exposed_allreduce_ms"],3))
except Exception as e: print("$name FAILED", e); print(open("$OUT/$name.err").read()[-800:])
PY
}
for rep in 1 2; do
EXTRA="" run plain_$rep X=1
EXTRA="--force-collectives" run forced_front_$rep DPFT_COMM_STREAM=front
EXTRA="--force-collectives" run forced_side_$rep DPFT_COMM_STREAM=side
EXTRA="--force-collectives" run forced_pg_$rep DPFT_COMM_STREAM=pg
done
EXTRA="--force-collectives --comm-dtype bf16" run forced_front_bf16 DPFT_COMM_STREAM=front
