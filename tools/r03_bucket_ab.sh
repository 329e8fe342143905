#!/bin/bash
# forced one-rank RCCL collectives: step time vs gradient bucket size (fewer, larger all-reduces), against the plain step
set -u
OUT=gpurun_out/r03_bucket_ab; mkdir -p $OUT
run() { name=$1; shift; python bench.py --gpus 1 --steps 30 --warmup 8 --latency-reps 10 --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    l=json.loads([x for x in open("$OUT/$name.json") if x.startswith("{")][-1])
    print("$name", round(l["ms_per_step"],2), "ms", round(l["value"],1), "samples/s exposed", round(l["exposed_allreduce_ms"],3))
except Exception as e: print("$name FAILED", e); print(open("$OUT/$name.err").read()[-800:])
PY
}
for rep in 1 2; do
run plain_$rep
for mb in 25 100; do run forced_${mb}mb_$rep --force-collectives --bucket-mb $mb; done
done
