#!/bin/bash
# where the forced one-rank RCCL step loses its ~1 ms: the step-decision collective (+ host sync) vs the bucket collectives
set -u
OUT=gpurun_out/r03_forced_breakdown; mkdir -p $OUT
run() { name=$1; shift; env "$@" python ${PYW:-bench.py} ${PYWARGS:-} --gpus 1 --steps 30 --warmup 8 --latency-reps 10 --no-cpu-baseline $EXTRA > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    l=json.loads([x for x in open("$OUT/$name.json") if x.startswith("{")][-1])
    print("$name", round(l["ms_per_step"],2), "ms", round(l["value"],1), "samples/s exposed", round(l["exposed_allreduce_ms"],3))
except Exception as e: print("$name FAILED", e); print(open("$OUT/$name.err").read()[-800:])
PY
}
for rep in 1 2; do
EXTRA="" run plain_$rep X=1
EXTRA="--force-collectives" run forced_$rep X=1
PYW=tools/exp_switches.py PYWARGS="--local-decision --" EXTRA="--force-collectives" run forced_local_decision_$rep X=1
PYW=tools/exp_switches.py PYWARGS="--skip-bucket-collectives --" EXTRA="--force-collectives" run forced_no_bucket_collectives_$rep X=1
PYW=tools/exp_switches.py PYWARGS="--skip-bucket-collectives --local-decision --" EXTRA="--force-collectives" run forced_neither_$rep X=1
EXTRA="--force-collectives" run forced_avg_op_$rep DPFT_COLLECTIVE_OP=avg
done
