#!/bin/bash
# step time with the pipelined kernels on / off (same box, alternating)
mkdir -p gpurun_out/r03
for rep in 1 2; do for p in 3 1 0 2; do
DPFT_PIPE=$p python bench.py --steps 30 --warmup 8 --no-cpu-baseline --latency-reps 20 > gpurun_out/r03/ab_pipe${p}_$rep.json 2>/dev/null
python - <<PY
import json
l=json.loads([x for x in open("gpurun_out/r03/ab_pipe${p}_$rep.json") if x.startswith("{")][-1]); r=l["roofline"]
print("DPFT_PIPE=$p rep $rep:", round(l["ms_per_step"],2), "ms", round(l["value"],1), "samples/s frac", round(r["frac"],4), {k: round(v,1) for k,v in r["per_kind_tflops"].items()})
PY
done; done
