#!/bin/bash
# DESIGN section 6 cost model on ONE GPU: the forced one-rank RCCL step with a stand-in kernel behind every
# bucket collective (32 workgroups, reads + writes the bucket twice: the CU + HBM footprint of an 8-rank ring all-reduce),
# for each placement of the communication stream.  Two alternating rounds.
cd /root/repo
OUT=gpurun_out/r05_comm_standin; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/probes/bin/libcomm_standin.so tools/probes/comm_standin.hip
run() { name=$1; shift; env "$@" python $PYW $PYWARGS --gpus 1 --steps 40 --warmup 10 --latency-reps 5 --no-cpu-baseline $EXTRA 2>$OUT/$name.err | grep "^{" | tail -n 1 > $OUT/$name.json; python - <<PY
import json
try:
    l=json.load(open("$OUT/$name.json")); print("$name", round(l["ms_per_step"],2), "ms", round(l["value"],1), "samples/s exposed", l["exposed_allreduce_ms"])
except Exception as e: print("$name FAILED", e); print(open("$OUT/$name.err").read()[-600:])
PY
}
for rep in 1 2; do
PYW=bench.py PYWARGS="" EXTRA="" run plain_$rep X=1
for place in side own pg; do
PYW=bench.py PYWARGS="" EXTRA="--force-collectives" run forced_${place}_$rep DPFT_COMM_STREAM=$place
PYW=tools/exp_switches.py PYWARGS="--standin-collective --" EXTRA="--force-collectives" run standin_${place}_$rep DPFT_COMM_STREAM=$place
done
done | tee $OUT/summary.txt
