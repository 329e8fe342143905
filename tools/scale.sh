#!/bin/bash
# Turn-key 1 / 2 / 4 / 8-GPU scaling matrix for the day an 8-GPU MI355X node is available (no such node was
# reachable in rounds 1-4: everything below is untested on N > 1 hardware, the N > 1 CODE PATH is exercised on one rank by
# `bench.py --force-collectives` and by the world_size-2 gloo tests).
#
#   bash tools/scale.sh [out_dir]          -> <out_dir>/scale_<N>_<bucket>_<wire>_<stream>.json + scale_table.txt
#
# Axes: ranks N in {1,2,4,8} x gradient bucket {25,100} MiB x wire format {fp32,bf16} x the collectives' stream
# DPFT_COMM_STREAM in {side (the camera's weight-gradient stream), pg (the process group's own), own (a dedicated stream)}.
# DESIGN.md section 6 holds the pre-registered prediction the first run tests.
OUT=${1:-/root/repo/gpurun_out/scale}; mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
STEPS=${STEPS:-100}; WARM=${WARM:-20}
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  [ "$N" -gt "$NGPU" ] && { echo "only $NGPU GPUs visible: skipping N=$N"; continue; }
  for BUCKET in 25 100; do for WIRE in fp32 bf16; do for COMM in side pg own; do
    [ "$N" = 1 ] && [ "$BUCKET$WIRE$COMM" != "25fp32side" ] && continue       # one rank: no exchange, one line
    f="$OUT/scale_${N}_${BUCKET}_${WIRE}_${COMM}.json"
    if [ "$N" = 1 ]; then
      python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-cpu-baseline --latency-reps 20 > "$f" 2> "$f.err"
    else
      DPFT_COMM_STREAM=$COMM python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
        --master-port $((29500 + N)) bench.py --gpus $N --steps $STEPS --warmup $WARM --bucket-mb $BUCKET --comm-dtype $WIRE \
        --no-cpu-baseline --latency-reps 20 > "$f" 2> "$f.err"
    fi
  done; done; done
done
python - "$OUT" <<'PY'
import glob, json, os, sys
rows = []
for f in sorted(glob.glob(os.path.join(sys.argv[1], "scale_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    n, bucket, wire, comm = os.path.basename(f)[6:-5].split("_")
    rows.append((int(n), int(bucket), wire, comm, d["value"], d["ms_per_step"], d.get("exposed_allreduce_ms"), d.get("step_ms_median")))
base = next((r[4] for r in rows if r[0] == 1), None)
with open(os.path.join(sys.argv[1], "scale_table.txt"), "w") as out:
    hdr = f"{'N':>2} {'bucket':>6} {'wire':>5} {'comm':>5} {'samples/s':>10} {'ms/step':>8} {'median':>8} {'exposed ms':>10} {'x vs N=1':>9} {'efficiency':>10}"
    print(hdr); out.write(hdr + "\n")
    for r in sorted(rows):
        sp = r[4] / base if base else float("nan")
        ln = f"{r[0]:>2} {r[1]:>6} {r[2]:>5} {r[3]:>5} {r[4]:>10.1f} {r[5]:>8.2f} {(r[7] or float('nan')):>8.2f} {(r[6] if r[6] is not None else float('nan')):>10.2f} {sp:>9.2f} {sp / r[0]:>10.3f}"
        print(ln); out.write(ln + "\n")
PY
