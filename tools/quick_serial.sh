# Quick iteration check (one gpurun call): serialized-step rocprof summary -> conv-family roofline, + a 30-step bench line.
# usage: bash tools/quick_serial.sh [tag]   -> gpurun_out/q_<tag>/
TAG=${1:-x}
OUT=/root/repo/gpurun_out/q_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SERIAL=1 STEPS=10 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q_serial_$TAG -- python /root/repo/tools/train_only.py </dev/null > $OUT/serial.log 2>&1
f=$(find /tmp/q_serial_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/serialized_step_kernel_stats.csv
python /root/repo/tools/roofline_from_rocprof.py $OUT/serialized_step_kernel_stats.csv 13 > $OUT/roofline_from_rocprof.json
cd /root/repo
timeout 600 python bench.py --steps ${STEPS_BENCH:-40} --warmup 10 --no-cpu-baseline --latency-reps 30 > $OUT/bench.json 2> $OUT/bench.err
python - $OUT <<'PY'
import json, sys
o = sys.argv[1]
r = json.load(open(o + "/roofline_from_rocprof.json"))
print({k: r[k] for k in ("frac", "frac_main_only", "total_kernel_ms_per_step")}, r["kernel_ms_per_step"], r["launches_per_step"])
try:
    b = json.loads(open(o + "/bench.json").read().strip().splitlines()[-1])
    print("bench", round(b["value"], 2), "samples/s", round(b["ms_per_step"], 3), "ms  median", round(b["step_ms_median"], 3), " fwd/frame", round(b["fwd_ms_per_frame"], 3),
          " conv frac", round(b["roofline"]["frac"], 4), " dec", round(b["roofline_decoder"]["decoder_fwd_us"], 1), "us  dec_train", round(b["roofline_decoder_train"]["decoder_train_fwd_bwd_us"], 1), "us")
except Exception as e:
    print("bench line unreadable:", e)
PY
