cd /root/repo
export PYTHONUNBUFFERED=1
for i in 1 2; do
python bench.py --steps 10 --warmup 5 --no-cpu-baseline --latency-reps 100 2>/dev/null | tail -n 1 > gpurun_out/lat_split_$i.json
python bench.py --steps 10 --warmup 5 --no-cpu-baseline --latency-reps 100 --no-split 2>/dev/null | tail -n 1 > gpurun_out/lat_nosplit_$i.json
done
python - <<'PY'
import json
for n in ("split_1","nosplit_1","split_2","nosplit_2"):
    d=json.load(open(f"gpurun_out/lat_{n}.json")); print(n, round(d["value"],1), round(d["ms_per_step"],2), "fwd/frame", round(d["fwd_ms_per_frame"],3), "batch1", round(d["fwd_ms_batch1"],3), d["roofline"]["frac"])
PY
