cd /root/repo
export PYTHONUNBUFFERED=1
for i in 1 2 3; do
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --latency-reps 5 2>/dev/null | grep "^{" | tail -n 1 > gpurun_out/bf_off_$i.json
DPFT_BN_FINAL_FUSE=2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --latency-reps 5 2>/dev/null | grep "^{" | tail -n 1 > gpurun_out/bf_on_$i.json
done
python - <<'PY'
import json
for n in ("off_1","on_1","off_2","on_2","off_3","on_3"):
    d=json.load(open(f"gpurun_out/bf_{n}.json")); print(n, round(d["value"],1), "mean", round(d["ms_per_step"],2), "median", round(d["step_ms_median"],2), "min", round(d["step_ms_min"],2))
PY
