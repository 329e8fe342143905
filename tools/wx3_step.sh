cd /root/repo
export PYTHONUNBUFFERED=1
for i in 1 2; do
DPFT_WGRAD_X3_1X1=0 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --latency-reps 5 2>/dev/null | grep "^{" | tail -n 1 > gpurun_out/wx_off_$i.json
DPFT_WGRAD_X3_1X1=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --latency-reps 5 2>/dev/null | grep "^{" | tail -n 1 > gpurun_out/wx_on_$i.json
done
python - <<'PY'
import json
for n in ("off_1","on_1","off_2","on_2"):
    d=json.load(open(f"gpurun_out/wx_{n}.json")); print(n, round(d["value"],1), "mean", round(d["ms_per_step"],2), "median", round(d["step_ms_median"],2), "frac", round(d["roofline"]["frac"],4), d["roofline"]["per_kind_tflops"])
PY
