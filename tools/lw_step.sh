cd /root/repo
export PYTHONUNBUFFERED=1
for i in 1 2; do
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --latency-reps 5 2>/dev/null | grep "^{" | tail -n 1 > gpurun_out/lw_new_$i.json
DPFT_LSAP_C=0 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --latency-reps 5 2>/dev/null | grep "^{" | tail -n 1 > gpurun_out/lw_scipy_$i.json
done
python - <<'PY'
import json
for n in ("new_1","scipy_1","new_2","scipy_2"):
    d=json.load(open(f"gpurun_out/lw_{n}.json")); print(n, round(d["value"],1), "mean", round(d["ms_per_step"],2), "median", round(d["step_ms_median"],2), "window", d.get("loss_window_us"))
PY
python -m pytest tests/test_gpu_trainer.py -m gpu -q 2>&1 | tail -3
