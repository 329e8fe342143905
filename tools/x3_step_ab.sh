# whole-step A/B: fp32 MFMA vs the 3 x bf16 split mode (same box, alternating)
cd /root/repo
export PYTHONUNBUFFERED=1
for i in 1 2; do
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --latency-reps 30 2>/dev/null | tail -n 1 > gpurun_out/step_f32_$i.json
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --latency-reps 30 --dtype f32x3 2>/dev/null | tail -n 1 > gpurun_out/step_x3_$i.json
done
python - <<'PY'
import json
for n in ("f32_1","x3_1","f32_2","x3_2"):
    try:
        d=json.load(open(f"gpurun_out/step_{n}.json"))
        print(n, d["value"], d["ms_per_step"], d.get("fwd_ms_per_frame"), d["roofline"].get("frac"))
    except Exception as e: print(n, "ERR", e)
PY
