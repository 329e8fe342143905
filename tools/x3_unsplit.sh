cd /root/repo
export PYTHONUNBUFFERED=1
for i in 1 2; do
DPFT_CONV_TABLE=gpurun_out/table_split_$i.txt python bench.py --steps 30 --warmup 8 --no-cpu-baseline --latency-reps 5 2>/dev/null | grep "^{" | tail -n 1 > gpurun_out/us_split_$i.json
DPFT_X3_UNSPLIT=1 DPFT_CONV_TABLE=gpurun_out/table_unsplit_$i.txt python bench.py --steps 30 --warmup 8 --no-cpu-baseline --latency-reps 5 2>/dev/null | grep "^{" | tail -n 1 > gpurun_out/us_unsplit_$i.json
done
python - <<'PY'
import json
for n in ("split_1","unsplit_1","split_2","unsplit_2"):
    d=json.load(open(f"gpurun_out/us_{n}.json")); print(n, round(d["value"],1), round(d["ms_per_step"],2), d["roofline"]["frac"])
    for l in open(f"gpurun_out/table_{n}.txt"):
        if " 32 57 256 256 3 1 " in l: print("   ", l.strip())
PY
