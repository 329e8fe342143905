# Per-shape effect of the igemm tile on the short-K / wide-N camera convs (epilogue-bound: all workgroups of a one-wave grid
# reach their epilogue together).  One bench run per forced tile, rows of the per-shape table side by side.
mkdir -p /root/repo/gpurun_out/tile_ab; cd /root/repo
for t in default 128,128,1 128,64,1 64,64,1; do
  f=gpurun_out/tile_ab/table_${t//,/_}.txt
  if [ $t = default ]; then DPFT_CONV_TABLE=$f python bench.py --steps 6 --warmup 3 --no-cpu-baseline --latency-reps 5 > /dev/null 2>&1
  else DPFT_FORCE_TILE=$t DPFT_CONV_TABLE=$f python bench.py --steps 6 --warmup 3 --no-cpu-baseline --latency-reps 5 > /dev/null 2>&1; fi
done
python - <<'PY'
import glob
tabs = {}
for f in sorted(glob.glob("gpurun_out/tile_ab/table_*.txt")):
    name = f.split("table_")[1][:-4]
    for l in open(f):
        p = l.split()
        if len(p) < 11 or p[0] not in ("fwd", "dgrad"): continue
        key = " ".join(p[:8])
        tabs.setdefault(key, {})[name] = (float(p[9]) / int(p[8]), float(p[10]))
cam_w = {"910", "228", "114", "57", "29"}
rows = [(v.get("default", (0, 0))[0] * 1, k, v) for k, v in tabs.items() if k.split()[3] in cam_w]
rows.sort(reverse=True)
names = ["default", "128_128_1", "128_64_1", "64_64_1"]
print(f"{'shape (kind B H W C K k s)':36s}" + "".join(f"{n:>20s}" for n in names))
for _, k, v in rows[:45]:
    print(f"{k:36s}" + "".join(f"{v[n][0]:10.1f}us {v[n][1]:5.1f}TF" if n in v else " " * 20 for n in names))
PY
