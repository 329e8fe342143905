# Per-shape sweep of the implicit-GEMM tile and split count over the whole training step (round 4: with the in-kernel split-K
# fix-up, split launches of LARGER tiles are candidates where a grid of 64 x 64 tiles was the only way to fill the chip).
# One short bench run per forced configuration; the serialized step's per-shape table of each run side by side.
# usage: bash tools/tile_sweep.sh [igemm|wgrad]   -> gpurun_out/tile_sweep/{igemm,wgrad}_best.txt
mkdir -p /root/repo/gpurun_out/tile_sweep; cd /root/repo
what=${1:-igemm}
B="python bench.py --steps 2 --warmup 3 --no-cpu-baseline --latency-reps 3 $BENCH_ARGS"      # e.g. BENCH_ARGS="--dtype bf16 --batch 8"
DPFT_CONV_TABLE=gpurun_out/tile_sweep/${what}_default.txt timeout 200 $B > /dev/null 2>&1
if [ $what = igemm ]; then
  for t in 128,128 128,64 64,64; do for s in ${SPLITS:-1 2 3 4 6 8}; do
    DPFT_FORCE_TILE=$t,$s DPFT_CONV_TABLE=gpurun_out/tile_sweep/igemm_${t/,/x}_$s.txt timeout 200 $B > /dev/null 2>&1 || echo "run $t,$s failed"
  done; done
else
  for t in 128 64; do for s in 1 2 4 7 8 14 16 28 32 64; do
    DPFT_FORCE_WGRAD=$t,$s DPFT_CONV_TABLE=gpurun_out/tile_sweep/wgrad_${t}_$s.txt timeout 200 $B > /dev/null 2>&1 || echo "run $t,$s failed"
  done; done
fi
python - $what <<'PY'
import glob, sys
what = sys.argv[1]
kinds = ("fwd", "dgrad") if what == "igemm" else ("wgrad",)
tabs = {}
for f in sorted(glob.glob(f"gpurun_out/tile_sweep/{what}_*.txt")):
    name = f.split(f"{what}_")[1][:-4]
    if name == "best": continue
    for l in open(f):
        p = l.split()
        if len(p) < 11 or p[0] not in kinds: continue
        tabs.setdefault(" ".join(p[:8]), {})[name] = (float(p[9]) / int(p[8]), int(p[8]))
rows = sorted(((v["default"][0] * v["default"][1], k, v) for k, v in tabs.items() if "default" in v), reverse=True)
gain = 0.0
with open(f"gpurun_out/tile_sweep/{what}_best.txt", "w") as out:
    for tot, k, v in rows:
        best = min((t[0], n) for n, t in v.items())
        d = v["default"][0]
        g = (d - best[0]) * v["default"][1]
        same = [n for n, t in v.items() if n != "default" and abs(t[0] - d) < 0.02 * d]
        line = f"{k:34s} calls {v['default'][1]:3d} default {d:8.1f} us  best {best[1]:12s} {best[0]:8.1f} us  gain {g:7.1f} us   default~ {','.join(same[:3])}"
        if g > 0.03 * tot: gain += g
        out.write(line + "\n")
    out.write(f"sum of gains > 3 %: {gain:.1f} us per step\n")
print(open(f"gpurun_out/tile_sweep/{what}_best.txt").read()[:6000])
PY
