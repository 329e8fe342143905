# Round-6 evidence run (one gpurun call).  usage: bash tools/r06_profile.sh  -> gpurun_out/r06_*  (copy into profiles/)
OUT=/root/repo/gpurun_out/r06p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (1) serialized steps: every conv kernel alone on the device (dpft_profile_serialize), no event brackets
SERIAL=1 STEPS=10 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_serial -- python /root/repo/tools/train_only.py </dev/null > $OUT/r06_serial.log 2>&1
f=$(find /tmp/p_serial -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/r06_serialized_step_kernel_stats.csv
# (2) plain steps (concurrent view streams + side-stream weight gradients)
STEPS=20 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_plain -- python /root/repo/tools/train_only.py </dev/null > $OUT/r06_plain.log 2>&1
f=$(find /tmp/p_plain -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/r06_train_step_kernel_stats.csv
# (3) conv family HBM traffic: FETCH_SIZE / WRITE_SIZE in separate passes (2 + 3 warm-up steps each)
for c in FETCH_SIZE WRITE_SIZE; do
  STEPS=2 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python /root/repo/tools/train_only.py </dev/null > /tmp/pmc_$c.log 2>&1
done
# (4) inference decoder: kernel trace, then FETCH_SIZE / WRITE_SIZE in separate passes
REPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dec -- python /root/repo/tools/decoder_only.py </dev/null > $OUT/r06_decoder.log 2>&1
f=$(find /tmp/p_dec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "Name\|decoder_\|pack_" "$f" > $OUT/r06_decoder_kernel_stats.csv
python /root/repo/tools/roofline_from_rocprof.py --decoder $OUT/r06_decoder_kernel_stats.csv > $OUT/r06_decoder_roofline_from_rocprof.json
for c in FETCH_SIZE WRITE_SIZE; do
  REPS=5 timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/p_dec_$c -- python /root/repo/tools/decoder_only.py </dev/null > /tmp/p_dec_$c.log 2>&1
done
python - <<PY
import csv, glob, collections, json
# ---- conv family
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    if f:
        for r in csv.DictReader(open(f[0])):
            if r.get("Counter_Name") != c: continue
            n = r["Kernel_Name"]
            key = "igemm" if ("igemm" in n or "conv16_" in n or "thin_dgrad" in n or "conv1x1_stream" in n) else "wgrad" if ("wgrad" in n) else "splitk" if ("splitk" in n or "slab_reduce" in n) else None
            if key is None: continue
            agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
    raw[c] = {k: {"sum": v[0], "launches": v[1]} for k, v in agg.items()}
try:
    launches = raw["FETCH_SIZE"]["igemm"]["launches"] + raw["FETCH_SIZE"]["wgrad"]["launches"]
    fetch = 2.0 * 1024 * sum(v["sum"] for v in raw["FETCH_SIZE"].values()) / launches
    write = 1024 * sum(v["sum"] for v in raw["WRITE_SIZE"].values()) / launches
    steps = launches / 720.0
    out = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (with --kernel-trace only) over tools/train_only.py (STEPS=2 + 3 warm-up); counter unit KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); WRITE_SIZE uncorrected; kernels: igemm_* / conv16 / thin_dgrad + wgrad_* + their split-K / slab reductions, per conv launch (igemm + wgrad launches)",
           "raw": raw, "steps_profiled": steps, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
           "traffic_bytes_per_launch": fetch + write, "traffic_gb_per_step": (fetch + write) * 720 / 1e9}
except Exception as e:
    out = {"error": repr(e), "raw": raw}
json.dump(out, open("$OUT/r06_conv_traffic_pmc.json", "w"), indent=1)
# ---- decoder
draw, per_fwd = {}, {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/p_dec_{c}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    if f:
        for r in csv.DictReader(open(f[0])):
            if r.get("Counter_Name") != c: continue
            n = r["Kernel_Name"].split("(")[0]
            if "decoder_" not in n: continue
            agg[n][0] += float(r["Counter_Value"]); agg[n][1] += 1
    draw[c] = {k: {"sum_kb": v[0], "launches": v[1], "kb_per_launch": v[0] / max(v[1], 1)} for k, v in agg.items()}
try:
    fwds = sum(v["launches"] for k, v in draw["FETCH_SIZE"].items() if "decoder_xattn" in k) / 4.0
    fetch = 2.0 * 1024 * sum(v["sum_kb"] for v in draw["FETCH_SIZE"].values()) / fwds
    write = 1024 * sum(v["sum_kb"] for v in draw["WRITE_SIZE"].values()) / fwds
    dout = {"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over tools/decoder_only.py (REPS=5 + 3 warm-up forwards of the fused inference decoder at B=4); counter unit KB; FETCH_SIZE doubled (gfx950 note; uncalibrated for the 16-byte gathers, i.e. an upper estimate), WRITE_SIZE uncorrected",
            "raw": draw, "forwards_profiled": fwds, "fetch_bytes_per_forward": fetch, "write_bytes_per_forward": write,
            "traffic_bytes_per_forward": fetch + write, "algorithmic_bytes_per_forward_survey_8d": 555917472}
except Exception as e:
    dout = {"error": repr(e), "raw": draw}
json.dump(dout, open("$OUT/r06_decoder_traffic_pmc.json", "w"), indent=1)
print(json.dumps({k: out.get(k) for k in ("traffic_bytes_per_launch", "traffic_gb_per_step", "error")}))
print(json.dumps({k: dout.get(k) for k in ("traffic_bytes_per_forward", "forwards_profiled", "error")}))
PY
# (5) the driver's own command (CPU leg + secondary legs included) with the per-shape conv table + the family flops of the step
cd /root/repo
DPFT_CONV_TABLE=$OUT/r06_conv_table_fp32.txt timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/r06_bench.json 2> $OUT/r06_bench.err
# (6) rooflines recomputed from the kernel-trace summaries, every conv launch priced against its own pipe (--families)
python tools/roofline_from_rocprof.py $OUT/r06_serialized_step_kernel_stats.csv 13 --families $OUT/r06_conv_table_fp32_families.json > $OUT/r06_roofline_from_rocprof.json
python tools/roofline_from_rocprof.py $OUT/r06_train_step_kernel_stats.csv 23 --families $OUT/r06_conv_table_fp32_families.json > $OUT/r06_roofline_from_rocprof_plain_steps.json
# (7) the RCCL code path on one rank (forced collectives): the line carries dp_exchange = per-bucket ready / start / end of the last timed step
timeout 900 python bench.py --gpus 1 --steps 40 --warmup 10 --force-collectives --no-cpu-baseline --latency-reps 20 > $OUT/r06_bench_forced_collectives.json 2> $OUT/r06_bench_forced.err
timeout 900 python bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --latency-reps 20 > $OUT/r06_bench_plain_40steps.json 2>> $OUT/r06_bench_forced.err
# (8) SQ wave-cycle / MFMA-busy breakdown of the conv kernels on the layer-3 problems (one PMC pass)
bash /root/repo/tools/pmc_conv_sq.sh > $OUT/r06_conv_sq_mfma_busy.txt 2>&1
tail -2 $OUT/r06_serial.log; grep "ms/step" $OUT/r06_plain.log; head -c 400 $OUT/r06_roofline_from_rocprof.json; tail -c 300 $OUT/r06_bench.json
