# Round-5 evidence run (one gpurun call).  usage: bash tools/r05_profile.sh  -> gpurun_out/r05_*  (copy into profiles/)
OUT=/root/repo/gpurun_out/r05p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (1) serialized steps: every conv kernel alone on the device (dpft_profile_serialize), no event brackets
SERIAL=1 STEPS=10 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_serial -- python /root/repo/tools/train_only.py </dev/null > $OUT/r05_serial.log 2>&1
f=$(find /tmp/p_serial -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/r05_serialized_step_kernel_stats.csv
# (2) plain steps (concurrent view streams + side-stream weight gradients)
STEPS=20 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_plain -- python /root/repo/tools/train_only.py </dev/null > $OUT/r05_plain.log 2>&1
f=$(find /tmp/p_plain -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/r05_train_step_kernel_stats.csv
python /root/repo/tools/roofline_from_rocprof.py $OUT/r05_serialized_step_kernel_stats.csv 13 > $OUT/r05_roofline_from_rocprof.json
python /root/repo/tools/roofline_from_rocprof.py $OUT/r05_train_step_kernel_stats.csv 23 > $OUT/r05_roofline_from_rocprof_plain_steps.json
# (3) conv family HBM traffic: FETCH_SIZE / WRITE_SIZE in separate passes (2 + 3 warm-up steps each)
for c in FETCH_SIZE WRITE_SIZE; do
  STEPS=2 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python /root/repo/tools/train_only.py </dev/null > /tmp/pmc_$c.log 2>&1
done
# (4) inference decoder: kernel trace, then FETCH_SIZE / WRITE_SIZE in separate passes
REPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dec -- python /root/repo/tools/decoder_only.py </dev/null > $OUT/r05_decoder.log 2>&1
f=$(find /tmp/p_dec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "Name\|decoder_\|pack_" "$f" > $OUT/r05_decoder_kernel_stats.csv
python /root/repo/tools/roofline_from_rocprof.py --decoder $OUT/r05_decoder_kernel_stats.csv > $OUT/r05_decoder_roofline_from_rocprof.json
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
            key = "igemm" if ("igemm" in n or "conv16_" in n or "thin_dgrad" in n) else "wgrad" if ("wgrad" in n) else "splitk" if ("splitk" in n or "slab_reduce" in n) else None
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
json.dump(out, open("$OUT/r05_conv_traffic_pmc.json", "w"), indent=1)
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
json.dump(dout, open("$OUT/r05_decoder_traffic_pmc.json", "w"), indent=1)
print(json.dumps({k: out.get(k) for k in ("traffic_bytes_per_launch", "traffic_gb_per_step", "error")}))
print(json.dumps({k: dout.get(k) for k in ("traffic_bytes_per_forward", "forwards_profiled", "error")}))
PY
# (5) SQ wave-cycle / MFMA-busy breakdown of the conv kernels on the layer-3 problems (one PMC pass)
bash /root/repo/tools/pmc_conv_sq.sh > $OUT/r05_conv_sq_mfma_busy.txt 2>&1
# (6) mixed precision (configs[4]): bf16 operands + bf16 activation storage, batch 8 per GPU: bench line + serialized kernel summary
DPFT_CONV_TABLE=$OUT/r05_conv_table_bf16_b8.txt timeout 600 python /root/repo/bench.py --dtype bf16 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --latency-reps 50 > $OUT/r05_bench_bf16_b8.json 2> $OUT/r05_bench_bf16_b8.err
timeout 600 python /root/repo/bench.py --dtype bf16 --batch 4 --steps 20 --warmup 5 --no-cpu-baseline --latency-reps 50 > $OUT/r05_bench_bf16_b4.json 2>> $OUT/r05_bench_bf16_b8.err
for b in 8 4; do
  SERIAL=1 STEPS=10 DTYPE=bf16 BATCH=$b timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_serial16_$b -- python /root/repo/tools/train_only.py </dev/null > $OUT/r05_serial_bf16_b$b.log 2>&1
  f=$(find /tmp/p_serial16_$b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/r05_bf16_b${b}_serialized_step_kernel_stats.csv
  python /root/repo/tools/roofline_from_rocprof.py $OUT/r05_bf16_b${b}_serialized_step_kernel_stats.csv 13 $(python -c "print(1839.439164384 * $b / 4)") 2500 > $OUT/r05_bf16_b${b}_roofline_from_rocprof.json
done
# conv family HBM traffic of the bf16 step at batch 8 (FETCH_SIZE / WRITE_SIZE, separate passes)
for c in FETCH_SIZE WRITE_SIZE; do
  STEPS=2 DTYPE=bf16 BATCH=8 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc16_$c -- python /root/repo/tools/train_only.py </dev/null > /tmp/pmc16_$c.log 2>&1
done
python - <<PY
import csv, glob, collections, json
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc16_{c}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    if f:
        for r in csv.DictReader(open(f[0])):
            if r.get("Counter_Name") != c: continue
            n = r["Kernel_Name"]
            key = "igemm" if ("igemm" in n or "conv16_" in n or "thin_dgrad" in n) else "wgrad" if ("wgrad" in n) else "splitk" if ("splitk" in n or "slab_reduce" in n) else "bn" if "bn_" in n else None
            if key is None: continue
            agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
    raw[c] = {k: {"sum_kb": v[0], "launches": v[1]} for k, v in agg.items()}
try:
    conv = ("igemm", "wgrad", "splitk")
    launches = raw["FETCH_SIZE"]["igemm"]["launches"] + raw["FETCH_SIZE"]["wgrad"]["launches"]
    steps = 5.0      # STEPS=2 + 3 warm-up steps of tools/train_only.py
    fetch = 2.0 * 1024 * sum(raw["FETCH_SIZE"][k]["sum_kb"] for k in conv if k in raw["FETCH_SIZE"])
    write = 1024 * sum(raw["WRITE_SIZE"][k]["sum_kb"] for k in conv if k in raw["WRITE_SIZE"])
    out = {"method": "as r05_conv_traffic_pmc.json (FETCH_SIZE doubled, WRITE_SIZE as is, KB units), bf16 mode (configs[4]) at batch 8; bn = the BatchNorm kernels of the same passes",
           "raw": raw, "steps_profiled": steps, "conv_launches_per_step": launches / steps, "conv_fetch_gb_per_step": fetch / steps / 1e9, "conv_write_gb_per_step": write / steps / 1e9,
           "conv_traffic_gb_per_step": (fetch + write) / steps / 1e9,
           "bn_traffic_gb_per_step": (2.0 * 1024 * raw["FETCH_SIZE"].get("bn", {"sum_kb": 0})["sum_kb"] + 1024 * raw["WRITE_SIZE"].get("bn", {"sum_kb": 0})["sum_kb"]) / steps / 1e9}
except Exception as e:
    out = {"error": repr(e), "raw": raw}
json.dump(out, open("$OUT/r05_bf16_b8_conv_traffic_pmc.json", "w"), indent=1)
print({k: v for k, v in out.items() if k not in ("raw", "method")})
PY
# (7) radar tesseract projection
bash /root/repo/tools/radar_prof.sh > $OUT/r05_radar_projection.txt 2>&1
# (7b) the RCCL path on one rank: forced collectives (line carries rccl_ranks 1, collectives_forced true)
# (100 timed steps each, plain and forced back to back in the same call)
timeout 900 python /root/repo/bench.py --gpus 1 --steps 100 --warmup 20 --no-cpu-baseline --latency-reps 20 > $OUT/r05_bench_plain_100steps.json 2> $OUT/r05_bench_forced.err
timeout 900 python /root/repo/bench.py --gpus 1 --steps 100 --warmup 20 --force-collectives --no-cpu-baseline --latency-reps 20 > $OUT/r05_bench_forced_collectives.json 2>> $OUT/r05_bench_forced.err
# (8) the default bench line of this state: the command the driver runs (python bench.py = 100 timed + 20 warm-up steps)
cd /root/repo
# (the driver's own command: python bench.py --steps 20 --warmup 5, CPU leg included)
DPFT_CONV_TABLE=$OUT/r05_conv_table_fp32.txt timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r05_bench.json 2> $OUT/r05_bench.err
# (8b) the same without the split kernels (fp32 MFMA everywhere): the round's A/B on one box
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split > $OUT/r05_bench_no_split.json 2> $OUT/r05_bench_no_split.err
# (9) loader-in-the-loop rate, frozen-BN / train-mode gradient probe
timeout 600 python tools/loader_rate.py > $OUT/r05_loader_rate.json 2> $OUT/r05_loader_rate.err
timeout 900 python tools/grad_gap_probe.py > $OUT/r05_grad_gap_probe_rerun.txt 2> /dev/null      # (profiles/r05_grad_gap_probe.txt also holds the four-other-seeds runs)
tail -2 $OUT/r05_serial.log; grep "ms/step" $OUT/r05_plain.log; grep decoder_fwd $OUT/r05_decoder.log; head -c 600 $OUT/r05_roofline_from_rocprof.json
# (10) vendor / ATen kernels inside one steady-state step (the whole-run statistics above also count the set-up copies and fills)
bash /root/repo/tools/step_vendor_rows.sh > $OUT/r05_step_vendor_rows.txt 2>&1
# (11) the loader rate again with the sample files read from disk (KRadarFolderDataset over a generated folder tree)
FILES=1 timeout 600 python /root/repo/tools/loader_rate.py > /dev/null 2>&1      # (first run on a fresh box: the tree is written, the workers' imports are cold -- 134 vs 152 samples/s)
FILES=1 timeout 600 python /root/repo/tools/loader_rate.py > $OUT/r05_loader_rate_files.json 2>> $OUT/r05_loader_rate.err
# (12) the driver's own command under the kernel trace: python bench.py --steps 20 --warmup 5 (CPU leg off: it is host work)
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_benchcmd -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline </dev/null > $OUT/r05_bench_cmd_under_rocprof.json 2> $OUT/r05_bench_cmd.err
f=$(find /tmp/p_benchcmd -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/r05_bench_cmd_kernel_stats.csv
cd /root/repo
