# SQ wave-cycle breakdown of the conv kernels on the layer3 problems (one PMC pass, 8 SQ counters)
cd /tmp && export TMPDIR=/tmp
S="fwd:4,32,57,256,256,3,1 dgrad:4,32,57,256,256,3,1 wgrad:4,32,57,256,256,3,1 fwd:4,32,57,256,1024,1,1 dgrad:4,128,228,64,64,3,1 wgrad:4,32,57,1024,256,1,1"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_sq -- env DPFT_COMPUTE=${DPFT_COMPUTE:-fp32} python /root/repo/tools/conv_bench.py $S </dev/null > /tmp/pmc_sq.log 2>&1
tail -8 /tmp/pmc_sq.log
python - <<'PY'
import csv, glob, collections, re
f = glob.glob("/tmp/pmc_sq/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    n = re.sub(r"\(.*", "", r["Kernel_Name"])
    if "dpft" not in n: continue
    key = (n, r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[key] += 1
for key, c in agg.items():
    w = c["SQ_WAVE_CYCLES"] or 1
    print(f"{key[0][:60]:60s} grid {key[1]:>8s} lds {key[2]:>6s} n={cnt[key]:3d} wait_any {c['SQ_WAIT_ANY']/w:.2f} wait_inst {c['SQ_WAIT_INST_ANY']/w:.2f} (lds {c['SQ_WAIT_INST_LDS']/w:.2f}) active {c['SQ_ACTIVE_INST_ANY']/w:.2f} | mfma_busy/wavecyc*4 {c['SQ_VALU_MFMA_BUSY_CYCLES']/(4*w):.3f} lds_conflict/lds_active {c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1):.3f}")
PY
