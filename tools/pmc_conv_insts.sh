# dynamic instruction mix of the conv kernels (one PMC pass)
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_[A-Z_0-9]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_INST_CYCLES_[A-Z_]*" | sort -u | tr '\n' ' '; echo
S="fwd:4,32,57,256,256,3,1 dgrad:4,32,57,256,256,3,1 wgrad:4,32,57,256,256,3,1 fwd:4,32,57,256,1024,1,1 dgrad:4,128,228,64,64,3,1"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmc_in -- python /root/repo/tools/conv_bench.py $S </dev/null > /tmp/pmc_in.log 2>&1
tail -3 /tmp/pmc_in.log | cut -c1-200
python - <<'PY'
import csv, glob, collections, re
f = glob.glob("/tmp/pmc_in/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    n = re.sub(r"\(.*", "", r["Kernel_Name"])
    if "igemm" not in n and "wgrad" not in n: continue
    key = (n, r.get("Grid_Size", ""))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": cnt[key] += 1
for key, c in agg.items():
    m = c["SQ_INSTS_MFMA"] or 1
    print(f"{key[0][:58]:58s} grid {key[1]:>8s} n={cnt[key]:3d} per-MFMA: valu {c['SQ_INSTS_VALU']/m:.2f} salu {c['SQ_INSTS_SALU']/m:.2f} lds {c['SQ_INSTS_LDS']/m:.2f} vmem_rd {c['SQ_INSTS_VMEM_RD']/m:.3f} vmem_wr {c['SQ_INSTS_VMEM_WR']/m:.3f} | wave quad-cycles per MFMA {c['SQ_WAVE_CYCLES']/m:.1f} | mfma/wave {m/max(c['SQ_WAVES'],1):.0f}")
PY
