# Idle time on the busiest stream of a training step, by (previous kernel -> next kernel) category.  bash tools/chain_gaps.sh
cd /tmp && export TMPDIR=/tmp
STEPS=3 timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/proft4 -- python /root/repo/tools/train_only.py </dev/null > /tmp/proft4.log 2>&1
f=$(find /tmp/proft4 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
step = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(step[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in step)
print("step span ms", (t1 - t0) / 1e6)
key = "Stream_Id" if "Stream_Id" in step[0] else "Queue_Id"
def cat(n):
    n = n.split("(")[0]
    for pat, c in (("igemm", "conv"), ("conv16", "conv"), ("thin_dgrad", "conv"), ("wgrad", "wgrad"), ("splitk", "splitk"), ("slab", "splitk"),
                   ("bn_bwd_reduce", "bn_bwd_reduce"), ("bn_bwd_apply", "bn_bwd_apply"), ("bn_finalize", "bn_finalize"), ("bn_act", "bn_act"),
                   ("bn_", "bn_other"), ("transpose", "transpose"), ("add_inplace", "add"), ("sa_train", "decoder"), ("xf_train", "decoder"),
                   ("hd_train", "decoder"), ("rows_outer", "decoder"), ("at::native", "aten"), ("rocclr", "rocclr")):
        if pat in n: return c
    return "other"
per = collections.defaultdict(list)
for r in step: per[r[key]].append(r)
main = max(per.values(), key=lambda v: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in v))
main.sort(key=lambda r: int(r["Start_Timestamp"]))
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in main)
print(f"main stream: {len(main)} kernels, busy {busy/1e6:.2f} ms, window {(int(main[0]['Start_Timestamp'])-t0)/1e6:.2f}..{(int(main[-1]['End_Timestamp'])-t0)/1e6:.2f}")
gaps = collections.defaultdict(lambda: [0, 0]); kt = collections.defaultdict(lambda: [0, 0])
for a, b in zip(main, main[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    k = (cat(a["Kernel_Name"]), cat(b["Kernel_Name"]))
    gaps[k][0] += max(g, 0); gaps[k][1] += 1
for r in main:
    c = cat(r["Kernel_Name"]); kt[c][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); kt[c][1] += 1
print("kernel time on the main stream by category:")
for c, (t, n) in sorted(kt.items(), key=lambda kv: -kv[1][0]): print(f"  {c:16s} {t/1e6:7.2f} ms {n:5d} launches  avg {t/n/1e3:6.1f} us")
tot = sum(v[0] for v in gaps.values())
print(f"idle between consecutive kernels: {tot/1e6:.2f} ms")
for k, (t, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]: print(f"  {k[0]:14s} -> {k[1]:14s} {t/1e6:6.2f} ms {n:4d} x avg {t/n/1e3:5.1f} us")
PY
