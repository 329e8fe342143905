"""Launches per training step by kernel name, from a rocprofv3 --kernel-trace csv (steps split at the optimizer kernel).
   python tools/step_kernel_counts.py kernel_trace.csv [pattern]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
marks = [s for s, e, n, q in ev if "adamw_kernel" in n]
t0, t1 = marks[-4], marks[-1]
cnt = collections.Counter(); dur = collections.Counter(); prev = collections.Counter(); nxt = collections.Counter()
win = [x for x in ev if t0 <= x[0] < t1]
for i, (s, e, n, q) in enumerate(win):
    k = n.split("(")[0][:80]
    cnt[k] += 1; dur[k] += e - s
    if pat and pat in n:
        prev[win[i - 1][2].split("(")[0][:60] + " | q" + str(q)] += 1
        if i + 1 < len(win): nxt[win[i + 1][2].split("(")[0][:60]] += 1
print("per step (3 steps averaged):")
for k, v in sorted(cnt.items(), key=lambda kv: -dur[kv[0]])[:60]:
    if not pat or pat in k:
        print(f"  {v / 3:7.1f} x  {dur[k] / v / 1e3:7.1f} us  {k}")
if pat:
    print("launched before it:"); [print(f"  {v / 3:6.1f}  {k}") for k, v in prev.most_common(25)]
    print("launched after it:"); [print(f"  {v / 3:6.1f}  {k}") for k, v in nxt.most_common(25)]
