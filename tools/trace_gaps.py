"""Where the GPU idles inside a training step: from a rocprofv3 --kernel-trace csv, the union of the kernel intervals per step
window, and the idle gaps grouped by the kernel that ENDS before the gap.   python tools/trace_gaps.py kernel_trace.csv [steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# the last `steps` steps: split at the optimizer kernel
marks = [s for s, e, n in ev if "adamw_kernel" in n]
marks = marks[-(steps + 1):]
t0, t1 = marks[0], marks[-1]
win = [x for x in ev if t0 <= x[0] < t1]
busy = 0; gaps = collections.Counter(); gapn = collections.Counter(); cur_end = win[0][0]; last = win[0][2]
conc = 0
big = []
for s, e, n in win:
    if s > cur_end:
        g = s - cur_end
        key = last.split("(")[0][:70] + "  ->  " + n.split("(")[0][:60]
        gaps[key] += g; gapn[key] += 1
        if g > 20000: big.append((g, key))
        busy_start = s
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e; last = n
nst = len(marks) - 1
print(f"steps {nst}  step {(t1 - t0) / nst / 1e6:.3f} ms  busy (union) {busy / nst / 1e6:.3f} ms  idle {(t1 - t0 - busy) / nst / 1e6:.3f} ms  kernels/step {len(win) / nst:.0f}")
tot = sum(gaps.values())
print("idle by (kernel that ended -> kernel that started), per step:")
for k, v in gaps.most_common(40):
    print(f"  {v / nst / 1e3:8.1f} us  x{gapn[k] / nst:6.1f}  avg {v / gapn[k] / 1e3:6.1f} us   {k}")
