"""GPU idle time inside steady-state training steps from a rocprofv3 kernel trace: the union of all queues' busy intervals
between two consecutive assignment kernels (one per step), and the largest gaps with the kernels on either side.
usage: python tools/step_gaps.py <kernel_trace.csv>
Caveat (round 5): under the kernel trace the step takes 38 ms instead of 24 -- the tracer makes it launch-bound, so the gaps it
shows (8 ms per step) are mostly the tracer's; usable for WHERE launches bunch up, not for how long the GPU idles untraced."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]) for r in rows))
marks = [s for s, e, n in ev if "lsap_batch_kernel" in n]
for k in range(len(marks) - 4, len(marks) - 1):
    t0, t1 = marks[k], marks[k + 1]
    seg = [(s, e, n) for s, e, n in ev if e > t0 and s < t1]
    busy, gaps, cur_e, last = 0, [], t0, "step start"
    for s, e, n in seg:
        s = max(s, t0); e = min(e, t1)
        if s > cur_e:
            gaps.append((s - cur_e, last, n))
            busy += e - s
            cur_e, last = e, n
        elif e > cur_e:
            busy += e - cur_e
            cur_e, last = e, n
    span = t1 - t0
    print(f"step {k}: span {span / 1e6:.3f} ms, busy (union of queues) {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms in {len(gaps)} gaps "
          f"(> 5 us: {sum(1 for g in gaps if g[0] > 5000)}, {sum(g[0] for g in gaps if g[0] > 5000) / 1e6:.3f} ms)")
    for g, a, b in sorted(gaps, reverse=True)[:12]:
        print(f"    {g / 1e3:7.1f} us  after {a}  before {b}")
