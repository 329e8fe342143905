cd /tmp && export TMPDIR=/tmp
STEPS=2 timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/proft2 -- python /root/repo/tools/train_only.py </dev/null > /tmp/proft2.log 2>&1
f=$(find /tmp/proft2 -name "*kernel_trace.csv" | head -1)
mkdir -p /root/repo/gpurun_out
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last step = after the last adamw kernel but one
idx = [i for i, n in enumerate(names) if "adamw_kernel" in n]
lo, hi = idx[-2] + 1, idx[-1] + 1
def short(n):
    n = re.sub(r"at::native::|\(anonymous namespace\)::|void ", "", n)
    return n[:90]
out = []
prev, cnt = None, 0
for r in rows[lo:hi]:
    s = short(r["Kernel_Name"]) + " q" + r.get("Queue_Id", "?")
    if s == prev:
        cnt += 1
    else:
        if prev is not None:
            out.append(f"{cnt:4d} x {prev}")
        prev, cnt = s, 1
out.append(f"{cnt:4d} x {prev}")
open("/root/repo/gpurun_out/step_sequence.txt", "w").write("\n".join(out))
print(len(rows[lo:hi]), "kernels in the step;", len(out), "runs")
# GPU busy time (union of kernel intervals) over the step, and the biggest idle gaps
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows[lo:hi])
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
busy, cur_s, cur_e, gaps, last_name = 0, iv[0][0], iv[0][1], [], iv[0][2]
for s_, e_, n_ in iv[1:]:
    if s_ > cur_e:
        busy += cur_e - cur_s
        gaps.append((s_ - cur_e, last_name, n_))
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
    last_name = n_
busy += cur_e - cur_s
gaps.sort(reverse=True)
with open("/root/repo/gpurun_out/step_gaps.txt", "w") as f:
    f.write(f"step span {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms, gaps>20us: {sum(1 for g in gaps if g[0] > 20000)}\n")
    tot_small = sum(g[0] for g in gaps if g[0] <= 20000)
    f.write(f"idle in gaps <= 20us: {tot_small / 1e6:.2f} ms over {sum(1 for g in gaps if g[0] <= 20000)} gaps\n")
    for g in gaps[:40]:
        f.write(f"{g[0] / 1e3:8.1f} us  after [{g[1][:60]}] before [{g[2][:60]}]\n")
print(open("/root/repo/gpurun_out/step_gaps.txt").read()[:3000])
PY
