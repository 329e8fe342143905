# Ordered kernel sequence of the captured training-decoder graphs (forward and backward) of one step.
cd /tmp && export TMPDIR=/tmp
STEPS=2 timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/proft5 -- python /root/repo/tools/train_only.py </dev/null > /tmp/proft5.log 2>&1
f=$(find /tmp/proft5 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
step = rows[idx[-2] + 1: idx[-1] + 1]
def short(n):
    n = n.split("(")[0].replace("void ", "")
    n = re.sub(r"at::native::(\(anonymous namespace\)::)?", "aten:", n)
    n = re.sub(r"<.*", "", n)
    return n[:44]
dec = [i for i, r in enumerate(step) if any(p in r["Kernel_Name"] for p in ("sa_train", "xf_train", "hd_train"))]
lo, hi = (0, len(step)) if __import__("os").environ.get("WHOLE") else (dec[0] - int(__import__("os").environ.get("BEFORE", "12")), dec[-1] + 12)
seq = []
for r in step[max(lo, 0):hi]:
    n = short(r["Kernel_Name"])
    if any(p in n for p in ("igemm", "wgrad", "bn_", "splitk", "slab", "conv16", "transpose", "thin_", "fpn_", "add_inplace", "bias_grad")):
        n = "(conv/bn kernel)" if __import__("os").environ.get("WHOLE") is None else n
    if seq and seq[-1][0] == n: seq[-1][1] += 1
    else: seq.append([n, 1])
for n, c in seq: print(f"{c:4d} x {n}")
PY
