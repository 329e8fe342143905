# GPU-side gaps around the loss section (kernel trace): end of the forward decoder graph -> match_cost -> set_loss -> first backward kernel
cd /tmp && export TMPDIR=/tmp
STEPS=6 timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/proft8 -- python /root/repo/tools/train_only.py </dev/null > /tmp/proft8.log 2>&1
f=$(find /tmp/proft8 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
for k in range(-4, -1):
    step = rows[idx[k] + 1: idx[k + 1] + 1]
    def first(pat, start=0):
        return next(i for i in range(start, len(step)) if pat in step[i]["Kernel_Name"])
    i_hd = max(i for i, r in enumerate(step) if "hd_train_fwd" in r["Kernel_Name"])
    i_mc = first("match_cost")
    i_sl = first("set_loss", i_mc)
    i_sb = first("set_loss", i_sl + 1)
    i_hb = first("hd_train_bwd", i_sb)
    S = lambda i: int(step[i]["Start_Timestamp"]); E = lambda i: int(step[i]["End_Timestamp"])
    print(f"fwd graph end -> match_cost start {(S(i_mc) - E(i_hd)) / 1e3:7.1f} us ({i_mc - i_hd - 1} kernels between) | "
          f"match_cost end -> set_loss fwd start {(S(i_sl) - E(i_mc)) / 1e3:7.1f} us | "
          f"set_loss fwd end -> set_loss bwd start {(S(i_sb) - E(i_sl)) / 1e3:7.1f} us ({i_sb - i_sl - 1} between) | "
          f"set_loss bwd end -> hd_train_bwd start {(S(i_hb) - E(i_sb)) / 1e3:7.1f} us ({i_hb - i_sb - 1} between)")
PY
