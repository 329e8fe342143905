"""Conv-family roofline recomputed from a rocprofv3 --kernel-trace --stats summary (the bench line's
`roofline.frac` must follow from what is committed under profiles/).

    python tools/roofline_from_rocprof.py <kernel_stats.csv> <steps> [algorithmic_gflop_per_step] [peak_tflops] [--families <conv_table>_families.json]

--families (round 6): the pipe-correct pricing of the bench line reproduced from the committed summary.  Every conv kernel NAME
belongs to one family -- x3 (conv_x3.hip: igemm_x3 / igemm_x3p / wgrad_x3), bf16 (igemm_pipe_kernel with its B16 template flag,
wgrad_pipe16, the BF16 instantiations of the vec kernels), vector (thin-channel / 16-channel / generic kernels: no matrix core),
f32 (everything else of the conv family: v_mfma_f32_32x32x2_f32) -- its time comes from the CSV, its algorithmic flops from the
JSON bench.py writes next to DPFT_CONV_TABLE (dpft_profile_get_family tags of the launches).  Output adds by_family,
split_share_of_conv_flops and frac_blended = sum_i(flops_i / peak_i) / conv time (peaks 157.3 | 416.7 | 2500 | 157.3 TF).

    python tools/roofline_from_rocprof.py --decoder <decoder_kernel_stats.csv> [iterations_per_forward] [algorithmic_MB]

--decoder: the second accounting of `roofline_decoder`: the SUM of the fused inference decoder's kernel
durations per forward (forwards = decoder_xattn launches / iterations; weight-pack kernels, which only run when a
parameter changed, are listed but not counted), priced against the SURVEY 8d bytes at 8 TB/s.  The bench line's
event-timed figure includes the gaps between the launches (8 since round 4) and excludes nothing; this one is kernels only.

Kernel time of the conv family per step = sum of TotalDurationNs of every kernel a dpft_conv2d_nhwc_* call launches
(main loops AND the split-K / slab reductions they need) / steps.  frac = algorithmic flops / that time / peak (157.3 TF)."""
import csv
import json
import re
import sys

MAIN = ("igemm_vec_kernel", "igemm_gen_kernel", "igemm_pipe_kernel", "igemm_x3_kernel", "igemm_x3p_kernel", "wgrad_x3_kernel", "wgrad_pipe_kernel", "wgrad_pipe16_kernel", "wgrad_vec_kernel", "wgrad_gen_kernel", "thin_dgrad_kernel",
        "conv16_", "wgrad16_", "thin_wgrad", "wgrad_stem7_kernel", "conv1x1_to16_kernel", "wgrad1x1_", "conv1x1_stream_kernel", "igemm_b16w_kernel")
AUX = ("splitk_reduce", "slab_reduce", "bias_grad_kernel", "zero_fill_kernel")
GROUPS = {"bn": ("bn_",), "decoder_train": ("sa_train", "xf_train", "hd_train", "pack_"),
          "decoder_infer": ("decoder_selfattn", "decoder_xattn"), "loss": ("match_cost", "set_loss", "giou3d", "lsap_batch"),
          "optimizer": ("adamw",), "weight_transpose": ("weight_transpose",), "fpn_misc": ("fpn_topdown", "add_pos", "add_inplace", "relu_bwd"),
          "vendor_aten": ("at::native", "Cijk_", "__amd_rocclr", "rocclr")}


def decoder_main(argv):
    path = argv[0]
    iters = int(argv[1]) if len(argv) > 1 else 4
    mb = float(argv[2]) if len(argv) > 2 else 555.917472          # kradar, B=4
    rows = list(csv.DictReader(open(path)))
    kern = {re.sub(r"\(.*", "", r["Name"]): (int(r["Calls"]), float(r["TotalDurationNs"])) for r in rows}
    xattn = [v for k, v in kern.items() if "decoder_xattn" in k]      # (+ decoder_xattn_last_kernel: the final iteration with the heads)
    fwds = sum(c for c, _ in xattn) / iters
    counted = {k: v for k, v in kern.items() if "decoder_" in k and "pack" not in k}
    total_ns = sum(ns for _, ns in counted.values())
    us = total_ns / fwds / 1e3
    out = {"source": path, "forwards": fwds, "iterations_per_forward": iters,
           "kernels": {k: {"calls_per_forward": c / fwds, "avg_us": ns / c / 1e3, "us_per_forward": ns / fwds / 1e3}
                       for k, (c, ns) in counted.items()},
           "not_counted": {k: {"calls": c, "total_us": ns / 1e3} for k, (c, ns) in kern.items() if k not in counted},
           "launches_per_forward": sum(c for c, _ in counted.values()) / fwds,
           "kernel_sum_us_per_forward": us, "algorithmic_mb": mb, "peak_gbps": 8000.0,
           "achieved_gbps": mb * 1e6 / (us * 1e-6) / 1e9, "frac": mb * 1e6 / (us * 1e-6) / 8.0e12}
    print(json.dumps(out, indent=1))


FAM_PEAK = {"f32": 157.3, "x3": 2500.0 / 6.0, "bf16": 2500.0, "vector": 157.3}


def family_of(name: str) -> str:
    """Kernel name (with template arguments) -> the pipe it runs on."""
    if "_x3_kernel" in name or "_x3p_kernel" in name:
        return "x3"
    if "wgrad_pipe16" in name or "igemm_b16w_kernel" in name:
        return "bf16"
    m = re.search(r"igemm_pipe_kernel<([^>]*)>", name)
    if m:
        a = [t.strip() for t in m.group(1).split(",")]
        return "bf16" if len(a) >= 8 and a[7] == "true" else "f32"
    m = re.search(r"(igemm_vec_kernel|wgrad_vec_kernel)<([^>]*)>", name)
    if m:
        a = [t.strip() for t in m.group(2).split(",")]
        idx = 7 if m.group(1) == "igemm_vec_kernel" else 5      # the BF16 template flag
        return "bf16" if len(a) > idx and a[idx] == "true" else "f32"
    if any(t in name for t in ("conv16_", "wgrad16_", "thin_", "igemm_gen", "wgrad_gen", "wgrad_stem7", "conv1x1_to16", "wgrad1x1_")):
        return "vector"
    return "f32"


def main():
    if sys.argv[1] == "--decoder":
        return decoder_main(sys.argv[2:])
    fam_json = None
    if "--families" in sys.argv:
        i = sys.argv.index("--families")
        fam_json = json.load(open(sys.argv[i + 1]))
        del sys.argv[i:i + 2]
    path, steps = sys.argv[1], int(sys.argv[2])
    gflop = float(sys.argv[3]) if len(sys.argv) > 3 else 1839.439164384       # kradar, B=4: 3 x 4 x 153.29 (bench.py log)
    peak = float(sys.argv[4]) if len(sys.argv) > 4 else 157.3      # fp32 MFMA; 2500 for the bf16 mode's summaries
    t = {"conv_main": 0.0, "conv_aux": 0.0, "other_dpft": 0.0}
    t.update({k: 0.0 for k in GROUPS})
    n = dict.fromkeys(t, 0)
    fam_ns = {}
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Name"])
        ns, calls = float(r["TotalDurationNs"]), int(r["Calls"])
        if any(m in name for m in MAIN):
            key = "conv_main"
            fam_ns[family_of(name)] = fam_ns.get(family_of(name), 0.0) + ns
        elif any(m in name for m in AUX):
            key = "conv_aux"
        else:
            key = next((g for g, pats in GROUPS.items() if any(p in name for p in pats)), "other_dpft")
        t[key] += ns
        n[key] += calls
    ms = {k: v / 1e6 / steps for k, v in t.items()}
    conv = ms["conv_main"] + ms["conv_aux"]
    out = {"source": path, "steps": steps, "kernel_ms_per_step": {k: round(v, 3) for k, v in ms.items()},
           "launches_per_step": {k: round(v / steps, 1) for k, v in n.items()},
           "algorithmic_gflop_per_step": gflop,
           "conv_family_tflops_main_only": gflop / ms["conv_main"], "conv_family_tflops": gflop / conv,
           "peak_tflops": peak, "frac_main_only": gflop / ms["conv_main"] / peak, "frac": gflop / conv / peak,
           "total_kernel_ms_per_step": round(sum(ms.values()), 3),
           "note_vendor_aten": "whole-run launches / steps: contains the set-up copies and fills (parameter flattening, bucket / arena "
                               "initialisation, warm-up); inside one steady-state step: profiles/r04_step_vendor_rows.txt (tools/step_vendor_rows.sh)"}
    if fam_json is not None:
        gf = fam_json["gflop_per_step"]
        ideal_ms = sum(gf[k] / FAM_PEAK[k] for k in gf)      # GFLOP / (TFLOP/s) = ms
        out["by_family"] = {k: {"gflop_per_step": gf.get(k, 0.0), "kernel_ms_per_step": round(fam_ns.get(k, 0.0) / 1e6 / steps, 3),
                                "tflops": (gf.get(k, 0.0) / (fam_ns[k] / 1e6 / steps)) if fam_ns.get(k) else None,
                                "peak_tflops": FAM_PEAK[k]} for k in sorted(set(gf) | set(fam_ns))}
        out["split_share_of_conv_flops"] = gf.get("x3", 0.0) / sum(gf.values())
        out["frac_blended"] = ideal_ms / conv
        out["frac_blended_main_only"] = ideal_ms / ms["conv_main"]
        out["families_source"] = fam_json.get("source")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
