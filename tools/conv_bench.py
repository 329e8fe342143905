"""Micro-benchmark of single conv problems through the C-ABI (tuning aid, not part of the product).
usage: python tools/conv_bench.py [kind:B,H,W,C,K,k,s ...]   (env DPFT_FORCE_TILE=bm,bn,splits)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.hip import ops

DEFAULT = ["fwd:4,32,57,256,256,3,1", "dgrad:4,32,57,256,256,3,1", "wgrad:4,32,57,256,256,3,1",
           "fwd:4,32,57,256,1024,1,1", "fwd:4,32,57,1024,256,1,1", "fwd:4,128,228,64,256,1,1",
           "fwd:4,128,228,64,64,3,1", "fwd:4,64,114,128,128,3,1", "fwd:4,16,29,512,512,3,1"]


def run(spec, reps=20, pro=True, stats=True):
    kind, dims = spec.split(":")
    B, H, W, C, K, k, s = map(int, dims.split(","))
    pad = k // 2
    cv = ops.conv_problem(B, H, W, C, K, k, k, s, pad)
    x = torch.randn(B, H, W, C, device="cuda")
    w = torch.randn(K, k, k, C, device="cuda") * 0.05
    dy = torch.randn(B, cv.OH, cv.OW, K, device="cuda")
    bnp = torch.stack((torch.zeros(C), torch.ones(C), torch.zeros(C), torch.ones(C))).cuda() if (pro and C % 32 == 0) else None
    wt = ops.weight_transpose(w)
    planes = os.environ.get("PLANES") == "1" and C % 64 == 0 and kind != "wgrad"      # operands as three bf16 planes (conv_x3.hip)
    if planes:
        xp, wp, dyp, wtp = ops.split_planes(x), ops.split_planes(w), ops.split_planes(dy), ops.split_planes(wt)
        bnp = None
    def go():
        if kind == "fwd":
            ops.conv_fwd(cv, x, w, pro=(bnp, True) if bnp is not None else None, want_stats=stats, planes=(xp, wp) if planes else None)
        elif kind == "dgrad":
            ops.conv_dgrad(cv, dy, wt, planes=(dyp, wtp) if planes else None)
        else:
            ops.conv_wgrad(cv, x, dy, pro=(bnp, True) if bnp is not None else None)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * cv.M * K * k * k * C
    print(f"{spec:34s} tile={os.environ.get('DPFT_FORCE_TILE','auto'):10s} {us:9.1f} us  {fl / us / 1e6:7.1f} TF", flush=True)


if __name__ == "__main__":
    if os.environ.get("DPFT_COMPUTE"):      # fp32 | bf16
        ops.conv_set_compute(os.environ["DPFT_COMPUTE"])
    specs = sys.argv[1:] or DEFAULT
    for sp in specs:
        run(sp, pro=os.environ.get("NOPRO") != "1", stats=os.environ.get("NOSTATS") != "1")
