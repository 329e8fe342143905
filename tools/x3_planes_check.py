"""Planes-input split convolution (igemm_x3p_kernel) vs the fp32 MFMA path and fp64: error and time per shape / tile."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.hip import ops
import torch.nn.functional as F

def one(kind, B, H, W, C, K, k, s, check=True):
    pad = k // 2
    cv = ops.conv_problem(B, H, W, C, K, k, k, s, pad)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, C, generator=g).cuda()
    w = (torch.randn(K, k, k, C, generator=g) / (C * k * k) ** 0.5).cuda()
    dy = torch.randn(B, cv.OH, cv.OW, K, generator=g).cuda()
    wt = ops.weight_transpose(w)
    xp, wp, dyp, wtp = (ops.split_planes(t) for t in (x, w, dy, wt))
    assert torch.equal(xp.float().sum(0), x), "planes do not add up"
    if kind == "fwd":
        f32 = lambda: ops.conv_fwd(cv, x, w, want_stats=True)[0]
        x3 = lambda: ops.conv_fwd(cv, x, w, want_stats=True, planes=(xp, wp))[0]
    else:
        f32 = lambda: ops.conv_dgrad(cv, dy, wt)
        x3 = lambda: ops.conv_dgrad(cv, dy, wt, planes=(dyp, wtp))
    a, b = f32(), x3()
    msg = ""
    if check:
        if kind == "fwd":
            ref = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu().permute(0, 3, 1, 2), stride=s, padding=pad).permute(0, 2, 3, 1)
        else:
            ref = torch.nn.grad.conv2d_input((B, C, H, W), w.double().cpu().permute(0, 3, 1, 2), dy.double().cpu().permute(0, 3, 1, 2), stride=s, padding=pad).permute(0, 2, 3, 1)
        e = lambda t: float((t.double().cpu() - ref).norm() / ref.norm())
        msg = f"err fp32 {e(a):.2e} planes {e(b):.2e}"
    def tm(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 50
    ta, tb = tm(f32), tm(x3)
    fl = 2.0 * cv.M * K * k * k * C
    print(f"{kind}:{B},{H},{W},{C},{K},{k},{s} tile={os.environ.get('DPFT_FORCE_TILE','auto'):9s} fp32 {ta:7.1f} us {fl/ta/1e6:6.1f} TF | planes {tb:7.1f} us {fl/tb/1e6:6.1f} TF  x{ta/tb:.2f}  {msg}", flush=True)

SPECS = ["fwd:4,32,57,256,256,3,1", "dgrad:4,32,57,256,256,3,1", "fwd:4,32,57,256,1024,1,1", "fwd:4,32,57,1024,256,1,1", "dgrad:4,32,57,1024,256,1,1",
         "dgrad:4,32,57,256,1024,1,1", "fwd:4,128,228,64,256,1,1", "fwd:4,128,228,64,64,3,1", "fwd:4,64,114,128,128,3,1", "fwd:4,16,29,512,512,3,1",
         "dgrad:4,128,228,256,64,1,1", "dgrad:4,64,114,512,128,1,1", "dgrad:4,32,57,512,512,3,2", "fwd:4,16,7,256,256,3,1"]
for sp in (sys.argv[1:] or SPECS):
    kind, dims = sp.split(":")
    one(kind, *map(int, dims.split(",")), check=os.environ.get("NOCHECK") != "1")
