"""Timing of one conv problem in bf16 compute mode with fp32 vs bf16 activation storage, with / without the BN prologue."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.hip import ops
from dpft_amd.hip.lib import lib, make_desc, ptr, stream
B, H, W, Cin, K, k, s = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,32,57,256,256,3,1").split(","))
dev = torch.device("cuda", 0)
ops.conv_set_compute("bf16")
pad = k // 2
res = {}
for a16 in (1, 0, 1, 0):
    d = make_desc(B, H, W, Cin, K, k, k, s, pad); d.act16 = a16
    dt = torch.bfloat16 if a16 else torch.float32
    x = torch.randn(B, H, W, Cin, device=dev).to(dt)
    dy = torch.randn(B, d.OH, d.OW, K, device=dev).to(dt)
    w = torch.randn(K, k, k, Cin, device=dev) * 0.05
    wt = ops.weight_transpose(w)
    bnp = torch.stack((torch.zeros(Cin), torch.ones(Cin), torch.zeros(Cin), torch.ones(Cin))).to(dev)
    y = torch.empty(B, d.OH, d.OW, K, dtype=dt, device=dev)
    dx = torch.empty(B, H, W, Cin, dtype=dt, device=dev)
    dw = torch.empty(K, k, k, Cin, device=dev)
    tr = C.c_int32(0)
    tiles = int(lib.dpft_conv2d_stats_tiles(C.byref(d), C.byref(tr)))
    stats = torch.empty(tiles, 2, K, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    runs = {"fwd pro+stats": lambda: lib.call("dpft_conv2d_nhwc_fwd_f32", C.byref(d), ptr(x), ptr(w), None, ptr(bnp), 1, ptr(y), ptr(stats), ptr(ws), stream()),
            "fwd plain": lambda: lib.call("dpft_conv2d_nhwc_fwd_f32", C.byref(d), ptr(x), ptr(w), None, None, 0, ptr(y), None, ptr(ws), stream()),
            "fwd pro": lambda: lib.call("dpft_conv2d_nhwc_fwd_f32", C.byref(d), ptr(x), ptr(w), None, ptr(bnp), 1, ptr(y), None, ptr(ws), stream()),
            "dgrad": lambda: lib.call("dpft_conv2d_nhwc_dgrad_f32", C.byref(d), ptr(dy), ptr(wt), ptr(dx), 0, ptr(ws), stream()),
            "wgrad pro": lambda: lib.call("dpft_conv2d_nhwc_wgrad_f32", C.byref(d), ptr(x), ptr(dy), ptr(bnp), 1, ptr(dw), ptr(ws), stream())}
    for name, fn in runs.items():
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        res[(name, a16)] = min(us, res.get((name, a16), 1e9))
fl = 2.0 * B * ((H + 2 * pad - k) // s + 1) * ((W + 2 * pad - k) // s + 1) * K * k * k * Cin
for name in ("fwd pro+stats", "fwd pro", "fwd plain", "dgrad", "wgrad pro"):
    print(f"{name:14s} fp32 storage {res[(name,0)]:7.1f} us ({fl/res[(name,0)]/1e6:6.1f} TF)   bf16 storage {res[(name,1)]:7.1f} us ({fl/res[(name,1)]/1e6:6.1f} TF)")
ops.conv_set_compute("fp32")
