"""Back-to-back GPU time of small forward / data-gradient convs (radar encoders), preallocated buffers, bare ctypes
calls (GPU-bound loop).  env DPFT_FORCE_TILE=bm,bn,splits to sweep."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.hip import ops
from dpft_amd.hip.lib import lib, ptr, stream
SHAPES = sys.argv[1:] or ["fwd:4,16,7,256,256,3,1", "fwd:4,16,7,1024,256,1,1", "fwd:4,16,7,256,1024,1,1", "fwd:4,3,7,256,256,3,1",
                          "fwd:4,3,7,1024,256,1,1", "fwd:4,8,4,512,2048,1,1", "fwd:4,32,14,128,128,3,1", "fwd:4,64,27,64,64,3,1",
                          "dgrad:4,16,7,256,256,3,1", "dgrad:4,16,7,256,1024,1,1", "dgrad:4,3,7,256,1024,1,1", "dgrad:4,32,14,128,128,3,1"]
L = lib.load()
for spec in SHAPES:
    kind, dims = spec.split(":")
    B, H, W, Cc, K, k, s = map(int, dims.split(","))
    cv = ops.conv_problem(B, H, W, Cc, K, k, k, s, k // 2)
    x = torch.randn(B, H, W, Cc, device="cuda"); dy = torch.randn(B, cv.OH, cv.OW, K, device="cuda")
    w = torch.randn(K, k, k, Cc, device="cuda") * 0.05; wt = ops.weight_transpose(w)
    y = torch.empty(B, cv.OH, cv.OW, K, device="cuda"); dx = torch.empty_like(x)
    stats = torch.empty(cv.M // 32 + 2, 2, K, device="cuda"); ws = ops.workspace(max(cv.ws_bytes, 64 << 20), x.device)
    bnp = torch.stack((torch.zeros(Cc), torch.ones(Cc), torch.zeros(Cc), torch.ones(Cc))).cuda()
    if kind == "fwd":
        fn, args = L.dpft_conv2d_nhwc_fwd_f32, (C.byref(cv.desc), ptr(x), ptr(w), None, ptr(bnp), 1, ptr(y), ptr(stats), ptr(ws), stream())
    else:
        fn, args = L.dpft_conv2d_nhwc_dgrad_f32, (C.byref(cv.desc), ptr(dy), ptr(wt), ptr(dx), 0, ptr(ws), stream())
    for _ in range(5): fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200): fn(*args)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    print(f"{spec:28s} force={os.environ.get('DPFT_FORCE_TILE','auto'):9s} {us:7.1f} us {2.0*cv.M*K*k*k*Cc/us/1e6:6.1f} TF", flush=True)
