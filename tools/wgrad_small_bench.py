"""Back-to-back GPU time of small weight-gradient problems (radar encoders) with preallocated buffers and a bare
ctypes call per launch (host cost ~3 us, so the loop is GPU-bound).  env DPFT_FORCE_WGRAD=tile,splits to sweep."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.hip import ops
from dpft_amd.hip.lib import lib, ptr, stream
SHAPES = sys.argv[1:] or ["4,16,7,256,1024,1,1", "4,16,7,256,256,3,1", "4,16,7,1024,256,1,1", "4,32,14,128,128,3,1",
                          "4,5,14,128,128,3,1", "4,3,7,256,1024,1,1", "4,3,7,256,256,3,1", "4,8,4,512,512,3,1",
                          "4,64,27,64,64,3,1", "4,32,14,512,128,1,1"]
fn = lib.load().dpft_conv2d_nhwc_wgrad_f32
for spec in SHAPES:
    B, H, W, Cc, K, k, s = map(int, spec.split(","))
    cv = ops.conv_problem(B, H, W, Cc, K, k, k, s, k // 2)
    x = torch.randn(B, H, W, Cc, device="cuda"); dy = torch.randn(B, cv.OH, cv.OW, K, device="cuda")
    dw = torch.empty(K, k, k, Cc, device="cuda"); ws = ops.workspace(cv.ws_bytes, x.device)
    bnp = torch.stack((torch.zeros(Cc), torch.ones(Cc), torch.zeros(Cc), torch.ones(Cc))).cuda()
    args = (C.byref(cv.desc), ptr(x), ptr(dy), ptr(bnp), 1, ptr(dw), ptr(ws), stream())
    for _ in range(5): fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200): fn(*args)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    print(f"{spec:24s} force={os.environ.get('DPFT_FORCE_WGRAD','auto'):8s} {us:7.1f} us {2.0*cv.M*K*k*k*Cc/us/1e6:6.1f} TF", flush=True)
