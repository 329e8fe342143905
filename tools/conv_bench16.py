"""bf16-operand conv micro-benchmark + check (act16 = 1: bf16 activations / fp32 weights, the round-2 kernels;
act16 = 2: bf16 activations AND weights, the LDS-DMA pipelined kernel).   python tools/conv_bench16.py [kind:B,H,W,C,K,k,s ...]"""
import ctypes as C, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.hip import ops
from dpft_amd.hip.lib import lib, make_desc, ptr, stream

DEFAULT = ["fwd:4,32,57,256,256,3,1", "dgrad:4,32,57,256,256,3,1", "fwd:4,32,57,256,1024,1,1", "fwd:4,32,57,1024,256,1,1",
           "dgrad:4,32,57,1024,256,1,1", "dgrad:4,32,57,256,1024,1,1", "fwd:4,128,228,64,64,3,1", "dgrad:4,128,228,64,64,3,1",
           "fwd:4,64,114,128,128,3,1", "dgrad:4,64,114,128,128,3,1", "fwd:8,32,57,256,256,3,1", "dgrad:8,32,57,256,256,3,1"]


def run(spec, mode, reps=20, check=True):
    kind, dims = spec.split(":")
    B, H, W, Cc, K, k, s = map(int, dims.split(","))
    pad = k // 2
    d = make_desc(B, H, W, Cc, K, k, k, s, pad)
    d.act16 = mode
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, Cc, generator=g).bfloat16().cuda()
    w = (torch.randn(K, k, k, Cc, generator=g) / (Cc * k * k) ** 0.5)
    dy = torch.randn(B, d.OH, d.OW, K, generator=g).bfloat16().cuda()
    w32 = w.cuda()
    wt32 = ops.weight_transpose(w32)
    wq, wtq = (w32.bfloat16(), wt32.bfloat16()) if mode == 2 else (w32, wt32)
    ws = torch.zeros(max(int(lib.dpft_conv2d_workspace_bytes(C.byref(d))), 16), dtype=torch.uint8, device="cuda")
    y = torch.empty(B, d.OH, d.OW, K, dtype=torch.bfloat16, device="cuda")
    dx = torch.empty(B, H, W, Cc, dtype=torch.bfloat16, device="cuda")
    dw = torch.empty(K, k, k, Cc, dtype=torch.float32, device="cuda")

    def go():
        if kind == "fwd":
            lib.call("dpft_conv2d_nhwc_fwd_f32", C.byref(d), ptr(x), ptr(wq), None, None, 0, ptr(y), None, ptr(ws), stream())
        elif kind == "wgrad":      # no prologue: DPFT_WGRAD16_PIPE=0 -> the round-2 kernel (A/B in two processes)
            lib.call("dpft_conv2d_nhwc_wgrad_f32", C.byref(d), ptr(x), ptr(dy), None, 0, ptr(dw), ptr(ws), stream())
        else:
            lib.call("dpft_conv2d_nhwc_dgrad_f32", C.byref(d), ptr(dy), ptr(wtq), ptr(dx), 0, ptr(ws), stream())
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    err = None
    if check:
        wr = (w32.bfloat16().double() if True else w32.double()).permute(0, 3, 1, 2).cpu()
        if kind == "fwd":
            ref = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), wr, stride=s, padding=pad).permute(0, 2, 3, 1)
            got = y.double().cpu()
        elif kind == "wgrad":
            wv = wr.clone().requires_grad_(True)
            F.conv2d(x.double().cpu().permute(0, 3, 1, 2), wv, stride=s, padding=pad).backward(dy.double().cpu().permute(0, 3, 1, 2))
            ref = wv.grad.permute(0, 2, 3, 1)
            got = dw.double().cpu()
        else:
            xin = torch.zeros(B, Cc, H, W, dtype=torch.float64, requires_grad=True)
            out = F.conv2d(xin, wr, stride=s, padding=pad)
            out.backward(dy.double().cpu().permute(0, 3, 1, 2))
            ref = xin.grad.permute(0, 2, 3, 1)
            got = dx.double().cpu()
        err = float((got - ref).norm() / ref.norm())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * B * d.OH * d.OW * K * k * k * Cc
    print(f"{spec:34s} act16={mode} {us:9.1f} us  {fl / us / 1e6:7.1f} TF  rel-L2 err {err}", flush=True)


if __name__ == "__main__":
    ops.conv_set_compute("bf16")
    specs = [a for a in sys.argv[1:] if ":" in a] or DEFAULT
    for mode in (1, 2):
        for sp in specs:
            run(sp, mode, check=os.environ.get("CHECK", "1") == "1")
