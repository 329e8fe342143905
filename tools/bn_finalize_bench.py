"""Latency of bn_finalize over the camera encoder's (tiles, channels) combinations (bare ctypes calls, GPU-bound loop)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.hip.lib import lib, ptr, stream
fn = lib.load().dpft_bn_finalize_f32
tot = 0.0
for tiles, rows, K, calls in [(1824, 64, 64, 7), (912, 128, 256, 4), (456, 64, 128, 8), (228, 128, 512, 5), (114, 64, 256, 46), (57, 128, 1024, 24),
                             (29, 64, 512, 6), (29, 64, 2048, 4)]:
    M = tiles * rows
    st = torch.rand(tiles, 2, K, device="cuda"); g = torch.ones(K, device="cuda"); b = torch.zeros(K, device="cuda")
    rm = torch.zeros(K, device="cuda"); rv = torch.ones(K, device="cuda"); bnp = torch.empty(4, K, device="cuda")
    args = (ptr(st), tiles, rows, C.c_int64(M), K, ptr(g), ptr(b), C.c_float(1e-5), C.c_float(0.1), ptr(rm), ptr(rv), ptr(bnp), stream())
    for _ in range(5): fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200): fn(*args)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    tot += us * calls
    print(f"tiles {tiles:5d} K {K:5d}: {us:6.1f} us", end=" | ")
print(f"\ncamera total per step {tot/1e3:.2f} ms")
