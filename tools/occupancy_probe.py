"""Would more resident waves help the layer-3 convs?  Run the same conv on ONE stream and on TWO streams concurrently
(independent buffers): if two concurrent launches take clearly less than 2x one launch, the kernel is latency-bound at
its 1.78 workgroups per CU and a form with more waves per tile (intra-workgroup split-K) would pay."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.hip import ops
dev = torch.device("cuda", 0)
for spec in [("fwd", 4, 32, 57, 256, 256, 3), ("dgrad", 4, 32, 57, 256, 256, 3), ("fwd", 4, 32, 57, 1024, 256, 1), ("fwd", 4, 32, 57, 256, 1024, 1),
             ("wgrad", 4, 32, 57, 256, 256, 3)]:
    kind, B, H, W, C, K, k = spec
    cv = ops.conv_problem(B, H, W, C, K, k, k, 1, k // 2)
    bufs = []
    for _ in range(2):
        x = torch.randn(B, H, W, C, device=dev); w = torch.randn(K, k, k, C, device=dev) * 0.05
        dy = torch.randn(B, H, W, K, device=dev); wt = ops.weight_transpose(w)
        bufs.append((x, w, dy, wt))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    def go(i):
        x, w, dy, wt = bufs[i]
        if kind == "fwd": ops.conv_fwd(cv, x, w)
        elif kind == "dgrad": ops.conv_dgrad(cv, dy, wt)
        else: ops.conv_wgrad(cv, x, dy)
    def timed(n_streams, reps=20):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in streams[:n_streams]: s.wait_event(e0)
        for _ in range(reps):
            for i in range(n_streams):
                with torch.cuda.stream(streams[i]): go(i)
        for s in streams[:n_streams]: torch.cuda.current_stream().wait_stream(s)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    for i in range(2):
        with torch.cuda.stream(streams[i]): go(i)
    t1, t2 = timed(1), timed(2)
    print(f"{spec}: one stream {t1:6.1f} us/launch, two streams {t2:6.1f} us per PAIR ({t2 / 2:6.1f} per launch) -> ratio {t2 / t1:.2f}")
