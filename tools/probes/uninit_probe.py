"""Does a train step read memory it never wrote?  Poisons the caching allocator's pool (64 x 256 MiB of 3e38 / NaN), then
runs the reduced radar-front train-parity case of tests/test_gpu_model.py.  usage: python tools/probes/uninit_probe.py [nan]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
val = float("nan") if len(sys.argv) > 1 and sys.argv[1] == "nan" else 3e38
bufs = [torch.full((64 << 20,), val, dtype=torch.float32, device="cuda") for _ in range(int(os.environ.get("POISON_BLOCKS", "48")))]
small = [torch.full((n,), val, dtype=torch.float32, device="cuda") for n in (256, 4096, 65536, 1 << 20) for _ in range(64)]
torch.cuda.synchronize()
del bufs, small
import test_gpu_model as T
name = os.environ.get("CFG", "kradar_radar_front")
try:
    T._train_parity(T.view_config(name, dropout=0.0), seed=32, batch=6 if name == "kradar_radar_front" else 2)
    print("PASS")
except AssertionError as e:
    print("FAIL", str(e)[:600])
