"""Time / HBM roofline of the radar tesseract projection + the numpy reference timing on the host."""
import os, sys, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.data import radar_projection
from oracle import radar_oracle as RO
rs = np.random.RandomState(0)
t = (10.0 ** (rs.rand(64, 256, 37, 107) * 12.0 + 4.0)).astype(np.float32)
td = torch.from_numpy(t).cuda()
for _ in range(3): radar_projection(td)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): radar_projection(td)
e1.record(); torch.cuda.synchronize()
dt = e0.elapsed_time(e1) / 10 * 1e-3
# yardstick: one streaming read of the same cube by the vendor's reduction kernel (what "HBM-bound" means on this box)
for _ in range(3): td.sum()
torch.cuda.synchronize(); e0.record()
for _ in range(10): td.sum()
e1.record(); torch.cuda.synchronize()
dt_sum = e0.elapsed_time(e1) / 10 * 1e-3
byts = t.nbytes
a = time.perf_counter(); RO.radar_projection(t, np.linspace(-1.9, 1.9, 64)); cpu = time.perf_counter() - a
print(json.dumps({"cube_mb": byts / 1e6, "gpu_ms_per_cube": dt * 1e3, "cubes_per_s": 1 / dt,
                  "achieved_GBs_algorithmic": byts / dt / 1e9, "frac_of_8TBs": byts / dt / 8e12,
                  "torch_sum_one_read_us": dt_sum * 1e6, "torch_sum_GBs": byts / dt_sum / 1e9, "numpy_reference_s_per_cube": cpu, "host_threads": torch.get_num_threads()}))
