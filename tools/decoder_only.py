"""N back-to-back calls of the fused inference decoder (one IMPFusion.forward each) on encoded synthetic pyramids --
the target of the decoder's rocprofv3 kernel-trace / PMC passes (roofline_decoder.traffic)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import decoder_runner
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch

reps = int(os.environ.get("REPS", "20"))
cfg = load_config("kradar")
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = build("dprt", cfg).to(dev)
data = make_batch(cfg["model"]["inputs"], int(os.environ.get("BATCH", "4")), device=dev)
run, _ = decoder_runner(model, data)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
print(f"decoder_fwd_us {e0.elapsed_time(e1) * 1e3 / reps:.1f} (reps {reps})")
