"""ATen ops on device tensors inside ONE eval forward at batch 1 (each is a vendor launch between our kernels)."""
import collections, os, sys, torch, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.utils._python_dispatch import TorchDispatchMode
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch
cfg = load_config("kradar")
dev = torch.device("cuda", 0)
model = build("dprt", cfg).to(dev).eval()
data = make_batch(cfg["model"]["inputs"], int(os.environ.get("BATCH", "1")), device=dev)
seen = collections.Counter()
where = {}


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        ts = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
        if any(t.is_cuda for t in ts) and name not in ("view", "_unsafe_view", "detach", "permute", "select", "slice", "as_strided",
                                                       "reshape", "unsqueeze", "squeeze", "expand", "alias", "transpose", "t",
                                                       "split", "unbind", "movedim", "narrow", "split_with_sizes", "lift_fresh"):
            seen[name] += 1
            fr = [f for f in traceback.extract_stack() if "dpft_amd" in f.filename]
            if fr:
                where.setdefault(name, collections.Counter())[f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}"] += 1
        return func(*args, **(kwargs or {}))


with torch.no_grad():
    for _ in range(5):
        model(data)
    with Spy():
        model(data)
torch.cuda.synchronize()
for k, v in seen.most_common():
    print(k, v, dict(where.get(k, {})))
