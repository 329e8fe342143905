"""dpft_amd -- MI355X-native (gfx950) hot path of DPFT behind the ``dprt`` model/config API.

Public surface (mirrors src/dprt/models/__init__.py:10-18 of the reference):
    dpft_amd.models.build(name, config) -> torch.nn.Module
    dpft_amd.models.load(checkpoint)    -> (module, epoch, timestamp)
    dpft_amd.models.dprt.DPRT           (from_config / forward)
All device compute goes through ``libdpft_hip.so`` (C-ABI in include/dpft_hip.h); there is no CPU
fallback: running a forward without the HIP library / a GPU raises.
"""
import os as _os

# see bench.py: device-resident kernel arguments matter for a launch-bound step; a no-op where the runtime already
# defaults to it, and only effective if set before the HIP runtime initialises (first CUDA call)
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

__version__ = "0.1.0"
