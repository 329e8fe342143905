"""Helper of test_host.py::test_every_compute_entry_rejects_null_arguments (runs in its own process: a missing
argument check would be a segfault, not an exception).  Calls every compute entry of the C-ABI with NULL pointers and
zero sizes and prints {name: [return code, dpft_last_error()]} as JSON."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.hip.lib import SIGNATURES, lib   # noqa: E402

out = {}
for name in sorted(SIGNATURES):
    if not (name.endswith("_f32") or name.endswith("_u8") or name in ("dpft_resnet_forward", "dpft_resnet_backward_stage")):
        continue
    _, args = SIGNATURES[name]
    vals = [0 if a in (C.c_int32, C.c_int64) else (0.0 if a is C.c_float else None) for a in args]
    rc = getattr(lib.load(), name)(*vals)
    out[name] = [int(rc), lib.dpft_last_error().decode("utf-8", "replace")]
print(json.dumps(out))
