"""One-rank timing experiments that produce WRONG training (no gradient exchange / a local step decision) -- kept out of the
package and the library: this wrapper flips the two switches and then runs bench.py in-process.
usage: python tools/exp_switches.py [--skip-bucket-collectives] [--local-decision] [--standin-collective] -- <bench.py arguments>
--standin-collective: behind every bucket's (one-rank, forced) collective a kernel with the footprint of an 8-rank ring
all-reduce runs on the same stream (tools/probes/comm_standin.hip; STANDIN_BLOCKS=32, STANDIN_PASSES=2)
(the round-3 forced-collectives breakdown, docs/history used DPFT_EXP_* environment variables for the same; those are gone.)"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    argv = sys.argv[1:]
    rest = argv[argv.index("--") + 1:] if "--" in argv else []
    mine = argv[:argv.index("--")] if "--" in argv else argv
    from dpft_amd.training import distributed, trainer
    if "--skip-bucket-collectives" in mine:
        distributed.GradBucketReducer.exp_skip_bucket_collectives = True
    if "--standin-collective" in mine:
        import ctypes
        import subprocess
        import torch
        so = os.path.join(ROOT, "tools", "probes", "bin", "libcomm_standin.so")
        if not os.path.exists(so):
            os.makedirs(os.path.dirname(so), exist_ok=True)
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so,
                                   os.path.join(ROOT, "tools", "probes", "comm_standin.hip")])
        lib = ctypes.CDLL(so)
        lib.standin_launch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        blocks, passes = int(os.environ.get("STANDIN_BLOCKS", "32")), int(os.environ.get("STANDIN_PASSES", "2"))

        def standin(wire):
            rc = lib.standin_launch(wire.data_ptr(), wire.numel() * wire.element_size(), passes, blocks,
                                    torch.cuda.current_stream(wire.device).cuda_stream)
            assert rc == 0, rc
        distributed.GradBucketReducer.exp_after_collective = staticmethod(standin)
    if "--local-decision" in mine:
        init = trainer.DataParallelTrainer.__init__

        def patched(self, *a, **k):
            init(self, *a, **k)
            self._exp_local_decision = True
        trainer.DataParallelTrainer.__init__ = patched
    sys.argv = [os.path.join(ROOT, "bench.py")] + rest
    runpy.run_path(sys.argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
