"""One-rank timing experiments that produce WRONG training (no gradient exchange / a local step decision) -- kept out of the
package and the library: this wrapper flips the two switches and then runs bench.py in-process.
usage: python tools/exp_switches.py [--skip-bucket-collectives] [--local-decision] -- <bench.py arguments>
(tools/r03_forced_breakdown.sh used DPFT_EXP_* environment variables for the same; those are gone.)"""
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
