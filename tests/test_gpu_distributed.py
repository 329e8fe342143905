"""World-size-2 data parallel on ONE GPU (gloo transport, both ranks on cuda:0): the real GPU training path --
view streams, side-stream weight gradients, decoder hipGraphs adding into the reducer buckets, bucket all-reduce --
must keep the two replicas bit-identical.  (RCCL itself cannot put two ranks on one device; the N>1 RCCL runs are
the driver's.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("graphs", ["1", "0"])
def test_two_ranks_one_gpu_stay_in_sync(graphs):
    env = dict(os.environ, GRAPHS=graphs, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29500 + (os.getpid() % 200) + (7 if graphs == "1" else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "dp2_gloo_gpu.py")]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "dp2 gloo-on-GPU OK" in res.stdout
