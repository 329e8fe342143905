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


@pytest.mark.gpu
def test_bench_multi_rank_launch_line():
    """The driver's N>1 launch line of bench.py (torch.distributed.run, one JSON line from rank 0), with both ranks on
    this box's single GPU over gloo (DPFT_BENCH_ONE_DEVICE_GLOO): every rank must pass through every collective --
    including the profiled step after the timed region -- and the line must carry the whole-job aggregate."""
    import json
    env = dict(os.environ, DPFT_BENCH_ONE_DEVICE_GLOO="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29750 + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "2", "--latency-reps", "5"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 8 and line["config"]["parallelism"] == "dp2"
    assert abs(line["value"] - 8 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    assert line["roofline"]["bound"] == "mfma" and 0 < line["roofline"]["frac"] < 1
    assert line["cpu_baseline"] is None          # the CPU leg is an N=1 measurement


@pytest.mark.gpu
@pytest.mark.parametrize("graphs,wire", [("1", "fp32"), ("0", "fp32"), ("1", "bf16")])
def test_rccl_one_rank_forced_collectives_match_plain_step(graphs, wire):
    """The RCCL branch of the reducer / trainer (backend "nccl", every bucket through dist.all_reduce on RCCL's stream,
    ncclAvg, bf16 wire staging, decoder graphs captured beside the RCCL watchdog) executed on the single GPU with a
    one-rank communicator; must reproduce the plain step (tools/rccl1_forced.py)."""
    env = dict(os.environ, GRAPHS=graphs, WIRE=wire, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl1_forced.py")], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "rccl1 forced-collectives OK" in res.stdout


@pytest.mark.gpu
def test_bench_forced_collectives_line():
    """bench.py --gpus 1 --force-collectives: the line says so (rccl_ranks 1, collectives_forced true)."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
           "--latency-reps", "5", "--force-collectives", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["collectives_forced"] is True
    assert line["exposed_allreduce_ms"] >= 0.0 and line["value"] > 0
