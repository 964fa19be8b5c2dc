"""The library-owned RCCL step across TWO real ranks, one fresh process per GPU (it had only ever run with
one-rank communicators and rehearsed shards).  Skipped where fewer than two devices are visible -- every box of this
pool so far; it is here for the first node that has them, and `python bench.py --gpus N` runs the same probe by itself
before it decides which all-reduce the measured steps use (bench.self_launch).

What the ranks check (bench.probe_native): level rows and f_total of the library's step -- shard -> kernels ->
ncclAllReduce(4(nz+1)+1 f64) on the handle's stream (the sum over bins of
/root/reference/src/radtran/clima_radtran_radiate.f90:184-192 distributed over the ranks) -- against torch's all-reduce
of the same partial rows: after a full step, after an IR-only step (the partial solar rows are put back before the
reduce), and after a hand-off timeout forced on rank 0 alone (the status word makes every rank repeat the step)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def test_native_step_on_two_ranks():
    import torch
    if torch.cuda.device_count() < 2:      # (counting devices does not initialise the GPU in this process)
        pytest.skip("needs two GPUs: %d visible" % torch.cuda.device_count())
    import bench
    env = dict(os.environ, CLIMA_BENCH_NATIVE_ALLREDUCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run(bench.launcher_command(2, ["--gpus", "2", "--probe-native"]), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "native step ok" in r.stdout, r.stdout[-2000:]


def test_the_probe_itself_with_one_rank():
    """The same probe with a one-rank process group (CLIMA_BENCH_FORCE_DIST=1): every line of it runs on a one-GPU box --
    torch's all-reduce of the partial rows, the library's communicator, the IR-only step, the forced repeat."""
    env = dict(os.environ, CLIMA_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--probe-native"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "native step ok" in r.stdout and "repeated steps 1" in r.stdout, r.stdout[-2000:]
