"""`python bench.py --gpus N` as the driver calls it (no launcher around it): the N ranks are started by bench.py
itself, as fresh child processes, before anything has touched the GPU.  CPU-side tests of that launcher: the command
it builds, the environment it hands down, argument pass-through, how the library-owned RCCL step is chosen, and that a
failing child fails the run.  (north_star: throughput "at 1, 2, 4 and 8 GPUs"; the reduction that is distributed:
/root/reference/src/radtran/clima_radtran_radiate.f90:184-192.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class FakeRun:
    """Stands in for subprocess.run: records the calls, answers the probe and the run proper in turn."""

    def __init__(self, probe_rc=0, probe_out=b"native step ok: ...\n", run_rc=0, probe_timeout=False):
        self.calls = []
        self.probe_rc, self.probe_out, self.run_rc, self.probe_timeout = probe_rc, probe_out, run_rc, probe_timeout

    def __call__(self, cmd, env=None, timeout=None, stdout=None, stderr=None):
        self.calls.append(dict(cmd=list(cmd), env=dict(env or {}), timeout=timeout))
        if "--probe-native" in cmd:
            if self.probe_timeout:
                raise subprocess.TimeoutExpired(cmd, timeout)
            return subprocess.CompletedProcess(cmd, self.probe_rc, stdout=self.probe_out, stderr=b"")
        return subprocess.CompletedProcess(cmd, self.run_rc)


def test_launcher_command_is_the_drivers_form():
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "20", "--warmup", "5"], port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"      # the container's hostname may not resolve
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]   # arguments pass through unchanged


def test_too_few_devices_is_a_one_line_failure(capsys):
    run = FakeRun()
    rc = bench.self_launch(8, ["--gpus", "8"], run=run, ndev=1)
    assert rc != 0 and run.calls == []
    err = capsys.readouterr().err.strip().splitlines()
    assert len(err) == 1 and "--gpus 8" in err[0] and "1 GPU" in err[0]


def test_native_step_is_used_only_after_a_clean_probe(monkeypatch):
    monkeypatch.delenv("CLIMA_BENCH_NATIVE_ALLREDUCE", raising=False)
    monkeypatch.delenv("CLIMA_BENCH_TORCH_ALLREDUCE", raising=False)
    run = FakeRun()
    assert bench.self_launch(2, ["--gpus", "2", "--steps", "7"], run=run, ndev=2) == 0
    assert len(run.calls) == 2
    probe, real = run.calls
    assert "--probe-native" in probe["cmd"] and probe["timeout"] and probe["env"]["CLIMA_BENCH_NATIVE_ALLREDUCE"] == "1"
    assert "--probe-native" not in real["cmd"] and real["cmd"][-4:] == ["--gpus", "2", "--steps", "7"]
    assert real["env"]["CLIMA_BENCH_NATIVE_ALLREDUCE"] == "1"
    assert real["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"   # dmabuf IPC only on this pool


@pytest.mark.parametrize("kw", [dict(probe_rc=1), dict(probe_out=b"native step DIFFERS\n"), dict(probe_timeout=True)])
def test_unclean_probe_keeps_torchs_all_reduce(monkeypatch, kw):
    monkeypatch.delenv("CLIMA_BENCH_NATIVE_ALLREDUCE", raising=False)
    monkeypatch.delenv("CLIMA_BENCH_TORCH_ALLREDUCE", raising=False)
    run = FakeRun(**kw)
    assert bench.self_launch(2, ["--gpus", "2"], run=run, ndev=8) == 0
    assert run.calls[-1]["env"]["CLIMA_BENCH_NATIVE_ALLREDUCE"] == "0"


def test_an_explicit_choice_skips_the_probe(monkeypatch):
    monkeypatch.setenv("CLIMA_BENCH_TORCH_ALLREDUCE", "1")
    run = FakeRun()
    bench.self_launch(2, ["--gpus", "2"], run=run, ndev=2)
    assert len(run.calls) == 1 and "--probe-native" not in run.calls[0]["cmd"]


def test_a_failing_child_fails_the_run(monkeypatch):
    monkeypatch.setenv("CLIMA_BENCH_NATIVE_ALLREDUCE", "0")
    run = FakeRun(run_rc=7)
    assert bench.self_launch(2, ["--gpus", "2"], run=run, ndev=2) == 7


def test_bench_gpus_2_on_a_box_without_two_gpus_exits_nonzero():
    """The real entry point, as the driver types it; this container has no GPU."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible: this would start a real run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "--gpus 2" in r.stderr and "visible" in r.stderr


def test_physical_cores_is_stated():
    n, how = bench.physical_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and "affinity" in how
