"""bench.py --gpus N: the parent starts N ranks itself (no WORLD_SIZE in the environment), relays ONE JSON line and
reports the number of ranks that really joined the process group.  On the CPU the ranks rehearse the rendezvous and a
gradient-sized exchange over gloo without a device (VT_BENCH_DRY=1); on the one-GPU box two ranks share the device and
the real training step runs with its exchange staged through gloo (RCCL refuses two ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                         timeout=timeout)
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    return out, lines


def test_gpus_flag_starts_that_many_ranks_dry():
    out, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"VT_BENCH_DRY": "1"})
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(lines) == 1, lines  # exactly one line on stdout: rank 0's
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["grad_allreduce_floats"] == 33512
    assert r["config"]["global_batch"] == 2 * r["config"]["per_gpu_batch"] and r["config"]["parallelism"] == "dp2"
    assert r["dry_run"] is True and r["value"] is None


def test_single_rank_needs_no_launcher_dry():
    out, lines = _run(["--gpus", "1"], {"VT_BENCH_DRY": "1"})
    assert out.returncode == 0 and len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["config"]["grad_allreduce_floats"] == 0


def test_parent_passes_on_a_rank_failure():
    out, lines = _run(["--gpus", "2", "--shape", "no-such-shape"], {"VT_BENCH_DRY": "1"})
    assert out.returncode != 0 and not lines


@pytest.mark.gpu
def test_gpus_flag_trains_on_two_ranks_sharing_the_device():
    out, lines = _run(["--gpus", "2", "--steps", "4", "--warmup", "3", "--batch", "64", "--no-cpu-baseline"],
                      {"VT_BENCH_BACKEND": "gloo"})
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["grad_allreduce_floats"] == 33512 and r["config"]["distinct_devices"] == 1
    assert r["value"] > 0 and r["weights_finite_after_run"] is True
    assert abs(r["value"] - 2 * 64 * 4 / (r["ms_per_step"] * 4e-3)) / r["value"] < 1e-3  # both ranks' images over the max time
