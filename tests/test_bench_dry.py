"""bench.py's control flow without a GPU (CODA_BENCH_DRY=1: a toy CPU module takes the detector's place,
every branch of the measurement loop is the real one): the process terminates, rank 0 prints exactly one
JSON line with the driver's contract fields, and with two gloo ranks every rank runs the same sequence of
collectives (a rank-0-only step after the timed region once deadlocked the DDP all-reduce)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def _run(cmd, port=None):
    env = dict(os.environ, CODA_BENCH_DRY="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env["MASTER_ADDR"] = "127.0.0.1"
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=480)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _check(out, world, steps, warmup):
    for k in CONTRACT:
        assert k in out, k
    assert out["n_gpus"] == world and out["steps"] == steps and out["warmup"] == warmup
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert abs(out["value"] - world * out["config"]["scenes_per_gpu"] / (out["ms_per_step"] * 1e-3)) < 0.01 * out["value"]
    assert out["config"]["parallelism"].startswith(f"dp{world}")
    for k in ["bound", "achieved", "peak", "unit", "frac", "traffic"]:
        assert k in out["roofline"], k


def test_single_process_flow():
    out = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1"])
    _check(out, 1, 3, 1)
    assert "cpu_baseline" not in out  # the dry run measures nothing, least of all the oracle
    # the unchanged caller's leg (engine.py:136-164 with its blocking finite check) is reported at top level
    assert out["value_unchanged"] > 0 and out["ms_per_step_unchanged"] > 0
    assert "loss.item()" in out["value_unchanged_caller"]["what"]
    assert out["comm"]["rccl_ranks"] == 1  # always present: a one-rank line cannot pass for an N-GPU line


@pytest.mark.timeout(300)
def test_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the driver's scaling command may be exactly that):
    bench.py starts the two ranks itself, like the reference's main.py:1103-1108."""
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    _check(out, 2, 3, 1)
    _check_multi(out, 2)
    assert "bench.py itself" in out["comm"]["launched_by"]


def test_rank_count_mismatch_is_fatal():
    """a launcher that started another number of ranks than --gpus says: no line, non-zero exit"""
    env = dict(os.environ, CODA_BENCH_DRY="1", OMP_NUM_THREADS="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and "refusing to measure" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    env.pop("WORLD_SIZE"), env.pop("RANK"), env.pop("LOCAL_RANK")
    env["WORLD_SIZE"] = "2"
    env["RANK"] = "0"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and "refusing to measure" in r.stderr


def test_a_failing_rank_takes_the_job_down():
    """one rank dying must not leave the others parked in a collective: the launcher stops them, exits non-zero"""
    env = dict(os.environ, CODA_BENCH_DRY="1", OMP_NUM_THREADS="1", CODA_BENCH_DRY_FAIL_RANK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and "rank 1 of 2 exited" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.timeout(300)
def test_two_ranks_run_the_same_collectives():
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", "29541", "bench.py", "--gpus", "2", "--steps", "3",
                "--warmup", "1"])
    _check(out, 2, 3, 1)
    _check_multi(out, 2)


def _check_multi(out, world):
    """what the first real scaling run needs to be self-diagnosing: the collectives of a step timed alone, and the
    reference's unchanged wrap (SyncBatchNorm + DistributedDataParallel + the per-step loss all-reduce and
    .item()) timed beside this package's gradient synchronisation"""
    assert out["comm"]["rccl_ranks"] == world and out["comm"]["backend"] == "gloo"
    assert all(v > 0 for v in out["comm"]["allreduce_ms"].values()) and out["comm"]["syncbn_ms"] > 0
    assert out["value_unchanged"] > 0
    assert "DistributedDataParallel" in out["value_unchanged_caller"]["what"]


@pytest.mark.timeout(600)
def test_eight_ranks_like_the_scaling_run():
    """the driver's 8-GPU command line, on 8 CPU processes"""
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                "--master-addr", "127.0.0.1", "--master-port", "29547", "bench.py", "--gpus", "8", "--steps", "2",
                "--warmup", "1"])
    _check(out, 8, 2, 1)
    _check_multi(out, 8)
