"""bench.py's N>1 launch plumbing on CPU (gloo): `python bench.py --gpus N` must start N ranks by itself, and the driver's
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` must produce the same single JSON line.

`--dry-run-ranks` replaces the device world by a host obs block per rank (no physics): what is under test is the
rendezvous, the env-shard bookkeeping, the per-step obs all-gather through raisimlib_amd.dist.ObsGatherer, the
barrier-bracketed MAX-over-ranks timing and that exactly ONE JSON line reaches stdout."""
import json
import os
import socket
import subprocess
import sys

import pytest

from common import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]


def _check(b, n, leg="second"):
    assert b["n_gpus"] == n and b["dry_run"] is True and b["scaling"] == "weak" and b["steps"] == 4 and b["warmup"] == 1
    # the two-leg order of an N > 1 run (bench.two_legs): the tested combination first, its numbers in `lockstep`; which leg `value` is
    assert b["value_leg"].startswith(leg) and b["lockstep"]["ran_first"] is True and b["lockstep"]["obs_all_gather"] == "in line"
    assert (b["pipelined_leg_error"] is None) == (leg == "second")
    # the communicator saw every rank
    assert b["rccl"]["rccl_ranks"] == n and b["rccl"]["allreduce_of_ones"] == float(n) and sorted(r["rank"] for r in b["rccl"]["by_rank"]) == list(range(n))
    assert len(b["ms_per_step_by_rank"]) == n
    assert abs(b["ms_per_step"] - max(b["ms_per_step_by_rank"])) < 1e-9          # the all-reduced MAX over ranks
    assert b["config"]["parallelism"] == f"env-shard x{n}"
    assert b["config"]["obs_all_gather"] != "none (1 rank)" and b["config"]["gathered_rows_correct"] is True
    assert abs(b["value"] - n * 32 * 4 * 4 / (b["ms_per_step"] * 4 * 1e-3)) < 1e-6 * b["value"]   # whole-job aggregate over all ranks


@pytest.mark.parametrize("n,extra", [(2, []), (3, ["--overlap-collective"])])
def test_self_spawn(n, extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--dry-run-ranks", "--steps", "4", "--warmup", "1", "--envs-per-gpu", "32", *extra],
                       capture_output=True, text=True, timeout=120, cwd="/tmp", env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout           # only rank 0 prints
    _check(lines[0], n)


@pytest.mark.parametrize("bad_rank", [0, 1])
def test_a_second_leg_that_raises_leaves_the_first_legs_line(bad_rank):
    """the leg that never ran on hardware (pipelined steps + gather on a side stream) must not be able to lose the run - it raises
    on one rank, every rank agrees to report the first leg (lock-step + in-line all-gather), the job exits 0 with ONE line that says so."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-ranks", "--steps", "4", "--warmup", "1", "--envs-per-gpu", "32", "--dry-run-fail-leg", str(bad_rank)],
                       capture_output=True, text=True, timeout=120, cwd="/tmp", env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    _check(lines[0], 2, leg="first")
    assert "injected failure" in lines[0]["pipelined_leg_error"] or "another rank" in lines[0]["pipelined_leg_error"]


def test_two_legs_unit():
    """bench.two_legs on its own: order, the guard, agreement, and that a single-leg run re-raises"""
    import bench
    calls = []
    r = bench.two_legs(lambda: calls.append("a") or {"x": 1}, lambda: calls.append("b") or {"x": 2}, lambda ok: ok)
    assert calls == ["a", "b"] and r["use"] == "second" and r["first"] == {"x": 1} and r["second"] == {"x": 2} and r["error"] is None

    def boom():
        raise ValueError("no")
    r = bench.two_legs(lambda: {"x": 1}, boom, lambda ok: ok)
    assert r["use"] == "first" and r["second"] is None and "ValueError: no" in r["error"]
    r = bench.two_legs(lambda: {"x": 1}, lambda: {"x": 2}, lambda ok: False)       # raised on ANOTHER rank
    assert r["use"] == "first" and r["second"] is None and "another rank" in r["error"]
    with pytest.raises(ValueError):
        bench.two_legs(None, boom, lambda ok: ok)
    assert bench.two_legs(None, lambda: {"x": 3}, lambda ok: ok)["use"] == "second"


def test_under_torch_distributed_run():
    """the driver's command line for N>1"""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_port()), BENCH, "--gpus", "2", "--dry-run-ranks", "--steps", "4", "--warmup", "1", "--envs-per-gpu", "32"],
                       capture_output=True, text=True, timeout=180, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    _check(lines[0], 2)


def test_more_ranks_than_gpus_fails_loudly():
    """without --dry-run-ranks the launcher refuses N > visible GPUs instead of running fewer ranks (this container has none)"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=120,
                       cwd="/tmp", env=env)
    assert r.returncode != 0 and "needs 2 visible GPUs" in r.stderr
    assert not _json_lines(r.stdout)


def test_a_failing_rank_fails_the_job():
    """every rank raises after the rendezvous (a negative shard size): the launcher returns non-zero and prints no JSON line"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-ranks", "--steps", "2", "--warmup", "1", "--envs-per-gpu", "-1"],
                       capture_output=True, text=True, timeout=120, cwd="/tmp", env=env)
    assert r.returncode != 0
    assert not _json_lines(r.stdout)
