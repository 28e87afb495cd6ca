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


def _check(b, n):
    assert b["n_gpus"] == n and b["dry_run"] is True and b["scaling"] == "weak" and b["steps"] == 4 and b["warmup"] == 1
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
