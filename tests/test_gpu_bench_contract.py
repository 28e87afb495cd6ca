"""bench.py end to end on the GPU with a short run: the one JSON line the driver parses carries every field of the contract."""
import json
import os
import subprocess
import sys

import pytest

from common import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config", [2, 3, 5])
def test_bench_prints_the_contract_line(built_lib, config):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--preroll", "8",
           "--config", str(config), "--cpu-seconds", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
    b = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["steps"] == 6 and b["warmup"] == 2 and b["higher_is_better"] is True and b["scaling"] == "weak"
    assert b["unit"] == "env-steps/s" and b["value"] > 1e6 and b["vs_baseline"] is None and b["data"] == "synthetic" and b["dtype"] == "f32"
    assert "workload" in b["config"] and "model" not in b["config"]
    assert abs(b["value"] - 4096 * 4 * 6 / (b["ms_per_step"] * 6 * 1e-3)) < 1e-3 * b["value"]      # whole-job env-steps over the timed region
    roof = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    # round 6: `value` is the RESIDENT leg (the bracket's control steps are one launch of the step kernel), the median of `repeats` fresh brackets; the
    # pipelined and the lock-step legs of the same line carry their own repeats
    assert b["value_leg"] == "resident" and b["resident_leg_error"] is None and b["resident"]["value"] == b["value"]
    # every step launch of the line's world ran a SPECIALISED code object of its kernel class (csrc/step_spec.h): the keys of bench.py's workloads are in
    # raisimlib_amd/spec_manifest.txt and build() compiled them; a key that is missing is compiled during warm-up (--specialization compile, the default)
    sp = b["specialization"]
    assert sp["mode"] == "compile" and sp["step_launches_generic"] == 0 and sp["step_launches_specialized"] > 7 * 6, sp
    for blk in (b, b["resident"], b["pipelined"], b["lockstep"]):
        assert blk["repeats"] == 7 and len(blk["value_repeats"]) == 7 and blk["value_min"] <= blk["value"] <= blk["value_max"]
        assert sorted(blk["value_repeats"])[3] == pytest.approx(blk["value"], rel=1e-9)
    assert roof["control_steps_per_launch"] == 6 and roof["kernel_launches_timed"] == 7
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["kernel_ms_mean"] * 1e-3) / 1e9) < 1e-6 * roof["achieved"]
    assert roof["algorithmic_bytes_per_launch"] == roof["algorithmic_bytes_per_env_step"] * 4096 * 4 * 6
    assert roof["kernel_ms_mean"] <= b["ms_per_step"] * 6 * 1.02          # the launch's own duration fits in the bracket it is the only launch of
    assert b["pipelined"]["value"] > 1e6
    roof = roof["control_step_launches"]            # the per-control-step launches of the pipelined leg, as round 5 reported them
    pipe_ms = b["pipelined"]["ms_per_step"]
    # consecutive control steps of that leg are pipelined: the same line also carries the lock-step number (same bracket, pipelining off) and
    # the roofline object says what its per-launch figures refer to
    assert b["config"]["step_pipelining"].startswith("on") and b["lockstep"]["value"] > 1e6 and b["lockstep"]["steps"] == 6
    ps = b["config"]["step_pipelining_stats"]
    assert ps["streams_overlap"] is True and ps["pipelined_launches"] >= 7 * 6 + 2 + 8      # timed + warm-up + pre-roll at least; the probe found a pair of streams that overlap
    assert abs(b["lockstep"]["value"] - 4096 * 4 * 6 / (b["lockstep"]["ms_per_step"] * 6 * 1e-3)) < 1e-3 * b["lockstep"]["value"]
    assert abs(roof["effective_ms_per_launch"] - pipe_ms) < 1e-9 and roof["launches_in_flight"] > 0.5
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["effective_ms_per_launch"] * 1e-3) / 1e9) < 1e-6 * roof["achieved"]
    assert abs(roof["achieved_over_one_launch_duration"] - roof["algorithmic_bytes_per_launch"] / (roof["kernel_ms_mean"] * 1e-3) / 1e9) < 1e-6 * roof["achieved"]
    cb = b["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["value"] > 0
    if config == 2:
        # the default single-GPU run of the headline configuration also carries configs 3 and 5 (same --steps / --warmup) and the
        # headline workload through the reference's own boundary, inside the same JSON object
        sec = b["secondary"]
        for name in ("config3", "config5"):
            c = sec[name]
            assert "error" not in c, c
            assert c["steps"] == 6 and c["warmup"] == 2 and c["value"] > 1e5 and c["lockstep_value"] > 1e5 and c["kernel_ms_mean"] > 0 and c["roofline"]["frac"] > 0
            assert c["cpu_baseline"]["value"] > 0 and abs(c["value"] - 4096 * 4 * 6 / (c["ms_per_step"] * 6 * 1e-3)) < 1e-3 * c["value"]
        tp = b["boundary_template_path"]
        assert "error" not in tp, tp
        assert tp["env_steps_per_s"] > 1e5 and tp["kernel_launches_per_control_step"] == 1.0
        assert len(tp["attempts_env_steps_per_s"]) == 3 and sorted(tp["attempts_env_steps_per_s"])[1] == tp["env_steps_per_s"]      # the median of three attempts
    else:
        assert "secondary" not in b


def test_bench_collective_path_on_one_rank(built_lib):
    """the obs all-gather (RCCL) of the N>1 step, forced onto a one-rank group: the same code path `--gpus N` runs per rank"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--preroll", "8",
           "--force-collective", "--no-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    b = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == 1 and b["config"]["obs_all_gather"].startswith("on a stream of its own behind each pipelined control step")
    assert len(b["ms_per_step_by_rank"]) == 1 and b["value"] > 1e6 and b["lockstep"]["value"] > 1e6
    # the resident leg's collective: ONE all-gather of the launch's [steps, N, obs] block behind the launch, and this rank's slice of the result is its own block
    rs = b["resident"]
    assert b["value_leg"] == "resident" and rs["gathered_rows_of_this_rank_correct"] is True and rs["obs_all_gather"]["block"] == [6, 4096, 49]


def test_bench_lockstep_flag(built_lib):
    """`--lockstep`: no pipelining anywhere, the all-gather in line on the launch stream (the line of rounds 1-3)"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--preroll", "8",
           "--force-collective", "--no-cpu", "--lockstep"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    b = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert b["config"]["step_pipelining"].startswith("off") and b["lockstep"] is None and "launches_in_flight" not in b["roofline"]
    assert b["config"]["obs_all_gather"] == "in line" and b["value"] > 1e6


def test_bench_peer_obs_exchange_on_one_rank(built_lib):
    """`--obs-exchange peer`: the rows reach the gathered block from the step kernel's epilogue (no collective); same line contract"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--preroll", "8",
           "--force-collective", "--obs-exchange", "peer", "--no-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    b = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == 1 and b["config"]["obs_all_gather"].startswith("peer-mapped") and b["value"] > 1e6


def test_bench_two_leg_order_of_an_n_gpu_run_on_one_rank(built_lib):
    """what `bench.py --gpus N` (N > 1) runs per rank - first the combination that has run before (lock-step control steps + the RCCL
    all-gather in line), then the pipelined steps + gather on a side stream under a guard, both in the one JSON line, the communicator's rank count
    and the ranks' devices with them - forced onto a one-rank group (RSB_BENCH_TWO_LEGS=1 --force-collective): the only way this code path meets a
    GPU before the driver's scaling run.  And the default single-GPU line's closed_loop block and roofline fractions."""
    env = dict(os.environ, RSB_BENCH_TWO_LEGS="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--preroll", "8", "--force-collective", "--no-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd="/tmp", env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    b = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert b["value_leg"] == "resident" and b["pipelined_leg_error"] is None and b["resident_leg_error"] is None and b["pipelined"]["leg"] == "pipelined"
    assert b["resident"]["gathered_rows_of_this_rank_correct"] is True
    ls = b["lockstep"]
    assert ls["ran_first"] is True and ls["obs_all_gather"] == "in line" and ls["gathered_rows_of_this_rank_correct"] is True and ls["value"] > 1e6
    assert b["value"] > 1e6                                          # (six steps are not a measurement: which leg is faster is bench.py's business)
    rc = b["rccl"]
    assert rc["rccl_ranks"] == 1 and rc["allreduce_of_ones"] == 1.0 and rc["distinct_devices"] == 1 and rc["by_rank"][0]["rank"] == 0 and rc["by_rank"][0]["compute_units"] > 0
    roof = b["roofline"]["control_step_launches"]
    assert abs(roof["frac"] - roof["frac_throughput"]) < 1e-12 and 0 < roof["frac_kernel_duration"] <= roof["frac_throughput"] * 1.05
    assert roof["timed_region_brackets"]["stride"] == 1 and roof["timed_region_brackets"]["n"] == 7 * 6


def test_bench_closed_loop_block(built_lib):
    """`closed_loop` of the default line (here on its own: --closed-loop-only): the policy-in-the-loop run pipelined and in lock-step from the same
    pre-rolled population - identical populations (the runs are bit-identical), the pipelined one faster, no pipeline fault"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--closed-loop-only", "--steps", "40", "--warmup", "5", "--preroll", "20"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    c = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])["closed_loop"]
    for blk, gain in ((c, 1.1), (c["mlp"], 0.9)):          # (40 steps: the MLP leg's 6 % are within a noisy box's reach; what is pinned is the block and the equal populations)          # the linear reference stage and the actor network (34-128-128-12) as the stage (whose 89 KB of weights per env block bound it: DESIGN.md 4.3)
        p, l, rs = blk["pipelined"], blk["lockstep"], blk["resident"]
        assert p["value"] > gain * l["value"] > 1e6 and rs["value"] > gain * l["value"]
        for k in ("resets_per_control_step_mean", "contacts_per_env", "solver_iters_mean", "base_height_mean"):
            assert p[k] == l[k] == rs[k], k
        for m in (p, l, rs):
            assert m["repeats"] == 7 and m["value_min"] <= m["value"] <= m["value_max"]
        assert rs["resident_launches"] >= 7
    assert c["pipeline"]["faults"] == 0 and c["pipeline"]["streams_overlap"] is True and c["pipeline"]["pipelined_launches"] >= 40


def test_bench_two_real_ranks_on_one_device(built_lib):
    """what one GPU allows of the first N > 1 run - TWO real processes, both on cuda:0, device tensors, torch.distributed gloo as the
    all-gather: leg 1 (lock-step + in-line gather), leg 2 (pipelined steps + gather on a side stream through rsb_step_pipeline_publish / _wait_event) and the
    resident leg (one gather of the launch's [steps, N, obs] block) with real kernels and real ranks.  Every leg's gathered block holds this rank's own rows
    in its slice, on BOTH ranks; the two ranks' shards differ (per-env seeds by global index)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--backend", "gloo", "--steps", "6", "--warmup", "2", "--preroll", "8",
           "--no-cpu", "--envs-per-gpu", "1024"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-3000:]
    b = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == 2 and b["rccl"]["backend"] == "gloo" and b["rccl"]["rccl_ranks"] == 2 and b["rccl"]["allreduce_of_ones"] == 2.0
    assert len(b["ms_per_step_by_rank"]) == 2
    assert b["pipelined_leg_error"] is None and b["resident_leg_error"] is None and b["value_leg"] in ("resident", "pipelined")      # (N > 1: the faster of the legs that succeeded everywhere)
    assert b["value"] == pytest.approx(max(b["resident"]["value"], b["pipelined"]["value"]), rel=1e-9)
    for leg in (b["lockstep"], b["pipelined"], b["resident"]):
        assert leg["gathered_rows_of_this_rank_correct"] is True and leg["gathered_rows_correct_on_all_ranks"] is True, leg
        assert leg["value"] > 0          # (gloo stages every device block through the host and the two ranks share one chip: not a measurement)
    assert b["lockstep"]["ran_first"] is True and b["resident"]["obs_all_gather"]["block"] == [6, 1024, 49]
