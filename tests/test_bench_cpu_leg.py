"""bench.py's CPU-baseline leg on a small sample (the leg is one of the three places allowed to use oracle/).  It only ever ran
on the GPU box, behind the GPU leg: a Python error in it would cost the driver's bench line.  Here it runs on 64 envs that include
fallen and self-colliding robots, so that its reset rule sees every kind of contact entry."""
import numpy as np

import bench
from common import standing_states


def test_cpu_baseline_leg_runs_and_reports(built_lib, monkeypatch):
    recipe = bench.Recipe(2, -1.0)
    n = 64
    gc0, gv0 = recipe.initial_state(n, 0)
    q0, u0 = standing_states(n, seed=3, z=(0.25, 0.55), vel=1.0)
    q0[:, 7:] += np.random.default_rng(3).uniform(-2.0, 2.0, (n, 12)) * (np.arange(n)[:, None] % 2)   # every other robot contorted
    out = bench.cpu_baseline(recipe, 0, True, 3.0, q0.astype(np.float32), u0.astype(np.float32),
                             gc0.astype(np.float32).astype(np.float64), gv0, 0)
    assert out["unit"] == "env-steps/s" and out["kind"] == "port" and out["value"] > 0 and out["cores"] >= 1
    assert "64 envs" in out["sample"]
    off = bench.cpu_baseline(recipe, 0, True, 3.0, q0.astype(np.float32), u0.astype(np.float32),
                             gc0.astype(np.float32).astype(np.float64), gv0, 0, self_collision=False)
    assert off["value"] > 0
