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


def test_cpu_baseline_leg_of_the_secondary_configs(built_lib):
    """config 3 (shared map and the one-map-per-env variant) and config 5 (standing and collapsing regimes): the recipes build
    their terrain / gains / targets and the CPU leg steps them"""
    n = 16
    for recipe in (bench.Recipe(3, -1.0), bench.Recipe(3, -1.0, per_env_maps=True), bench.Recipe(5, -1.0), bench.Recipe(5, -1.0, "collapsing")):
        gc0, gv0 = recipe.initial_state(n, 0)
        if recipe.config == 3:
            maps, env_map = recipe.terrain(n, 0)
            assert maps.shape == ((n if recipe.per_env_maps else 1), 128, 128) and env_map.shape == (n,)
            assert np.abs(gc0[:, :2]).max() > 3.0 and np.abs(gc0[:, :2]).max() <= 6.0      # spread over the map, not +-0.1 m around its centre
            shard = recipe.initial_state(4, 8)                                             # rank-invariant: envs 8..11 as their own shard
            assert np.array_equal(shard[0], gc0[8:12])
        else:
            assert (recipe.kp[6:].max() == 3000.0) == (recipe.atlas_regime == "standing")
            assert np.array_equal(recipe.targets(4, 3, 8), recipe.targets(n, 3, 0)[8:12])
        out = bench.cpu_baseline(recipe, 0, True, 2.5, gc0.astype(np.float32), gv0.astype(np.float32),
                                 gc0.astype(np.float32).astype(np.float64), gv0, 0)
        assert out["value"] > 0 and f"{n} envs" in out["sample"]
