#!/usr/bin/env python3
"""Second golden fixture (round 5): the oracle frozen on the features the first fixture (flat-ground quadruped) does not touch.
Run from the repo root: python tests/golden/make_golden_features.py  ->  tests/golden/features_golden.npz

  hm_*    configs[2]'s recipe (bench.Recipe(3): quadrupeds on the shared 128 x 128 height map, closest-feature narrow phase with the height-field
          outer-side test of round 5): 32 envs after 30 oracle control steps, then ONE integrate() from the f32-rounded state, cold solver
  coul_*  the first fixture's 24 quadruped states under RSB_SLIP_COULOMB (orc_params::slip_rule = 1)
  atlas_* configs[4]'s recipe (bench.Recipe(5): Atlas-like humanoid, standing regime, multi-contact solver settings, Anderson step): 16 envs after
          20 oracle control steps, then ONE integrate()

Like the first fixture these pin the ORACLE against silent drift (tests/test_oracle_golden.py) and give the device a committed target that does
not move with the oracle's source (tests/test_gpu_parity.py); /root/reference holds no vectors."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402
from common import Oracle, f32  # noqa: E402
from raisimlib_amd import workload  # noqa: E402


def settle(recipe, n, steps):
    """the recipe's envs after `steps` oracle control steps (reset rule off: a fallen robot stays down and keeps its contacts)"""
    o = Oracle(recipe.model.blob)
    recipe.setup_oracle(o, n, 0)
    kp, kd = np.asarray(recipe.kp, np.float64), np.asarray(recipe.kd, np.float64)
    gc, gv = recipe.initial_state(n, 0)
    q, u = f32(gc), f32(gv)
    warm = o.new_warm_state(n)
    for k in range(steps):
        r = o.step_batch(q, u, workload.SUBSTEPS, kp, kd, f32(recipe.targets(n, k, 0)), np.zeros((n, recipe.model.nv)), lam_warm=warm)
        q, u = r["q"], r["u"]
    return f32(q), f32(u), f32(recipe.targets(n, steps, 0))


def one_step(recipe, q, u, pt, tweak=None):
    o = Oracle(recipe.model.blob)
    recipe.setup_oracle(o, len(q), 0)
    if tweak:
        tweak(o)
    kp, kd = np.asarray(recipe.kp, np.float64), np.asarray(recipe.kd, np.float64)
    r = o.step_batch(q, u, 1, kp, kd, pt, np.zeros((len(q), recipe.model.nv)), want_contacts=True, lam_warm=o.new_warm_state(len(q)))
    ids = np.full((len(q), o.p.kmax), -1, np.int32)
    for e in range(len(q)):
        ids[e, :r["n_contacts"][e]] = r["contacts"][e][:r["n_contacts"][e]]["collision"]
    return dict(q1=r["q"], u1=r["u"], n=r["n_contacts"], iters=r["iters"], flags=r["flags"], ids=ids)


def main():
    out = {}
    r3 = bench.Recipe(3, -1.0)
    q, u, pt = settle(r3, 32, 30)
    out.update({"hm_gc": q, "hm_gv": u, "hm_pt": pt}, **{"hm_" + k: v for k, v in one_step(r3, q, u, pt).items()})
    g = np.load(os.path.join(ROOT, "tests", "golden", "anymal_golden.npz"))
    r2 = bench.Recipe(2, -1.0)

    def coulomb(o):
        o.p.slip_rule = 1
    out.update({"coul_" + k: v for k, v in one_step(r2, g["gc"], g["gv"], g["pt"], coulomb).items()})
    r5 = bench.Recipe(5, -1.0)
    q, u, pt = settle(r5, 16, 20)
    out.update({"atlas_gc": q, "atlas_gv": u, "atlas_pt": pt}, **{"atlas_" + k: v for k, v in one_step(r5, q, u, pt).items()})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "features_golden.npz"), **out)
    print("wrote features_golden.npz: height map", int(out["hm_n"].sum()), "contacts in 32 envs; Coulomb", int(out["coul_n"].sum()), "in 24; Atlas-like",
          int(out["atlas_n"].sum()), "in 16; sweeps", out["hm_iters"].max(), out["coul_iters"].max(), out["atlas_iters"].max())


if __name__ == "__main__":
    main()
