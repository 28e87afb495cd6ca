"""Differential fuzz (GPU vs oracle, one integrate() each): random kinematic trees (branching below non-base bodies,
revolute and prismatic joints, rotated joint frames, spheres / capsules / boxes, joint limits), random states that
touch the ground, random gains and solver settings."""
import numpy as np
import pytest

from common import Oracle, f32
from raisimlib_amd import BatchedWorld, Model

pytestmark = pytest.mark.gpu


def random_urdf(rng, n_links):
    parts = ['<robot name="fuzz">']
    parents = [-1] + [int(rng.integers(0, i)) for i in range(1, n_links)]
    for i in range(n_links):
        m = float(rng.uniform(0.3, 3.0)) * (4.0 if i == 0 else 1.0)
        I = np.sort(rng.uniform(0.005, 0.05, 3)) * m
        com = rng.uniform(-0.05, 0.05, 3)
        parts.append(f'<link name="l{i}"><inertial><origin xyz="{com[0]:.4f} {com[1]:.4f} {com[2]:.4f}" rpy="{rng.uniform(-0.5, 0.5):.3f} {rng.uniform(-0.5, 0.5):.3f} 0"/>'
                     f'<mass value="{m:.4f}"/><inertia ixx="{I[0]:.5f}" ixy="0" ixz="0" iyy="{I[1]:.5f}" iyz="0" izz="{I[2]:.5f}"/></inertial>')
        kind = rng.integers(0, 4)
        p = rng.uniform(-0.1, 0.1, 3)
        if kind == 0 or i == 0:
            parts.append(f'<collision><origin xyz="{p[0]:.3f} {p[1]:.3f} {p[2]:.3f}"/><geometry><sphere radius="{rng.uniform(0.03, 0.09):.3f}"/></geometry></collision>')
        elif kind == 1:
            parts.append(f'<collision><origin xyz="{p[0]:.3f} {p[1]:.3f} {p[2]:.3f}" rpy="{rng.uniform(-1, 1):.3f} {rng.uniform(-1, 1):.3f} 0"/>'
                         f'<geometry><capsule radius="{rng.uniform(0.02, 0.05):.3f}" length="{rng.uniform(0.05, 0.2):.3f}"/></geometry></collision>')
        elif kind == 2 and n_links <= 6:
            parts.append(f'<collision><origin xyz="{p[0]:.3f} {p[1]:.3f} {p[2]:.3f}" rpy="0 0 {rng.uniform(-1, 1):.3f}"/>'
                         f'<geometry><box size="{rng.uniform(0.05, 0.2):.3f} {rng.uniform(0.05, 0.2):.3f} {rng.uniform(0.05, 0.1):.3f}"/></geometry></collision>')
        parts.append("</link>")
    for i in range(1, n_links):
        jt = "prismatic" if rng.random() < 0.25 else "revolute"
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        o = rng.uniform(-0.25, 0.25, 3)
        lo, hi = (-0.3, 0.3) if rng.random() < 0.3 else (-6.0, 6.0)
        parts.append(f'<joint name="j{i}" type="{jt}"><origin xyz="{o[0]:.3f} {o[1]:.3f} {o[2]:.3f}" rpy="{rng.uniform(-1, 1):.3f} {rng.uniform(-1, 1):.3f} {rng.uniform(-1, 1):.3f}"/>'
                     f'<parent link="l{parents[i]}"/><child link="l{i}"/><axis xyz="{ax[0]:.4f} {ax[1]:.4f} {ax[2]:.4f}"/>'
                     f'<limit effort="{0 if rng.random() < 0.5 else 30}" velocity="50" lower="{lo}" upper="{hi}"/>'
                     f'<dynamics damping="{rng.uniform(0, 0.05):.3f}"/></joint>')
    parts.append("</robot>")
    return "\n".join(parts)


@pytest.mark.parametrize("seed", range(12))
def test_random_tree_one_step_parity(built_lib, seed):
    rng = np.random.default_rng(1000 + seed)
    n_links = int(rng.integers(2, 11))
    model = Model(urdf_string=random_urdf(rng, n_links))
    nq, nv, N = model.nq, model.nv, 128
    kmax = 16 if model.ncol > 8 else 8
    gc = np.zeros((N, nq)); gc[:, 0:2] = rng.uniform(-1, 1, (N, 2)); gc[:, 2] = rng.uniform(0.0, 0.5, N)
    qq = rng.normal(size=(N, 4)); gc[:, 3:7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
    gc[:, 7:] = rng.uniform(-0.5, 0.5, (N, nq - 7))
    gv = rng.normal(size=(N, nv)) * 1.0
    kp = np.zeros(nv, np.float32); kd = np.zeros(nv, np.float32)
    kp[6:] = rng.uniform(0, 60, nv - 6); kd[6:] = rng.uniform(0, 1.0, nv - 6)
    pt = gc.copy(); pt[:, 7:] += rng.uniform(-0.3, 0.3, (N, nq - 7))
    w = BatchedWorld(model, N); w.set_max_contacts(kmax)
    o = Oracle(model.blob); o.p.kmax = kmax
    mu = float(rng.choice([0.3, 0.8, 1.2])); w.set_default_friction(mu); o.p.mu = mu
    dtg = np.zeros((N, nv))
    w.set_pd_gains(kp, kd); w.set_pd_target(pt, dtg); w.set_state(gc, gv)
    w.integrate(1)
    q1, u1 = w.get_state(); cnt, _ = w.get_contacts(); fl = w.get_flags()
    ref = o.step_batch(f32(gc), f32(gv), 1, kp.astype(np.float64), kd.astype(np.float64), f32(pt), dtg)
    w.close()
    assert np.array_equal(cnt, ref["n_contacts"]), (seed, n_links)
    conv = ((ref["flags"] | fl) & 5) == 0
    assert conv.mean() > 0.6, (seed, conv.mean())
    eu = np.abs(u1 - ref["u"]).max(axis=1) / (1 + np.abs(ref["u"]).max(axis=1))
    eq = np.abs(q1 - ref["q"]).max(axis=1)
    assert np.isfinite(q1).all() and np.isfinite(u1).all()
    assert eu[conv].max() < 2e-3 and np.median(eu) < 2e-5 and eq[conv].max() < 2e-5, (seed, n_links, eu[conv].max(), eq[conv].max())


@pytest.mark.parametrize("seed", range(6))
def test_random_fixed_base_tree_parity(built_lib, seed):
    """The same generator with the root link named "world" (a fixed-base system: the base rows are inert), joints anywhere in
    +-1.5 rad so that links also hit each other (self-collision), a ground plane some of them reach; two sub-steps."""
    rng = np.random.default_rng(4000 + seed)
    n_links = int(rng.integers(3, 9))
    urdf = random_urdf(rng, n_links).replace('"l0"', '"world"')
    model = Model(urdf_string=urdf)
    assert model.blob.fixed_base == 1
    nq, nv, N = model.nq, model.nv, 128
    gc = np.zeros((N, nq)); gc[:, 3] = 1.0
    gc[:, 7:] = rng.uniform(-1.5, 1.5, (N, nq - 7))
    gv = np.zeros((N, nv)); gv[:, 6:] = rng.normal(size=(N, nv - 6))
    kp = np.zeros(nv, np.float32); kd = np.zeros(nv, np.float32)
    kp[6:] = rng.uniform(0, 60, nv - 6); kd[6:] = rng.uniform(0, 1.0, nv - 6)
    pt = gc.copy(); pt[:, 7:] += rng.uniform(-0.3, 0.3, (N, nq - 7))
    w = BatchedWorld(model, N); w.set_max_contacts(16); w.add_ground(-0.25)
    o = Oracle(model.blob); o.p.kmax = 16; o.set_ground(-0.25)
    dtg = np.zeros((N, nv))
    w.set_pd_gains(kp, kd); w.set_pd_target(pt, dtg); w.set_state(gc, gv)
    w.integrate(2)
    q1, u1 = w.get_state(); cnt, _ = w.get_contacts(); fl = w.get_flags()
    ref = o.step_batch(f32(gc), f32(gv), 2, kp.astype(np.float64), kd.astype(np.float64), f32(pt), dtg, lam_warm=o.new_warm_state(N))
    w.close()
    assert np.array_equal(q1[:, :7], np.tile([0, 0, 0, 1, 0, 0, 0], (N, 1))) and not u1[:, :6].any()
    conv = ((ref["flags"] | fl) & 5) == 0
    same = cnt == ref["n_contacts"]
    assert same.mean() > 0.97, (seed, same.mean())                 # (a contact that appears in the second sub-step sits at a threshold)
    ok = conv & same
    assert ok.mean() > 0.3, (seed, ok.mean())                       # (random poses start jammed into the ground: many solves stagnate)
    eu = np.abs(u1 - ref["u"]).max(axis=1) / (1 + np.abs(ref["u"]).max(axis=1))
    eq = np.abs(q1 - ref["q"]).max(axis=1)
    assert np.isfinite(q1).all() and np.isfinite(u1).all()
    assert np.percentile(eu[ok], 99) < 2e-3 and np.median(eu[ok]) < 2e-5 and np.percentile(eq[ok], 99) < 2e-5, (seed, n_links, eu[ok].max(), eq[ok].max())


@pytest.mark.parametrize("seed", range(6))
def test_random_tree_three_substeps_on_a_height_map(built_lib, seed):
    """Same generator, three fused sub-steps (warm state in use from the second one) on a random Perlin terrain."""
    from raisimlib_amd.world import heightmap_perlin
    rng = np.random.default_rng(2000 + seed)
    model = Model(urdf_string=random_urdf(rng, int(rng.integers(3, 9))))
    nq, nv, N = model.nq, model.nv, 96
    kmax = 16 if model.ncol > 8 else 8
    H = heightmap_perlin(48, 48, 4.8, 4.8, frequency=0.6, z_scale=0.2, seed=int(rng.integers(1, 1000)))
    hm = (48, 48, 4.8, 4.8, 0.0, 0.0, H)
    gc = np.zeros((N, nq)); gc[:, 0:2] = rng.uniform(-1.5, 1.5, (N, 2)); gc[:, 2] = rng.uniform(0.0, 0.5, N)
    qq = rng.normal(size=(N, 4)); gc[:, 3:7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
    gc[:, 7:] = rng.uniform(-0.4, 0.4, (N, nq - 7))
    gv = rng.normal(size=(N, nv)) * 0.5
    kp = np.zeros(nv, np.float32); kd = np.zeros(nv, np.float32); kp[6:] = 30.0; kd[6:] = 0.5
    w = BatchedWorld(model, N); w.set_max_contacts(kmax); w.add_height_map(*hm)
    o = Oracle(model.blob); o.p.kmax = kmax; o.set_heightmap(*hm)
    dtg = np.zeros((N, nv))
    w.set_pd_gains(kp, kd); w.set_pd_target(gc, dtg); w.set_state(gc, gv)
    w.integrate(3)
    q1, u1 = w.get_state(); fl = w.get_flags()
    ref = o.step_batch(f32(gc), f32(gv), 3, kp.astype(np.float64), kd.astype(np.float64), f32(gc), dtg, lam_warm=o.new_warm_state(N))
    w.close()
    conv = ((ref["flags"] | fl) & 5) == 0
    eu = np.abs(u1 - ref["u"]).max(axis=1) / (1 + np.abs(ref["u"]).max(axis=1))
    assert np.isfinite(q1).all() and conv.mean() > 0.5
    assert np.median(eu) < 5e-5 and np.percentile(eu[conv], 95) < 5e-3, (seed, np.median(eu), np.percentile(eu[conv], 95))
