"""Pins the oracle from SECOND FORMULATIONS that share no code and no derivation with oracle/rsb_oracle.c (VERDICT r03, weak #1 /
next-round #5: the CRBA / RNEA / ABA cross-checks of tests/test_oracle_crosschecks.py all run through one kinematics() and one
common-frame formulation, so a shared design error would be invisible).  Parity with RaiSim stays unpinned (no reference source);
what this file adds is that the oracle's rigid-body dynamics and its one-contact rule are checked against independently derived
statements of the same physics:

  (i)   a SYMBOLIC projected Newton-Euler (Kane) model of a floating-base chain with two revolute links, built with sympy from
        positions only: velocities, bias accelerations and Jacobians come from symbolic differentiation (dR/dt = [w]x R, dp/dt = v),
        M = sum J_v^T m J_v + J_w^T I J_w, h = sum J_v^T m (a_bias - g) + J_w^T (I alpha_bias + w x I w).  No spatial algebra, no
        recursion.  Compared with the oracle's CRBA / RNEA (and, on a GPU, with the device query kernel: tests/test_gpu_parity.py
        already pins device == oracle for M and h).
  (ii)  a numpy BODY-FRAME Pluecker implementation of RNEA (RBDA Table 5.1), CRBA (Table 6.2) and ABA (Table 7.1) written from the
        book's tables with 6 x 6 transforms X = [E 0; -E rx E] (the oracle works in ONE world-aligned frame with its origin at the base
        and never forms an X), on the ANYmal-like and the Atlas-like model.
  (iii) the one-contact rule written twice more, textbook style, on random Delassus blocks and on the quadruped's own foot blocks:
        (A) Hwangbo, Lee, Hutter 2018 as a plain BISECTION on the slip angle - the root of dE/dtheta along the curve {v_n+ = 0} x
            {cone boundary}, E = contact-space kinetic energy - from a brute-force bracket: must agree with the oracle's accelerated
            search (16 directions + 16-section + Newton polish), and does;
        (B) the CLASSICAL Coulomb statement (slip velocity anti-parallel to the friction impulse: Stewart-Trinkle, Anitescu-Potra),
            also by bisection: identical to (A) where the normal row does not couple with the tangential ones, different where it
            does - the per-contact method's maximum-dissipation principle is a statement about the whole contact-space energy, not
            about the tangent plane.  HOW different is measured and reported, not assumed.
"""
import numpy as np
import pytest

from common import Oracle
from raisimlib_amd import Model, rsc_path

CHAIN_URDF = """<?xml version="1.0"?>
<robot name="chain">
  <link name="trunk">
    <inertial><origin xyz="0.03 -0.02 0.05"/><mass value="7.0"/>
      <inertia ixx="0.21" ixy="0.013" ixz="-0.02" iyy="0.34" iyz="0.017" izz="0.27"/></inertial>
  </link>
  <link name="upper">
    <inertial><origin xyz="0.02 0.11 -0.17"/><mass value="2.3"/>
      <inertia ixx="0.031" ixy="0.002" ixz="0.004" iyy="0.027" iyz="-0.003" izz="0.012"/></inertial>
  </link>
  <link name="lower">
    <inertial><origin xyz="-0.04 0.01 -0.12"/><mass value="0.9"/>
      <inertia ixx="0.011" ixy="-0.001" ixz="0.0007" iyy="0.012" iyz="0.0011" izz="0.004"/></inertial>
  </link>
  <joint name="hip" type="revolute">
    <origin xyz="0.27 0.12 -0.03" rpy="0.3 -0.2 0.5"/><parent link="trunk"/><child link="upper"/><axis xyz="0.6 0.0 0.8"/>
    <limit effort="0" velocity="100" lower="-10" upper="10"/>
  </joint>
  <joint name="knee" type="revolute">
    <origin xyz="0.05 0.02 -0.31" rpy="-0.4 0.1 0.2"/><parent link="upper"/><child link="lower"/><axis xyz="0.0 1.0 0.0"/>
    <limit effort="0" velocity="100" lower="-10" upper="10"/>
  </joint>
</robot>
"""


def quat_to_rot(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rand_state(model, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    q = np.zeros(model.nq)
    q[:3] = rng.uniform(-1, 1, 3)
    q[3:7] = rng.normal(size=4)
    q[3:7] /= np.linalg.norm(q[3:7])
    q[7:] = rng.uniform(-1.0, 1.0, model.nq - 7)
    u = rng.normal(size=model.nv) * scale
    return q, u


# ------------------------------------------------------------------------------------------------- (i) symbolic projected Newton-Euler
def _symbolic_chain(blob, gravity):
    """(M(x), h(x)) as numpy functions of x = [p (3), R (9, row-major), qj (nb - 1), u (nv)] for the floating-base tree in `blob`,
    derived with sympy from POSITIONS: every velocity / acceleration is a symbolic time derivative."""
    import sympy as sp
    nb, nv = blob.nb, blob.nv
    p = sp.Matrix(sp.symbols("p0:3"))
    Rs = sp.Matrix(3, 3, sp.symbols("r0:9"))
    qj = list(sp.symbols(f"q1:{nb}"))
    u = sp.Matrix(sp.symbols(f"u0:{nv}"))
    vw, ww = u[0:3, 0], u[3:6, 0]

    def skew(a):
        return sp.Matrix([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])

    # time derivative of an expression in (p, R, qj, u) with du/dt = 0: the chain rule over the primitive symbols
    prim = list(p) + list(Rs) + qj
    rate = list(vw) + list(skew(ww) * Rs) + [u[5 + i] for i in range(1, nb)]

    def ddt(expr):
        return sp.Matrix(expr).applyfunc(lambda e: sum(sp.diff(e, s) * r for s, r in zip(prim, rate)))

    # poses of the bodies (world frame) from the blob's tree: child frame = parent * (ptree, rtree * Rot(axis, q))
    Rw, pw = [Rs], [p]
    for i in range(1, nb):
        par = blob.parent[i]
        ax = sp.Matrix([sp.Float(blob.axis[i][k]) for k in range(3)])
        Rt = sp.Matrix(3, 3, [sp.Float(blob.rtree[i][k]) for k in range(9)])
        pt = sp.Matrix([sp.Float(blob.ptree[i][k]) for k in range(3)])
        assert blob.jtype[i] == 1, "this symbolic model is written for revolute joints"
        c, s_ = sp.cos(qj[i - 1]), sp.sin(qj[i - 1])
        Rq = c * sp.eye(3) + (1 - c) * (ax * ax.T) + s_ * skew(ax)            # Rodrigues
        Rw.append(Rw[par] * Rt * Rq)
        pw.append(pw[par] + Rw[par] * pt)
    g = sp.Matrix([sp.Float(x) for x in gravity])
    M = sp.zeros(nv, nv)
    h = sp.zeros(nv, 1)
    for i in range(nb):
        m = sp.Float(blob.mass[i])
        com = sp.Matrix([sp.Float(blob.com[i][k]) for k in range(3)])
        ii = [sp.Float(x) for x in blob.inertia[i]]
        Ic = sp.Matrix([[ii[0], ii[1], ii[2]], [ii[1], ii[3], ii[4]], [ii[2], ii[4], ii[5]]])
        c_w = pw[i] + Rw[i] * com
        v_c = ddt(c_w)                                               # velocity of the centre of mass
        Wx = ddt(Rw[i]) * Rw[i].T                                    # [w_i]x = dR/dt R^T
        w_i = sp.Matrix([Wx[2, 1], Wx[0, 2], Wx[1, 0]])
        a_b, al_b = ddt(v_c), ddt(w_i)                               # bias accelerations (du/dt = 0)
        Jv, Jw = v_c.jacobian(u), w_i.jacobian(u)
        Iw = Rw[i] * Ic * Rw[i].T
        M += m * Jv.T * Jv + Jw.T * Iw * Jw
        h += Jv.T * (m * (a_b - g)) + Jw.T * (Iw * al_b + w_i.cross(Iw * w_i))
    syms = list(p) + list(Rs) + qj + list(u)
    fM = sp.lambdify(syms, M, modules="numpy", cse=True)
    fh = sp.lambdify(syms, h, modules="numpy", cse=True)
    return fM, fh


def test_symbolic_projected_newton_euler_matches_crba_and_rnea(built_lib):
    sp = pytest.importorskip("sympy")      # noqa: F841  (part of this image; the check is skipped, not faked, without it)
    m = Model(urdf_string=CHAIN_URDF)
    assert (m.nb, m.nq, m.nv) == (3, 9, 8)
    o = Oracle(m.blob)
    grav = [o.p.gravity[k] for k in range(3)]
    fM, fh = _symbolic_chain(m.blob, grav)
    worst_M = worst_h = 0.0
    for seed in range(12):
        q, u = rand_state(m, seed, scale=2.0)
        R = quat_to_rot(q[3:7])
        x = list(q[:3]) + list(R.reshape(9)) + list(q[7:]) + list(u)
        Ms, hs = np.asarray(fM(*x), float), np.asarray(fh(*x), float).reshape(-1)
        Mo, ho = o.mass_matrix(q), o.nonlinearities(q, u)
        arm = np.array([m.blob.armature[i] for i in range(1, m.nb)])
        Ms[6:, 6:] += np.diag(arm)                                   # rotor inertia is added to the diagonal (rsb.h)
        assert np.allclose(Ms, Ms.T, atol=1e-12)
        worst_M = max(worst_M, np.abs(Ms - Mo).max() / np.abs(Mo).max())
        worst_h = max(worst_h, np.abs(hs - ho).max() / (np.abs(ho).max() + 1e-9))
    print(f"symbolic projected Newton-Euler vs oracle: max rel |dM| {worst_M:.1e}, max rel |dh| {worst_h:.1e} over 12 random states")
    assert worst_M < 1e-11 and worst_h < 1e-10


# ------------------------------------------------------------------------------------------------- (ii) body-frame Pluecker algorithms
def _skew(a):
    return np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])


def _xform(E, r):
    """Pluecker motion transform from frame A to frame B: B's axes are E (B <- A), B's origin sits at r in A (RBDA eq. 2.24)."""
    X = np.zeros((6, 6))
    X[:3, :3] = E
    X[3:, 3:] = E
    X[3:, :3] = -E @ _skew(r)
    return X


def _crm(v):
    X = np.zeros((6, 6))
    X[:3, :3] = _skew(v[:3]); X[3:, 3:] = _skew(v[:3]); X[3:, :3] = _skew(v[3:])
    return X


def _rot_axis(a, q):
    a = np.asarray(a, float)
    return np.cos(q) * np.eye(3) + (1 - np.cos(q)) * np.outer(a, a) + np.sin(q) * _skew(a)


class PlueckerModel:
    """RBDA's model data (lambda, X_tree, S, I) from the blob; floating base = body 0 with the 6-DoF "joint" RaiSim's generalized
    velocity describes: (world-frame linear velocity of the base origin, world-frame angular velocity)."""

    def __init__(self, blob):
        self.b = blob
        self.nb, self.nv = blob.nb, blob.nv
        self.I = []
        for i in range(self.nb):
            m, c = blob.mass[i], np.array(blob.com[i][:])
            ii = blob.inertia[i]
            Ic = np.array([[ii[0], ii[1], ii[2]], [ii[1], ii[3], ii[4]], [ii[2], ii[4], ii[5]]])
            cx = _skew(c)
            I6 = np.zeros((6, 6))
            I6[:3, :3] = Ic + m * cx @ cx.T; I6[:3, 3:] = m * cx; I6[3:, :3] = m * cx.T; I6[3:, 3:] = m * np.eye(3)   # RBDA eq. 2.63
            self.I.append(I6)

    def joint(self, i, q):
        """(X_up = X_J X_tree: parent -> child, S) of joint i at the joint coordinate q (RBDA 4.4)"""
        b = self.b
        ax = np.array(b.axis[i][:]); Rt = np.array(b.rtree[i][:]).reshape(3, 3); pt = np.array(b.ptree[i][:])
        if b.jtype[i] == 1:
            E = (Rt @ _rot_axis(ax, q)).T
            return _xform(E, pt), np.r_[ax, 0, 0, 0]
        E = Rt.T
        return _xform(E, pt + Rt @ ax * q), np.r_[0, 0, 0, ax]

    def base(self, q, u, udot, gravity):
        """base spatial velocity / acceleration in BASE coordinates from RaiSim's world-frame base rates; gravity as a fictitious
        acceleration (RBDA 5.3: a_0 = -a_g)"""
        E = quat_to_rot(q[3:7]).T
        v0 = np.r_[E @ u[3:6], E @ u[0:3]]
        # spatial linear acceleration = classical acceleration of the origin - w x v (RBDA eq. 2.48)
        a0 = np.r_[E @ udot[3:6], E @ (udot[0:3] - np.cross(u[3:6], u[0:3]) - np.asarray(gravity))]
        return E, v0, a0

    def rnea(self, q, u, udot, gravity):
        """tau = ID(q, u, udot), RBDA Table 5.1 in body coordinates"""
        nb = self.nb
        E, v0, a0 = self.base(q, u, udot, gravity)
        v, a, f, Xup, S = [v0], [a0], [None] * nb, [None] * nb, [None] * nb
        f[0] = self.I[0] @ a0 + _crm(v0).T @ -(-(self.I[0] @ v0))            # v x* I v = -crm(v)^T (I v)
        f[0] = self.I[0] @ a0 - _crm(v0).T @ (self.I[0] @ v0)
        for i in range(1, nb):
            lam = self.b.parent[i]
            Xup[i], S[i] = self.joint(i, q[6 + i])
            vJ = S[i] * u[5 + i]
            v.append(Xup[i] @ v[lam] + vJ)
            a.append(Xup[i] @ a[lam] + S[i] * udot[5 + i] + _crm(v[i]) @ vJ)
            f[i] = self.I[i] @ a[i] - _crm(v[i]).T @ (self.I[i] @ v[i])
        tau = np.zeros(self.nv)
        for i in range(nb - 1, 0, -1):
            tau[5 + i] = S[i] @ f[i] + self.b.armature[i] * udot[5 + i]
            f[self.b.parent[i]] = f[self.b.parent[i]] + Xup[i].T @ f[i]
        R = E.T
        tau[0:3] = R @ f[0][3:]           # the base's generalized force in RaiSim's coordinates: world-frame force and moment
        tau[3:6] = R @ f[0][:3]
        return tau

    def crba(self, q):
        """M(q), RBDA Table 6.2 (composite rigid bodies in body coordinates), base block mapped to RaiSim's base coordinates"""
        nb, nv = self.nb, self.nv
        Xup, S = [None] * nb, [None] * nb
        Ic = [I.copy() for I in self.I]
        for i in range(1, nb):
            Xup[i], S[i] = self.joint(i, q[6 + i])
        for i in range(nb - 1, 0, -1):
            Ic[self.b.parent[i]] = Ic[self.b.parent[i]] + Xup[i].T @ Ic[i] @ Xup[i]
        E = quat_to_rot(q[3:7]).T
        T = np.zeros((6, 6))          # base twist [w_b; v_b] = T [v_w; w_w]
        T[:3, 3:] = E; T[3:, :3] = E
        M = np.zeros((nv, nv))
        M[:6, :6] = T.T @ Ic[0] @ T
        for i in range(1, nb):
            F = Ic[i] @ S[i]
            M[5 + i, 5 + i] = S[i] @ F + self.b.armature[i]
            j = i
            while self.b.parent[j] > 0:
                F = Xup[j].T @ F
                j = self.b.parent[j]
                M[5 + i, 5 + j] = M[5 + j, 5 + i] = F @ S[j]
            F = Xup[j].T @ F          # into the base's coordinates
            M[5 + i, :6] = M[:6, 5 + i] = T.T @ F
        return M

    def aba(self, q, u, tau, gravity):
        """udot = FD(q, u, tau), RBDA Table 7.1 with a floating base (9.4): articulated-body inertias in body coordinates"""
        nb = self.nb
        E = quat_to_rot(q[3:7]).T
        R = E.T
        v0 = np.r_[E @ u[3:6], E @ u[0:3]]
        v, c, Xup, S = [v0], [np.zeros(6)], [None] * nb, [None] * nb
        IA = [I.copy() for I in self.I]
        pA = [-_crm(v0).T @ (self.I[0] @ v0)]
        for i in range(1, nb):
            lam = self.b.parent[i]
            Xup[i], S[i] = self.joint(i, q[6 + i])
            vJ = S[i] * u[5 + i]
            v.append(Xup[i] @ v[lam] + vJ)
            c.append(_crm(v[i]) @ vJ)
            pA.append(-_crm(v[i]).T @ (self.I[i] @ v[i]))
        # external generalized force on the base (RaiSim coordinates -> a spatial force in base coordinates)
        pA[0] = pA[0] - np.r_[E @ tau[3:6], E @ tau[0:3]]
        U, d, uu = [None] * nb, [None] * nb, [None] * nb
        for i in range(nb - 1, 0, -1):
            U[i] = IA[i] @ S[i]
            d[i] = S[i] @ U[i] + self.b.armature[i]
            uu[i] = tau[5 + i] - S[i] @ pA[i]
            Ia = IA[i] - np.outer(U[i], U[i]) / d[i]
            pa = pA[i] + Ia @ c[i] + U[i] * uu[i] / d[i]
            lam = self.b.parent[i]
            IA[lam] = IA[lam] + Xup[i].T @ Ia @ Xup[i]
            pA[lam] = pA[lam] + Xup[i].T @ pa
        a = [None] * nb
        a[0] = np.linalg.solve(IA[0], -pA[0])                       # spatial acceleration of the base, gravity not yet in
        ag = np.r_[0, 0, 0, E @ np.asarray(gravity)]
        udot = np.zeros(self.nv)
        a0_true = a[0]                                               # (gravity enters as a_0 = -a_g: add it back when reading classical rates)
        # with gravity as a fictitious acceleration the recursion runs on a' = a - a_g; solve the base in those terms
        a[0] = np.linalg.solve(IA[0], -(pA[0] - IA[0] @ ag)) - ag
        for i in range(1, nb):
            ap = Xup[i] @ a[self.b.parent[i]] + c[i]
            udot[5 + i] = (uu[i] - U[i] @ ap) / d[i]
            a[i] = ap + S[i] * udot[5 + i]
        a0 = a[0] + ag                                               # true spatial acceleration of the base
        udot[3:6] = R @ a0[:3]
        udot[0:3] = R @ a0[3:] + np.cross(u[3:6], u[0:3])           # classical acceleration of the origin (RBDA eq. 2.48)
        del a0_true
        return udot


@pytest.mark.parametrize("which", ["anymal", "atlas"])
def test_body_frame_pluecker_rnea_crba_aba_match_the_common_frame_oracle(which, request):
    m = request.getfixturevalue(which)
    o = Oracle(m.blob)
    pm = PlueckerModel(m.blob)
    g = [o.p.gravity[k] for k in range(3)]
    rng = np.random.default_rng(5)
    eM = eh = eid = ea = 0.0
    for seed in range(6):
        q, u = rand_state(m, 100 + seed, scale=1.5)
        udot = rng.normal(size=m.nv)
        tau = rng.normal(size=m.nv) * 5.0
        M, Mo = pm.crba(q), o.mass_matrix(q)
        eM = max(eM, np.abs(M - Mo).max() / np.abs(Mo).max())
        h, ho = pm.rnea(q, u, np.zeros(m.nv), g), o.nonlinearities(q, u)
        eh = max(eh, np.abs(h - ho).max() / (np.abs(ho).max() + 1e-9))
        t, to = pm.rnea(q, u, udot, g), o.inverse_dynamics(q, u, udot)
        eid = max(eid, np.abs(t - to).max() / (np.abs(to).max() + 1e-9))
        a, ao = pm.aba(q, u, tau, g), o.aba(q, u, tau)
        ea = max(ea, np.abs(a - ao).max() / (np.abs(ao).max() + 1e-9))
        assert np.allclose(M @ a + h, tau, rtol=1e-8, atol=1e-8 * np.abs(tau).max() * np.linalg.cond(M))   # the three agree among themselves
    print(f"{which}: body-frame Pluecker vs oracle, max relative error: M {eM:.1e}, h {eh:.1e}, inverse dynamics {eid:.1e}, ABA {ea:.1e}")
    assert eM < 1e-11 and eh < 1e-9 and eid < 1e-9
    assert ea < 1e-7            # (forward dynamics amplifies by cond(M): 1e3 on the quadruped, 4e5 on the humanoid)


# ------------------------------------------------------------------------------------------------- (iii) the published one-contact rule
def bisection_contact(G, v, mu, form, iters=200):
    """Open / stick / slip for ONE contact, textbook style: G 3x3 Delassus block in the contact frame [t1 t2 n], v the contact
    velocity without this contact's impulse.  Slip: on the curve {v_n+ = 0} x {cone boundary}, parametrised by the direction angle
    theta of the tangential impulse, BISECTION on
      form "energy"  : the minimum of E = 1/2 lam.G lam + lam.v, the contact-space kinetic energy (Hwangbo, Lee, Hutter 2018: the
                       per-contact maximum-dissipation principle): section search inside the bracket of the least value on a 0.5 degree grid;
      form "coulomb" : the component of the post-impulse tangential velocity across the impulse's direction (classical Coulomb:
                       slip anti-parallel to the friction impulse), roots with d . v_t+ < 0.
    "coulomb": every sign change on a 0.5 degree grid is bisected; of several roots the one with the least energy is returned."""
    G, v = np.asarray(G, float), np.asarray(v, float)
    if v[2] > 0:
        return np.zeros(3), "open"
    ls = -np.linalg.solve(G, v)
    if ls[2] >= 0 and np.hypot(ls[0], ls[1]) <= mu * ls[2]:
        return ls, "stick"

    def lam_of(th):
        d = np.array([np.cos(th), np.sin(th)])
        den = G[2, 2] + mu * (G[2, :2] @ d)
        if den <= 1e-12 * G[2, 2]:
            return None
        ln = -v[2] / den
        return np.r_[mu * ln * d, ln]

    def f(th):
        lam = lam_of(th)
        if lam is None:
            return None
        if form == "coulomb":      # z-component of d x v_t+ : zero where the slip velocity is (anti-)parallel to the impulse; second: d . v_t+
            vt = v[:2] + G[:2, :] @ lam
            return np.cos(th) * vt[1] - np.sin(th) * vt[0], np.cos(th) * vt[0] + np.sin(th) * vt[1]
        eps = 1e-6                 # dE/dtheta = v+ . dlam/dtheta by a central difference of the curve; second: minus its slope (a minimum has d2E > 0)
        lp_, lm_ = lam_of(th + eps), lam_of(th - eps)
        if lp_ is None or lm_ is None:
            return None
        de = lambda l: 0.5 * l @ G @ l + l @ v                    # noqa: E731
        return (de(lp_) - de(lm_)) / (2 * eps), -(de(lp_) - 2 * de(lam) + de(lm_)) / eps ** 2

    grid = np.linspace(-np.pi, np.pi, 721)
    if form == "energy":
        # the minimum can sit at the end of the feasible arc (den -> 0), where dE/dtheta has no root: bracket the least grid value and
        # shrink the bracket by golden section on E itself (a plain one-dimensional section search)
        en = lambda t: (lambda l: np.inf if l is None else 0.5 * l @ G @ l + l @ v)(lam_of(t))     # noqa: E731
        vals_e = np.array([en(t) for t in grid])
        k = int(np.argmin(vals_e))
        lo, hi = grid[max(k - 1, 0)], grid[min(k + 1, len(grid) - 1)]
        gr = 0.5 * (np.sqrt(5.0) - 1.0)
        for _ in range(iters):
            a_, b_ = hi - gr * (hi - lo), lo + gr * (hi - lo)
            if en(a_) < en(b_):
                hi = b_
            else:
                lo = a_
        return lam_of(0.5 * (lo + hi)), "slip"
    # bracket every sign change of f on a fine grid, bisect, keep the roots with d . v_t+ < 0; several roots: the most dissipative
    vals = [f(t) for t in grid]
    roots = []
    for k in range(len(grid) - 1):
        a, b = vals[k], vals[k + 1]
        if a is None or b is None or a[0] * b[0] > 0:
            continue
        lo, hi, flo = grid[k], grid[k + 1], a[0]
        for _ in range(iters):
            mid = 0.5 * (lo + hi)
            fm = f(mid)
            if fm is None:
                break
            if (fm[0] > 0) == (flo > 0):
                lo, flo = mid, fm[0]
            else:
                hi = mid
        fm = f(0.5 * (lo + hi))
        if fm is not None and fm[1] < 0:
            roots.append(0.5 * (lo + hi))
    if not roots:
        return None, "slip-no-root"
    cand = [lam_of(t) for t in roots]
    E = [0.5 * l @ G @ l + l @ v for l in cand]
    return cand[int(np.argmin(E))], "slip"


def _random_blocks(n, seed, coupled=True):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        A = rng.normal(size=(3, 5))
        G = A @ A.T / 5 + 0.05 * np.eye(3)            # a Delassus block: symmetric positive definite
        if not coupled:
            G[2, :2] = G[:2, 2] = 0.0                  # no normal-tangential coupling (a point mass, a sphere pushed through its centre)
        v = rng.normal(size=3)
        v[2] = -abs(v[2])
        out.append((G, v, rng.uniform(0.2, 1.2)))
    return out


def test_one_contact_rule_against_two_textbook_bisections_on_random_blocks(anymal):
    """(A) the energy form by plain bisection == the oracle's accelerated search; (B) the classical Coulomb form == both where the
    normal row does not couple with the tangential ones (the curve is then a circle |lam_t| = const and "v_t+ anti-parallel to
    lam_t" and "dE/dtheta = 0" are one equation) and differs with coupling, by how much is measured and quoted in DESIGN.md 2."""
    o = Oracle(anymal.blob)
    # decoupled blocks: all three identical
    worst_a = worst_b = 0.0
    kinds = {}
    for G, v, mu in _random_blocks(200, 11, coupled=False):
        la, kind = bisection_contact(G, v, mu, "energy")
        lb, _ = bisection_contact(G, v, mu, "coulomb")
        lo = o.solve_contact(G, v, mu, section_rounds=8)
        kinds[kind] = kinds.get(kind, 0) + 1
        assert la is not None and lb is not None
        worst_a = max(worst_a, np.abs(la - lo).max() / (np.abs(la).max() + 1e-9))
        worst_b = max(worst_b, np.abs(lb - lo).max() / (np.abs(lb).max() + 1e-9))
    assert kinds.get("slip", 0) >= 50 and kinds.get("stick", 0) >= 20
    assert worst_a < 2e-5 and worst_b < 2e-6, (worst_a, worst_b)
    # coupled blocks (what a foot on a leg has)
    rel_a, rel_b, dE, coupling = [], [], [], []
    for G, v, mu in _random_blocks(400, 12, coupled=True):
        la, kind = bisection_contact(G, v, mu, "energy")
        lb, _ = bisection_contact(G, v, mu, "coulomb")
        lo = o.solve_contact(G, v, mu, section_rounds=8)
        if kind in ("open", "stick"):
            assert np.abs(la - lo).max() <= 1e-9 * (1 + np.abs(la).max())       # these cases are shared word for word
            continue
        assert la is not None
        Ea, Eo = 0.5 * la @ G @ la + la @ v, 0.5 * lo @ G @ lo + lo @ v
        assert abs((v + G @ lo)[2]) < 1e-7 * (1 + np.abs(v).max())              # on v_n+ = 0 and on the cone boundary
        assert abs(np.hypot(lo[0], lo[1]) - mu * lo[2]) < 1e-7 * (1 + lo[2])
        assert Eo <= Ea + 1e-7 * (abs(Ea) + 1e-3)                               # the accelerated search never ends above the brute-force minimum
        rel_a.append(np.abs(la - lo).max() / (np.abs(la).max() + 1e-12))
        if lb is not None:
            Eb = 0.5 * lb @ G @ lb + lb @ v
            assert Eo <= Eb + 1e-9 * (abs(Eb) + 1)
            rel_b.append(np.abs(lb - lo).max() / (np.abs(lb).max() + 1e-12))
            dE.append((Eb - Eo) / (abs(Eb) + 1e-12))
            coupling.append(np.hypot(G[2, 0], G[2, 1]) * mu / G[2, 2])
    rel_a, rel_b, dE, coupling = map(np.array, (rel_a, rel_b, dE, coupling))
    lo_c, hi_c = coupling < 0.1, coupling > 0.4
    print(f"400 random coupled blocks, {len(rel_a)} slip cases.  (A) energy form by plain bisection vs the oracle: relative impulse difference "
          f"p50 {np.median(rel_a):.1e} p99 {np.percentile(rel_a, 99):.1e} max {rel_a.max():.1e}.  (B) classical Coulomb form vs the oracle ({len(rel_b)} with a root): "
          f"p50 {np.median(rel_b):.1e} p90 {np.percentile(rel_b, 90):.1e}; weak coupling (mu |G_nt| / G_nn < 0.1, {int(lo_c.sum())} cases) p50 {np.median(rel_b[lo_c]):.1e}; "
          f"strong (> 0.4, {int(hi_c.sum())} cases) p50 {np.median(rel_b[hi_c]):.1e}; energy the Coulomb form leaves on the table p50 {np.median(dE):.1e}")
    assert len(rel_a) >= 150
    assert np.percentile(rel_a, 99) < 1e-4                        # (A): the same answer (the brute-force bracket and the 16-direction scan find the same minimum)
    assert np.median(rel_b[lo_c]) < 0.05                          # (B): coincides near the decoupled limit ...
    assert np.median(rel_b[hi_c]) > np.median(rel_b[lo_c])        # ... and departs with the coupling


def test_one_contact_rule_on_the_quadruped_foot_blocks(anymal):
    """The same on the blocks that matter: the diagonal Delassus blocks of the feet of the standing quadruped (config-2 recipe) with
    sliding feet, where the coupling is what a bent leg gives it (a normal impulse on the foot moves it sideways: mu |G_nt| / G_nn
    ~ 0.8).  (A) must agree; (B) is reported."""
    from raisimlib_amd import workload
    o = Oracle(anymal.blob)
    kp, kd = (a.astype(np.float64) for a in workload.anymal_gains())
    gc, gv = workload.anymal_initial_state(48, height=0.52)
    rng = np.random.default_rng(3)
    gv = gv + rng.normal(size=gv.shape) * 0.8              # make the feet slide
    rel_a, rel_b, ang, coupling = [], [], [], []
    for e in range(48):
        d = o.step_debug(gc[e], gv[e], kp, kd, workload.anymal_targets(48, 0)[e], np.zeros(18))
        n = len(d["c"]) // 3
        for i in range(n):
            G = d["G"][3 * i:3 * i + 3, 3 * i:3 * i + 3]
            v = d["c"][3 * i:3 * i + 3]
            la, kind = bisection_contact(G, v, 0.8, "energy")
            if kind != "slip":
                continue
            lb, _ = bisection_contact(G, v, 0.8, "coulomb")
            lo = o.solve_contact(G, v, 0.8, section_rounds=8)
            rel_a.append(np.abs(la - lo).max() / (np.abs(la).max() + 1e-12))
            vt = (v + G @ lo)[:2]
            ang.append(np.degrees(np.arccos(np.clip(-(vt @ lo[:2]) / (np.linalg.norm(vt) * np.linalg.norm(lo[:2]) + 1e-30), -1, 1))))
            coupling.append(0.8 * np.hypot(G[2, 0], G[2, 1]) / G[2, 2])
            if lb is not None:
                rel_b.append(np.abs(lb - lo).max() / (np.abs(lb).max() + 1e-12))
    rel_a, rel_b, ang, coupling = map(np.array, (rel_a, rel_b, ang, coupling))
    assert len(rel_a) >= 30
    print(f"quadruped foot blocks: {len(rel_a)} slipping one-contact problems, coupling mu |G_nt| / G_nn p50 {np.median(coupling):.2f}.  (A) energy form by plain "
          f"bisection vs the oracle: p50 {np.median(rel_a):.1e} max {rel_a.max():.1e}.  (B) classical Coulomb form vs the oracle: relative impulse difference "
          f"p50 {np.median(rel_b):.1e} p90 {np.percentile(rel_b, 90):.1e}; angle between the oracle's friction impulse and the opposite of its slip velocity "
          f"p50 {np.median(ang):.0f} deg p90 {np.percentile(ang, 90):.0f} deg")
    assert np.percentile(rel_a, 99) < 1e-4


def test_coulomb_slip_rule_of_the_oracle_against_the_textbook_bisection(anymal):
    """Round 5 (VERDICT r04 #4b): `slip_rule = COULOMB` in the oracle (what rsb_set_slip_rule selects on the device) - the one-contact rule with the
    slip point where the post-impulse slip velocity is anti-parallel to the friction impulse.  On random COUPLED blocks it returns the root the
    brute-force bisection (B) above finds - of several roots possibly another upward crossing: then both satisfy the law - and always a point of
    the curve {v_n+ = 0} x {cone boundary} that obeys Coulomb's law to 1e-6; without a bracketed root it falls back to the energy rule (counted);
    open and stick cases are shared with the energy rule word for word."""
    o = Oracle(anymal.blob)
    n_slip = n_same = n_fallback = 0
    worst_law = 0.0
    for G, v, mu in _random_blocks(600, 21, coupled=True):
        le = o.solve_contact(G, v, mu, section_rounds=8)
        lc = o.solve_contact(G, v, mu, section_rounds=8, rule=1)
        lb, kind = bisection_contact(G, v, mu, "coulomb")
        if kind in ("open", "stick"):
            assert np.array_equal(le, lc)
            continue
        vp = v + G @ lc
        assert abs(vp[2]) < 1e-7 * (1 + np.abs(v).max()) and abs(np.hypot(lc[0], lc[1]) - mu * lc[2]) < 1e-7 * (1 + lc[2]) and lc[2] >= 0
        d = lc[:2] / (np.hypot(lc[0], lc[1]) + 1e-300)
        cross, along = d[0] * vp[1] - d[1] * vp[0], d @ vp[:2]
        obeys = abs(cross) <= 1e-6 * (np.hypot(vp[0], vp[1]) + 1e-6) and along <= 1e-9
        if not obeys:
            # the fallback: no upward crossing between two feasible grid directions -> the energy rule's point
            assert np.abs(lc - le).max() <= 1e-9 * (1 + np.abs(le).max()), (lc, le)
            n_fallback += 1
            continue
        n_slip += 1
        worst_law = max(worst_law, abs(cross) / (np.hypot(vp[0], vp[1]) + 1e-6))
        if lb is not None and np.abs(lb - lc).max() <= 1e-5 * (1 + np.abs(lb).max()):
            n_same += 1
    assert n_slip >= 150 and n_fallback <= 0.1 * (n_slip + n_fallback), (n_slip, n_fallback)
    assert n_same >= 0.9 * n_slip, (n_same, n_slip)        # the same root as the brute-force bisection (its pick among several roots: least energy)
    print(f"\n[coulomb rule] {n_slip} slipping blocks obey Coulomb's law (worst |v_t+ x d| / |v_t+| {worst_law:.1e}), {n_same} equal to the brute-force bisection's root, "
          f"{n_fallback} without a bracketed root (energy rule)")
    # decoupled blocks: Coulomb == energy (the curve is a circle)
    for G, v, mu in _random_blocks(100, 22, coupled=False):
        le, lc = o.solve_contact(G, v, mu, section_rounds=8), o.solve_contact(G, v, mu, section_rounds=8, rule=1)
        assert np.abs(le - lc).max() <= 2e-6 * (1 + np.abs(le).max())
