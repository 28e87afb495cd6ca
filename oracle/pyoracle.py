"""ctypes wrapper of oracle/_build/librsb_oracle.so — TEST INFRASTRUCTURE ONLY (see rsb_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED: the oracle is written from published algorithms; /root/reference has no source.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raisimlib_amd._capi import ModelBlob, RSB_MAX_CONTACTS

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "librsb_oracle.so")


class Params(C.Structure):
    _fields_ = [
        ("dt", C.c_double), ("gravity", C.c_double * 3), ("mu", C.c_double), ("erp", C.c_double),
        ("alpha_init", C.c_double), ("alpha_min", C.c_double), ("alpha_decay", C.c_double),
        ("threshold", C.c_double),
        ("max_iter", C.c_int32), ("section_rounds", C.c_int32), ("kmax", C.c_int32), ("control_mode", C.c_int32),
        ("warm_start", C.c_int32), ("freeze_after", C.c_int32),
        ("terrain_type", C.c_int32), ("hm_xs", C.c_int32), ("hm_ys", C.c_int32), ("stall_window", C.c_int32),
        ("dir_per_sweep", C.c_int32), ("refine", C.c_int32), ("group_parallel", C.c_int32), ("self_collision", C.c_int32),
        ("ground_z", C.c_double), ("stall_factor", C.c_double), ("restitution", C.c_double), ("res_threshold", C.c_double),
        ("settle_tol", C.c_double),
        ("hm_xsize", C.c_double), ("hm_ysize", C.c_double), ("hm_cx", C.c_double), ("hm_cy", C.c_double),
        ("hm_heights", C.c_void_p),
        ("col_mu", C.c_void_p), ("col_restitution", C.c_void_p), ("col_res_threshold", C.c_void_p),
        ("self_ignore", C.c_void_p), ("self_mu", C.c_void_p), ("self_restitution", C.c_void_p), ("self_res_threshold", C.c_void_p),
        ("hm_index", C.c_void_p),
        ("multi_depth", C.c_int32), ("multi_light", C.c_int32), ("multi_freeze_after", C.c_int32), ("multi_stall_window", C.c_int32),
        ("body_stick", C.c_int32),
        ("anderson", C.c_int32),
        ("anderson_clip", C.c_double),
        ("hm_contacts", C.c_int32),
        ("hm_second_cos", C.c_double),
        ("integ_theta", C.c_double),
        ("hm_capsule", C.c_int32),
        ("hm_plane_test", C.c_int32),
        ("slip_rule", C.c_int32),
        ("pair_inner", C.c_int32),
        ("reduce_dist", C.c_double),
    ]


class OContact(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("normal", C.c_double * 3), ("impulse", C.c_double * 3),
                ("depth", C.c_double), ("body", C.c_int32), ("collision", C.c_int32)]


CONTACT_DTYPE = np.dtype([("position", "f8", 3), ("normal", "f8", 3), ("impulse", "f8", 3), ("depth", "f8"),
                          ("body", "i4"), ("collision", "i4")])

_lib = None


def build():
    subprocess.run(["make", "-C", HERE, "-s"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        _lib = C.CDLL(SO)
        _lib.orc_max_threads.restype = C.c_int
        _lib.orc_step_batch.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _d(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    """fp64 single-env / batched CPU oracle bound to one model blob."""

    def __init__(self, blob: ModelBlob):
        self.L = lib()
        self.blob = blob
        self.nq, self.nv, self.nb = blob.nq, blob.nv, blob.nb
        self.p = Params()
        self.L.orc_default_params(C.byref(self.p))
        self._hm = None

    # -- configuration ---------------------------------------------------------------------
    def set_heightmap(self, xs, ys, xsize, ysize, cx, cy, heights):
        self._hm = np.ascontiguousarray(heights, dtype=np.float32).reshape(ys, xs)
        self.p.terrain_type = 1
        self.p.hm_xs, self.p.hm_ys = xs, ys
        self.p.hm_xsize, self.p.hm_ysize, self.p.hm_cx, self.p.hm_cy = xsize, ysize, cx, cy
        self.p.hm_heights = self._hm.ctypes.data

    def set_heightmaps(self, heights, xsize, ysize, cx, cy, env_map):
        """Terrain curricula for step_batch(): heights [n_maps, ys, xs], env_map [N] -> the map env e stands on."""
        h = np.ascontiguousarray(heights, dtype=np.float32)
        self.set_heightmap(h.shape[2], h.shape[1], xsize, ysize, cx, cy, h[0])
        self._hm = h
        self._hmi = np.ascontiguousarray(env_map, dtype=np.int32)
        self.p.hm_heights = self._hm.ctypes.data
        self.p.hm_index = self._hmi.ctypes.data

    def set_collision_materials(self, mu=None, restitution=None, res_threshold=None):
        """Per collision primitive contact material against the terrain ([ncol] arrays; None = the scalar default)."""
        self._cm = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (mu, restitution, res_threshold)]
        self.p.col_mu, self.p.col_restitution, self.p.col_res_threshold = (None if a is None else a.ctypes.data for a in self._cm)

    def set_self_collision(self, enable=True, ignore=None, mu=None, restitution=None, res_threshold=None):
        """Self-collision between non-adjacent bodies; ignore: [nb, nb] bool (ignoreCollisionBetween), materials per candidate pair."""
        self.p.self_collision = int(bool(enable))
        self._si = None if ignore is None else np.ascontiguousarray(ignore, dtype=np.uint8).reshape(self.nb, self.nb)
        self.p.self_ignore = None if self._si is None else self._si.ctypes.data
        self._sm = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (mu, restitution, res_threshold)]
        self.p.self_mu, self.p.self_restitution, self.p.self_res_threshold = (None if a is None else a.ctypes.data for a in self._sm)

    def self_pairs(self):
        """[(i, j)] candidate primitive pairs of self-collision in enumeration order."""
        ign = getattr(self, "_si", None)
        n = self.L.orc_self_pairs(C.byref(self.blob), _p(ign), None, 0)
        out = np.zeros((n, 2), np.int32)
        self.L.orc_self_pairs(C.byref(self.blob), _p(ign), _p(out), n)
        return out

    def set_ground(self, z):
        self.p.terrain_type = 0
        self.p.ground_z = z

    # -- queries -----------------------------------------------------------------------------
    def mass_matrix(self, q):
        M = np.zeros((self.nv, self.nv))
        self.L.orc_mass_matrix(C.byref(self.blob), _p(_d(q)), _p(M))
        return M

    def mass_matrix_rne(self, q):
        M = np.zeros((self.nv, self.nv))
        self.L.orc_mass_matrix_rne(C.byref(self.blob), _p(_d(q)), _p(M))
        return M

    def nonlinearities(self, q, u):
        h = np.zeros(self.nv)
        self.L.orc_nonlinearities(C.byref(self.blob), C.byref(self.p), _p(_d(q)), _p(_d(u)), _p(h))
        return h

    def inverse_dynamics(self, q, u, udot):
        t = np.zeros(self.nv)
        self.L.orc_inverse_dynamics(C.byref(self.blob), C.byref(self.p), _p(_d(q)), _p(_d(u)), _p(_d(udot)), _p(t))
        return t

    def aba(self, q, u, tau):
        a = np.zeros(self.nv)
        self.L.orc_aba(C.byref(self.blob), C.byref(self.p), _p(_d(q)), _p(_d(u)), _p(_d(tau)), _p(a))
        return a

    def forward_dynamics(self, q, u, tau):
        a = np.zeros(self.nv)
        self.L.orc_forward_dynamics(C.byref(self.blob), C.byref(self.p), _p(_d(q)), _p(_d(u)), _p(_d(tau)), _p(a))
        return a

    def point_jacobian(self, q, body, p_local):
        pos = np.zeros(3)
        J = np.zeros((3, self.nv))
        self.L.orc_point_jacobian(C.byref(self.blob), _p(_d(q)), C.c_int(body), _p(_d(p_local)), _p(pos), _p(J))
        return pos, J

    def energy(self, q, u):
        k, v = C.c_double(), C.c_double()
        self.L.orc_energy(C.byref(self.blob), C.byref(self.p), _p(_d(q)), _p(_d(u)), C.byref(k), C.byref(v))
        return k.value, v.value

    def momentum(self, q, u):
        lin, ang = np.zeros(3), np.zeros(3)
        self.L.orc_momentum(C.byref(self.blob), _p(_d(q)), _p(_d(u)), _p(lin), _p(ang))
        return lin, ang

    def terrain(self, x, y):
        h = C.c_double()
        n = np.zeros(3)
        self.L.orc_terrain(C.byref(self.p), C.c_double(x), C.c_double(y), C.byref(h), _p(n))
        return h.value, n

    def actuation(self, q, u, kp, kd, pt, dt_, tau_ff=None):
        t = np.zeros(self.nv)
        self.L.orc_actuation(C.byref(self.blob), C.byref(self.p), _p(_d(q)), _p(_d(u)), _p(_d(kp)), _p(_d(kd)),
                             _p(_d(pt)), _p(_d(dt_)), _p(_d(tau_ff)), _p(t))
        return t

    def solve_contact(self, G, v, mu, section_rounds=2, rule=0):
        """Open / stick / slip rule for one isolated contact (G 3x3 contact-frame Delassus block, v free velocity); rule 0 = the published
        least-energy slip point, 1 = classical Coulomb (orc_params::slip_rule)."""
        lam = np.zeros(3)
        if rule:
            self.L.orc_solve_contact_rule(_p(_d(np.asarray(G).reshape(9))), _p(_d(v)), C.c_double(mu), C.c_int(section_rounds), C.c_int(rule), _p(lam))
            return lam
        self.L.orc_solve_contact(_p(_d(np.asarray(G).reshape(9))), _p(_d(v)), C.c_double(mu), C.c_int(section_rounds), _p(lam))
        return lam

    # -- stepping ----------------------------------------------------------------------------
    def step(self, q, u, kp=None, kd=None, pt=None, dt_=None, tau_ff=None):
        """One integrate() of one env. Returns (q+, u+, contacts(structured array), iters, flags)."""
        q = np.array(q, dtype=np.float64).copy()
        u = np.array(u, dtype=np.float64).copy()
        con = np.zeros(RSB_MAX_CONTACTS, dtype=CONTACT_DTYPE)
        nc, it, fl = C.c_int32(), C.c_int32(), C.c_int32()
        self.L.orc_step(C.byref(self.blob), C.byref(self.p), _p(q), _p(u), _p(_d(kp)), _p(_d(kd)), _p(_d(pt)),
                        _p(_d(dt_)), _p(_d(tau_ff)), _p(con), C.byref(nc), C.byref(it), C.byref(fl))
        return q, u, con[:nc.value], it.value, fl.value

    def step_debug(self, q, u, kp=None, kd=None, pt=None, dt_=None, tau_ff=None, lam_warm=None):
        """As step(), plus the contact problem (G [3nc,3nc], c [3nc], lam [3nc]) in contact-frame coordinates."""
        q = np.array(q, dtype=np.float64).copy()
        u = np.array(u, dtype=np.float64).copy()
        K = RSB_MAX_CONTACTS
        con = np.zeros(K, dtype=CONTACT_DTYPE)
        G = np.zeros(9 * K * K)
        c = np.zeros(3 * K)
        lam = np.zeros(3 * K)
        nc, it, fl = C.c_int32(), C.c_int32(), C.c_int32()
        self.L.orc_step_debug(C.byref(self.blob), C.byref(self.p), _p(q), _p(u), _p(_d(kp)), _p(_d(kd)), _p(_d(pt)),
                              _p(_d(dt_)), _p(_d(tau_ff)), _p(con), C.byref(nc), C.byref(it), C.byref(fl),
                              _p(lam_warm), _p(G), _p(c), _p(lam))
        n3 = 3 * (nc.value - int((con["collision"][:nc.value] & 0x20000).astype(bool).sum()))   # a self-collision: two entries, one solver contact
        return dict(q=q, u=u, contacts=con[:nc.value], iters=it.value, flags=fl.value,
                    G=G[:n3 * n3].reshape(n3, n3).copy(), c=c[:n3].copy(), lam=lam[:n3].copy())

    def new_warm_state(self, n):
        """Zeroed warm-start state for n envs ([n, 6*ncol] float64: impulse 3, friction direction 2, valid flag per
        collision primitive), to be passed to step_batch(lam_warm=...) / step_debug(lam_warm=...)."""
        return np.zeros((n, 6 * self.blob.ncol))

    def step_batch(self, q, u, substeps=1, kp=None, kd=None, pt=None, dt_=None, tau_ff=None, nthreads=0,
                   want_contacts=False, lam_warm=None):
        """N envs x `substeps` integrate() calls (OpenMP over envs). q,u: [N,nq],[N,nv] float64, updated copies
        are returned."""
        q = np.ascontiguousarray(q, dtype=np.float64).copy()
        u = np.ascontiguousarray(u, dtype=np.float64).copy()
        N = q.shape[0]
        kmax = self.p.kmax
        con = np.zeros((N, kmax), dtype=CONTACT_DTYPE) if want_contacts else None
        ncs = np.zeros(N, dtype=np.int32)
        its = np.zeros(N, dtype=np.int32)
        fls = np.zeros(N, dtype=np.int32)
        used = self.L.orc_step_batch(C.byref(self.blob), C.byref(self.p), C.c_int(N), C.c_int(substeps), _p(q), _p(u),
                                     _p(_d(kp)), _p(_d(kd)), _p(_d(pt)), _p(_d(dt_)), _p(_d(tau_ff)), _p(con),
                                     _p(ncs), _p(its), _p(fls), _p(lam_warm), C.c_int(nthreads))
        out = dict(q=q, u=u, n_contacts=ncs, iters=its, flags=fls, threads=used)
        if want_contacts:
            out["contacts"] = con
        return out

    def max_threads(self):
        return self.L.orc_max_threads()
