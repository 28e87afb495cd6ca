"""Thin Python host mirror of the batched world behind the C-ABI (include/rsb.h).

`Model` wraps rsb_model (URDF -> flat blob, host only); `BatchedWorld` wraps rsb_world: N replicas of
{raisim::World + one raisim::ArticulatedSystem + Ground/HeightMap} resident on one GPU.  Method names
follow the raisim::World / raisim::ArticulatedSystem surface [RECALL, SURVEY.md §8b] in snake_case.
Everything here is plumbing: pointers in, status codes out.  No physics and no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import (RSB_DEVICE, RSB_F_GENERALIZED_FORCE, RSB_HOST, RSB_MAX_CONTACTS, Contact, ModelBlob, check, lib)

RSC_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rsc")

CONTACT_DTYPE = np.dtype([("position", "f4", 3), ("normal", "f4", 3), ("impulse", "f4", 3), ("depth", "f4"),
                          ("body", "i4"), ("collision", "i4")])
assert CONTACT_DTYPE.itemsize == C.sizeof(Contact)


def rsc_path(name):
    return os.path.join(RSC_DIR, name)


class Model:
    """Host-side articulated-system description (the role of World::addArticulatedSystem's URDF load)."""

    def __init__(self, urdf_path=None, urdf_string=None, blob=None, sample_spacing=0.0):
        """sample_spacing > 0: sampled colliders (capsule axes / box surfaces get sample primitives no further apart; rsb.h)"""
        L = lib()
        h = C.c_void_p()
        if urdf_path is not None:
            check(L.rsb_model_from_urdf_file_sampled(os.fspath(urdf_path).encode(), float(sample_spacing), C.byref(h)), "rsb_model_from_urdf_file")
        elif urdf_string is not None:
            check(L.rsb_model_from_urdf_string_sampled(urdf_string.encode(), float(sample_spacing), C.byref(h)), "rsb_model_from_urdf_string")
        elif blob is not None:
            check(L.rsb_model_from_blob(C.byref(blob), C.byref(h)), "rsb_model_from_blob")
        else:
            raise ValueError("Model needs urdf_path, urdf_string or blob")
        self.handle = h
        self.blob = ModelBlob()
        check(L.rsb_model_get_blob(self.handle, C.byref(self.blob)), "rsb_model_get_blob")

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                lib().rsb_model_destroy(h)
            except Exception:      # interpreter shutdown: the module globals lib() needs may already be gone
                pass
            self.handle = None

    nb = property(lambda self: self.blob.nb)
    nq = property(lambda self: self.blob.nq)
    nv = property(lambda self: self.blob.nv)
    ncol = property(lambda self: self.blob.ncol)
    skipped_collisions = property(lambda self: lib().rsb_model_skipped_collisions(self.handle))   # <collision> elements the loader dropped

    def body_index(self, link_name):
        return lib().rsb_model_body_index(self.handle, link_name.encode())

    def joint_index(self, joint_name):
        return lib().rsb_model_joint_index(self.handle, joint_name.encode())

    def total_mass(self):
        return lib().rsb_model_total_mass(self.handle)

    def body_names(self):
        return [self.blob.body_name[i].value.decode() for i in range(self.nb)]

    def collision_names(self):
        return [self.blob.col_name[i].value.decode() for i in range(self.ncol)]

    def lds_bytes(self, kmax=8, self_collision=True, lanes_per_env=0):
        """LDS bytes of one workgroup of the step kernel for this model (host only; rsb_model_lds_bytes)"""
        n = lib().rsb_model_lds_bytes(self.handle, int(kmax), int(bool(self_collision)), int(lanes_per_env))
        check(min(n, 0), "rsb_model_lds_bytes")
        return n

    def collision_indices(self, suffix):
        return [i for i, n in enumerate(self.collision_names()) if n.endswith(suffix)]

    def collision_materials(self):
        """Material name of each collision primitive (<collision><material name=../> of the URDF, "default" if absent)."""
        return [(self.blob.col_material[i].value.decode() or "default") for i in range(self.ncol)]


def heightmap_from_png(path, height_scale=1.0, height_offset=0.0):
    """[y_samples, x_samples] float32 heights of a PNG (rsb_heightmap_png_*)."""
    L = lib()
    xs, ys = C.c_int(), C.c_int()
    check(L.rsb_heightmap_png_size(os.fspath(path).encode(), C.byref(xs), C.byref(ys)), "rsb_heightmap_png_size")
    h = np.zeros((ys.value, xs.value), np.float32)
    check(L.rsb_heightmap_png_read(os.fspath(path).encode(), float(height_scale), float(height_offset), _hp(h), h.size),
          "rsb_heightmap_png_read")
    return h


def heightmap_perlin(x_samples=100, y_samples=100, x_size=10.0, y_size=10.0, frequency=0.1, z_scale=1.0, fractal_octaves=5,
                     fractal_lacunarity=2.0, fractal_gain=0.5, step_size=0.0, seed=6479, height_offset=0.0):
    """[y_samples, x_samples] float32 Perlin terrain (rsb_heightmap_perlin; defaults = raisim::TerrainProperties [RECALL])."""
    tp = _capi.TerrainProperties(frequency, z_scale, x_size, y_size, x_samples, y_samples, fractal_octaves, seed,
                                 fractal_lacunarity, fractal_gain, step_size, height_offset)
    h = np.zeros((y_samples, x_samples), np.float32)
    check(lib().rsb_heightmap_perlin(C.byref(tp), _hp(h)), "rsb_heightmap_perlin")
    return h


def heightmap_from_text(path):
    """(heights [y_samples, x_samples] float32, x_size, y_size) of the text format (rsb_heightmap_text_*)."""
    L = lib()
    xs, ys, sx, sy = C.c_int(), C.c_int(), C.c_double(), C.c_double()
    check(L.rsb_heightmap_text_size(os.fspath(path).encode(), C.byref(xs), C.byref(ys), C.byref(sx), C.byref(sy)),
          "rsb_heightmap_text_size")
    h = np.zeros((ys.value, xs.value), np.float32)
    check(L.rsb_heightmap_text_read(os.fspath(path).encode(), _hp(h), h.size), "rsb_heightmap_text_read")
    return h, sx.value, sy.value


def _host(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def _hp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class BatchedWorld:
    def __init__(self, model: Model, num_envs: int, device: int = 0):
        self.L = lib()
        self.model = model
        h = C.c_void_p()
        check(self.L.rsb_create(model.handle, int(num_envs), int(device), C.byref(h)), "rsb_create")
        self.handle = h
        self.N, self.nq, self.nv = num_envs, model.nq, model.nv
        self.device = device

    def close(self):
        if getattr(self, "handle", None):
            self.L.rsb_destroy(self.handle)
            self.handle = None

    __del__ = close

    # -- raisim::World surface ---------------------------------------------------------------
    def set_time_step(self, dt):
        check(self.L.rsb_set_timestep(self.handle, float(dt)), "rsb_set_timestep")

    def get_time_step(self):
        return self.L.rsb_get_timestep(self.handle)

    def get_world_time(self):
        return self.L.rsb_get_world_time(self.handle)

    def set_gravity(self, g):
        arr = (C.c_double * 3)(*[float(x) for x in g])
        check(self.L.rsb_set_gravity(self.handle, arr), "rsb_set_gravity")

    def set_erp(self, erp):
        check(self.L.rsb_set_erp(self.handle, float(erp)), "rsb_set_erp")

    def set_default_material(self, mu, restitution=0.0, res_threshold=0.0):
        check(self.L.rsb_set_material(self.handle, float(mu), float(restitution), float(res_threshold)), "rsb_set_material")

    def set_collision_materials(self, mu=None, restitution=None, res_threshold=None):
        """Per collision primitive contact material against the terrain (the resolved World::setMaterialPairProp table):
        [ncol] arrays, None (or a negative entry) = the world's default material."""
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (mu, restitution, res_threshold)]
        for a in arrs:
            if a is not None and a.shape != (self.model.ncol,):
                raise ValueError("set_collision_materials: arrays of ncol entries expected")
        check(self.L.rsb_set_collision_materials(self.handle, *[None if a is None else a.ctypes.data for a in arrs]),
              "rsb_set_collision_materials")

    def enable_generalized_force_output(self, on=True):
        """Have every launch also write the generalized force the actuators applied in its last sub-step ([N, nv])."""
        check(self.L.rsb_enable_generalized_force_output(self.handle, int(bool(on))), "rsb_enable_generalized_force_output")

    def get_generalized_force(self):
        """ArticulatedSystem::getGeneralizedForce() of every env: clipped PD + feed-forward of the last sub-step, [N, nv]."""
        out = np.zeros((self.N, self.model.nv), np.float32)
        check(self.L.rsb_get_field(self.handle, RSB_F_GENERALIZED_FORCE, _hp(out), RSB_HOST), "rsb_get_field(GENERALIZED_FORCE)")
        return out

    def get_pd_target(self):
        """the world's own copy of the position targets, [N, nq] (what the last control step / setPdTarget left)"""
        out = np.zeros((self.N, self.model.nq), np.float32)
        check(self.L.rsb_get_field(self.handle, 2, _hp(out), RSB_HOST), "rsb_get_field(PTARGET)")
        return out

    def set_self_collision(self, enable=True):
        """Collisions between non-adjacent bodies of the system (sphere x sphere; on by default, as in RaiSim)."""
        check(self.L.rsb_set_self_collision(self.handle, int(bool(enable))), "rsb_set_self_collision")

    def ignore_collision_between(self, body_a, body_b):
        """ArticulatedSystem::ignoreCollisionBetween(bodyIdx1, bodyIdx2); per-pair materials are reset."""
        check(self.L.rsb_ignore_collision_between(self.handle, int(body_a), int(body_b)), "rsb_ignore_collision_between")

    def self_collision_pairs(self):
        """[(i, j)] candidate primitive pairs (i < j) of self-collision."""
        n = self.L.rsb_self_collision_pairs(self.handle, None, 0)
        out = np.zeros((n, 2), np.int32)
        if n:
            self.L.rsb_self_collision_pairs(self.handle, out.ctypes.data, n)
        return out

    def set_self_collision_materials(self, mu=None, restitution=None, res_threshold=None):
        """One (mu, restitution, res_threshold) per candidate pair of self_collision_pairs(); None / negative = world default."""
        n = len(self.self_collision_pairs())
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (mu, restitution, res_threshold)]
        for a in arrs:
            if a is not None and a.shape != (n,):
                raise ValueError("set_self_collision_materials: one entry per candidate pair expected")
        check(self.L.rsb_set_self_collision_materials(self.handle, *[None if a is None else a.ctypes.data for a in arrs]),
              "rsb_set_self_collision_materials")

    def set_default_friction(self, mu):
        check(self.L.rsb_set_friction(self.handle, float(mu)), "rsb_set_friction")

    def set_contact_solver_param(self, alpha_init, alpha_min, alpha_decay, max_iter, threshold):
        check(self.L.rsb_set_contact_solver_param(self.handle, float(alpha_init), float(alpha_min), float(alpha_decay),
                                                  int(max_iter), float(threshold)), "rsb_set_contact_solver_param")

    def set_solver_stagnation_exit(self, window, factor):
        check(self.L.rsb_set_solver_stagnation_exit(self.handle, int(window), float(factor)), "rsb_set_solver_stagnation_exit")

    def set_solver_friction_lag(self, freeze_after, refine=True, settle_tol=1e-4):
        check(self.L.rsb_set_solver_friction_lag(self.handle, int(freeze_after), int(bool(refine)), float(settle_tol)),
              "rsb_set_solver_friction_lag")

    def set_solver_multi_contact(self, depth=3, light_passes=False, freeze_after=0, stall_window=16):
        """Solver settings of envs with >= depth contacts on one limb (redundant sets; see rsb.h). depth 0 = no distinction."""
        check(self.L.rsb_set_solver_multi_contact(self.handle, int(depth), int(bool(light_passes)), int(freeze_after), int(stall_window)),
              "rsb_set_solver_multi_contact")

    def set_integration_scheme(self, scheme="semi_implicit"):
        """'semi_implicit' (default), 'euler', 'trapezoid' (rsb.h: rsb_set_integration_scheme); 'runge_kutta_4' is refused."""
        code = {"trapezoid": 0, "semi_implicit": 1, "euler": 2, "runge_kutta_4": 3}[scheme] if isinstance(scheme, str) else int(scheme)
        check(self.L.rsb_set_integration_scheme(self.handle, code), "rsb_set_integration_scheme")

    def set_slip_rule(self, rule="energy"):
        """Slip rule of the per-contact iteration: "energy" (default: the published least-energy point) or "coulomb" (slip velocity anti-parallel to the
        friction impulse; a kernel class of its own, see rsb_set_slip_rule in include/rsb.h)."""
        code = {"energy": 0, "coulomb": 1}[rule] if isinstance(rule, str) else int(rule)
        check(self.L.rsb_set_slip_rule(self.handle, code), "rsb_set_slip_rule")

    def set_heightmap_contacts(self, per_primitive=2, min_angle_deg=45.0):
        """Contacts per collision primitive against a height map (1 = closest feature; 2 = also a second flank's; see rsb.h)."""
        check(self.L.rsb_set_heightmap_contacts(self.handle, int(per_primitive), float(min_angle_deg)), "rsb_set_heightmap_contacts")

    def set_step_pipelining(self, on=True):
        """Consecutive control steps overlap on the device (rsb_set_step_pipelining): bit-identical results, no launch waits for the slowest wave
        of the one before it; any other call joins the pipeline first."""
        check(self.L.rsb_set_step_pipelining(self.handle, int(bool(on))), "rsb_set_step_pipelining")
        return bool(self.L.rsb_step_pipelining_enabled(self.handle))     # False although asked for: RSB_STEP_PIPELINING=0 / a serialising profiler

    def step_pipelining_enabled(self):
        return bool(self.L.rsb_step_pipelining_enabled(self.handle))

    def step_pipeline_publish(self, stream_ptr):
        """`stream_ptr` (a hipStream_t as an integer) waits for the most recent pipelined control step; the pipeline keeps running"""
        check(self.L.rsb_step_pipeline_publish(self.handle, C.c_void_p(stream_ptr)), "rsb_step_pipeline_publish")

    def step_pipeline_wait_event(self, event_ptr):
        """the next control step additionally waits for the hipEvent_t `event_ptr` (an integer; the caller keeps it alive until then)"""
        check(self.L.rsb_step_pipeline_wait_event(self.handle, C.c_void_p(event_ptr)), "rsb_step_pipeline_wait_event")

    def step_pipelining_stats(self):
        a, b = C.c_longlong(0), C.c_longlong(0)
        st = self.L.rsb_step_pipelining_stats(self.handle, C.byref(a), C.byref(b))
        if st not in (0, 1):
            check(st, "rsb_step_pipelining_stats")
        self.pipeline_overlaps = st == 0        # False: no two streams on different hardware queues were found (correct, but in order)
        return int(a.value), int(b.value)

    def step_pipeline_join(self):
        """Waits for every pipelined step / stage pass in flight; raises RsbError (status RSB_E_PIPELINE) ONCE after a device-side fault - the
        library has by then restored the last joined state and replayed the steps in lock-step (include/rsb_pipeline.h)."""
        check(self.L.rsb_step_pipeline_join(self.handle), "rsb_step_pipeline_join")

    def step_pipeline_fault(self):
        """(faults so far, device code of the last one: RSB_PIPE_ERR_*)"""
        a, b = C.c_int(0), C.c_int(0)
        check(self.L.rsb_step_pipeline_fault(self.handle, C.byref(a), C.byref(b)), "rsb_step_pipeline_fault")
        return int(a.value), int(b.value)

    def debug_pipeline_wait_stats(self):
        """(mean wait of a pipelined step workgroup for its block in us, share of workgroups that waited); needs RSB_PIPE_STATS=1"""
        a, b = C.c_double(0), C.c_double(0)
        check(self.L.rsb_debug_pipeline_wait_stats(self.handle, C.byref(a), C.byref(b)), "rsb_debug_pipeline_wait_stats")
        return float(a.value), float(b.value)

    def debug_pipeline_fault(self, kind):
        """tests: the next pipelined launch fails on the device (1: ticket, 2: time-out, 4: error word set)"""
        check(self.L.rsb_debug_pipeline_fault(self.handle, int(kind)), "rsb_debug_pipeline_fault")

    def set_capsule_contacts(self, on=True):
        """Exact capsule / cylinder / box x height map: the barrel between a capsule's or cylinder's ends and the faces and edges between a box's corners
        report their deepest point (rsb_set_capsule_contacts)."""
        check(self.L.rsb_set_capsule_contacts(self.handle, int(bool(on))), "rsb_set_capsule_contacts")

    def set_solver_anderson(self, first_sweep=2, clip=20.0):
        """Anderson acceleration of the sweep in multi-contact envs of worlds with > 8 contact slots (see rsb.h). first_sweep 0 = off."""
        check(self.L.rsb_set_solver_anderson(self.handle, int(first_sweep), float(clip)), "rsb_set_solver_anderson")

    def set_early_termination(self, on=True):
        check(self.L.rsb_set_early_termination(self.handle, 1 if on else 0), "rsb_set_early_termination")

    def set_solver_warm_start(self, on=True):
        check(self.L.rsb_set_solver_warm_start(self.handle, 1 if on else 0), "rsb_set_solver_warm_start")

    def set_max_contacts(self, kmax):
        check(self.L.rsb_set_max_contacts(self.handle, int(kmax)), "rsb_set_max_contacts")

    def max_contacts(self):
        k = C.c_int()
        check(self.L.rsb_dims(self.handle, None, None, None, None, C.byref(k)), "rsb_dims")
        return k.value

    def set_lanes_per_env(self, lanes):
        check(self.L.rsb_set_lanes_per_env(self.handle, int(lanes)), "rsb_set_lanes_per_env")

    def lanes_per_env(self):
        return self.L.rsb_get_lanes_per_env(self.handle)

    def add_ground(self, z=0.0):
        check(self.L.rsb_set_ground(self.handle, float(z)), "rsb_set_ground")

    def add_height_map(self, x_samples, y_samples, x_size, y_size, center_x, center_y, heights):
        h = _host(heights, np.float32).reshape(y_samples, x_samples)
        check(self.L.rsb_set_heightmap(self.handle, int(x_samples), int(y_samples), float(x_size), float(y_size),
                                       float(center_x), float(center_y), _hp(h)), "rsb_set_heightmap")

    def add_height_maps(self, heights, x_size, y_size, center_x, center_y, env_map):
        """Terrain curricula: heights [n_maps, y_samples, x_samples], env_map [N] -> the map each env stands on."""
        h = _host(heights, np.float32)
        idx = _host(env_map, np.int32)
        assert h.ndim == 3 and idx.shape == (self.N,)
        check(self.L.rsb_set_heightmaps(self.handle, h.shape[0], h.shape[2], h.shape[1], float(x_size), float(y_size),
                                        float(center_x), float(center_y), _hp(h), _hp(idx)), "rsb_set_heightmaps")

    def integrate(self, n_substeps=1):
        check(self.L.rsb_integrate(self.handle, int(n_substeps)), "rsb_integrate")

    def integrate_masked(self, mask, n_substeps=1):
        """integrate() of the envs whose mask byte is non-zero only (host uint8 [N]); the others are left untouched."""
        m = _host(mask, np.uint8)
        assert m.shape == (self.N,)
        check(self.L.rsb_integrate_masked(self.handle, int(n_substeps), _hp(m), RSB_HOST), "rsb_integrate_masked")

    def set_done_output(self, done_device_ptr):
        """Device uint8 [N] buffer that every following control step fills with its done flags (0 / None: off)."""
        check(self.L.rsb_set_done_output(self.handle, C.c_void_p(done_device_ptr) if done_device_ptr else None), "rsb_set_done_output")

    def integrate1(self):
        check(self.L.rsb_integrate1(self.handle), "rsb_integrate1")

    def integrate2(self):
        check(self.L.rsb_integrate2(self.handle), "rsb_integrate2")

    def synchronize(self):
        check(self.L.rsb_synchronize(self.handle), "rsb_synchronize")

    def get_stream(self):
        """The world's stream (hipStream_t as an integer).  Joins pipelined control steps first: work enqueued on it afterwards sees them."""
        return self.L.rsb_get_stream(self.handle) or 0

    def set_stream(self, hip_stream_ptr):
        check(self.L.rsb_set_stream(self.handle, C.c_void_p(hip_stream_ptr)), "rsb_set_stream")

    # -- raisim::ArticulatedSystem surface (batched) ---------------------------------------------
    def set_state(self, gc=None, gv=None, mask=None):
        gc, gv, mask = _host(gc, np.float32), _host(gv, np.float32), _host(mask, np.uint8)
        check(self.L.rsb_set_state(self.handle, _hp(gc), _hp(gv), _hp(mask), RSB_HOST), "rsb_set_state")

    def get_state(self):
        gc = np.empty((self.N, self.nq), np.float32)
        gv = np.empty((self.N, self.nv), np.float32)
        check(self.L.rsb_get_state(self.handle, _hp(gc), _hp(gv), RSB_HOST), "rsb_get_state")
        return gc, gv

    def set_control_mode(self, mode):
        check(self.L.rsb_set_control_mode(self.handle, int(mode)), "rsb_set_control_mode")

    def set_pd_gains(self, kp, kd):
        kp, kd = _host(kp, np.float32), _host(kd, np.float32)
        assert kp.shape == (self.nv,) and kd.shape == (self.nv,)
        check(self.L.rsb_set_pd_gains(self.handle, _hp(kp), _hp(kd)), "rsb_set_pd_gains")

    def set_pd_target(self, p_target=None, d_target=None):
        p, d = _host(p_target, np.float32), _host(d_target, np.float32)
        check(self.L.rsb_set_pd_target(self.handle, _hp(p), _hp(d), RSB_HOST), "rsb_set_pd_target")

    def set_pd_target_device(self, p_ptr=None, d_ptr=None):
        check(self.L.rsb_set_pd_target(self.handle, C.c_void_p(p_ptr) if p_ptr else None,
                                       C.c_void_p(d_ptr) if d_ptr else None, RSB_DEVICE), "rsb_set_pd_target")

    def set_generalized_force(self, tau):
        t = _host(tau, np.float32)
        check(self.L.rsb_set_generalized_force(self.handle, _hp(t), RSB_HOST), "rsb_set_generalized_force")

    def get_contacts(self):
        kmax = self.max_contacts()
        counts = np.empty(self.N, np.int32)
        con = np.zeros((self.N, kmax), CONTACT_DTYPE)
        check(self.L.rsb_get_contacts(self.handle, _hp(counts), _hp(con), RSB_HOST), "rsb_get_contacts")
        return counts, con

    def get_mass_matrix(self):
        M = np.empty((self.N, self.nv, self.nv), np.float32)
        check(self.L.rsb_get_mass_matrix(self.handle, _hp(M), RSB_HOST), "rsb_get_mass_matrix")
        return M

    def get_inverse_mass_matrix(self):
        out = np.zeros((self.N, self.model.nv, self.model.nv), np.float32)
        check(self.L.rsb_get_inverse_mass_matrix(self.handle, _hp(out), RSB_HOST), "rsb_get_inverse_mass_matrix")
        return out

    def get_nonlinearities(self):
        h = np.empty((self.N, self.nv), np.float32)
        check(self.L.rsb_get_nonlinearities(self.handle, _hp(h), RSB_HOST), "rsb_get_nonlinearities")
        return h

    def get_flags(self):
        f = np.empty(self.N, np.int32)
        check(self.L.rsb_get_flags(self.handle, _hp(f), RSB_HOST), "rsb_get_flags")
        return f

    def get_solver_iterations(self):
        f = np.empty(self.N, np.int32)
        check(self.L.rsb_get_solver_iterations(self.handle, _hp(f), RSB_HOST), "rsb_get_solver_iterations")
        return f

    # -- observation block for the vectorised env / RCCL gather --------------------------------------
    def obs_dim(self, n_force_slots):
        return self.L.rsb_obs_dim(self.handle, int(n_force_slots))

    def gather_obs(self, out_device_ptr, collision_indices):
        idx = _host(collision_indices, np.int32)
        n = 0 if idx is None else idx.shape[0]
        check(self.L.rsb_gather_obs(self.handle, C.c_void_p(out_device_ptr), _hp(idx), n, RSB_DEVICE), "rsb_gather_obs")

    def reset_terminated(self, allowed_collisions, gc0, gv0):
        """Reset envs that touch the terrain with anything but `allowed_collisions` (host arrays). Returns done [N]."""
        idx = _host(allowed_collisions, np.int32)
        g0, v0 = _host(gc0, np.float32), _host(gv0, np.float32)
        rows = 1 if g0.ndim == 1 or g0.shape[0] == 1 else g0.shape[0]
        done = np.zeros(self.N, np.uint8)
        check(self.L.rsb_reset_terminated(self.handle, _hp(idx), idx.shape[0], _hp(g0), _hp(v0), rows, _hp(done), RSB_HOST),
              "rsb_reset_terminated")
        return done

    def reset_terminated_device(self, allowed_collisions, gc0_ptr, gv0_ptr, rows, done_ptr=None):
        idx = _host(allowed_collisions, np.int32)
        check(self.L.rsb_reset_terminated(self.handle, _hp(idx), idx.shape[0], C.c_void_p(gc0_ptr), C.c_void_p(gv0_ptr),
                                          int(rows), C.c_void_p(done_ptr) if done_ptr else None, RSB_DEVICE),
              "rsb_reset_terminated")

    def device_ptr(self, field):
        return self.L.rsb_device_ptr(self.handle, int(field))

    def enable_timing(self, on=True):
        """False/0: off; True/1: one event pair (last_kernel_ms); n > 1: ring of n event pairs (read_kernel_ms)."""
        check(self.L.rsb_enable_timing(self.handle, int(on)), "rsb_enable_timing")

    def set_timing_stride(self, stride):
        check(self.L.rsb_set_timing_stride(self.handle, int(stride)), "rsb_set_timing_stride")

    def read_kernel_ms(self, n):
        """Durations (ms) of the last n step-kernel launches (oldest first); synchronises the stream."""
        out = np.zeros(int(n), np.float32)
        got = self.L.rsb_read_kernel_ms(self.handle, _hp(out), int(n))
        if got < 0:
            check(got, "rsb_read_kernel_ms")
        return out[:got]

    def control_step_plan(self, n_substeps, obs_ptr, force_collisions, allowed_collisions, gc0_ptr, gv0_ptr, rows):
        """Pre-marshalled rsb_control_step call: returns f(p_target_ptr) that enqueues one control step (PD targets ->
        n_substeps x integrate -> obs gather -> reset of terminated envs) with a single foreign call."""
        fidx = _host(force_collisions, np.int32)
        aidx = _host(allowed_collisions, np.int32) if allowed_collisions is not None else None
        fn, h = self.L.rsb_control_step, self.handle
        args = (int(n_substeps), C.c_void_p(obs_ptr) if obs_ptr else None, _hp(fidx), 0 if fidx is None else fidx.shape[0],
                _hp(aidx), 0 if aidx is None else aidx.shape[0],
                C.c_void_p(gc0_ptr) if gc0_ptr else None, C.c_void_p(gv0_ptr) if gv0_ptr else None, int(rows))
        keep = (fidx, aidx)

        def step(p_target_ptr, _keep=keep):
            st = fn(h, C.c_void_p(p_target_ptr), None, *args)
            if st != 0:
                check(st, "rsb_control_step")
        return step

    # -- resident control steps (rsb_control_steps / rsb_set_step_residency: K control steps per launch, the env blocks stay in LDS) ------------
    def set_step_residency(self, on=True):
        """rsb_control_steps and the closed-loop runs with an in-repo stage use ONE resident launch per call where the world's kernel class has a
        resident twin (residency_status says whether it does)."""
        check(self.L.rsb_set_step_residency(self.handle, int(bool(on))), "rsb_set_step_residency")

    def residency_status(self, stage=0):
        """True when a resident launch exists for this world as configured now (stage 0 open loop, 1 linear policy, 2 actor network)"""
        return bool(self.L.rsb_step_residency_status(self.handle, int(stage)))

    def residency_launches(self):
        return int(self.L.rsb_step_residency_launches(self.handle))

    # -- specialised step kernels (rsb_set_specialization: the model's dimensions and the world's switches as compile-time constants) ------------
    SPEC_OFF, SPEC_CACHED, SPEC_COMPILE = 0, 1, 2

    def set_specialization(self, mode):
        """"off" / "cached" (default: use a code object found in rsb_spec_dir()) / "compile" (a miss compiles it first, ~25 s once per key)"""
        m = {"off": 0, "cached": 1, "compile": 2}.get(mode, mode)
        check(self.L.rsb_set_specialization(self.handle, int(m)), "rsb_set_specialization")

    def specialization_status(self):
        """(mode, step launches that ran a specialised code object, step launches that ran an ahead-of-time class)"""
        a, b = C.c_longlong(0), C.c_longlong(0)
        mode = self.L.rsb_specialization_status(self.handle, C.byref(a), C.byref(b))
        return int(mode), int(a.value), int(b.value)

    def debug_resident_full_writes(self, on=True):
        check(self.L.rsb_debug_resident_full_writes(self.handle, int(bool(on))), "rsb_debug_resident_full_writes")

    def control_steps_plan(self, n_substeps, targets_ptr, period, obs_ptr, obs_step_stride, force_collisions, allowed_collisions, gc0_ptr, gv0_ptr, rows,
                           done_ptr=0, done_step_stride=0):
        """Pre-marshalled rsb_control_steps call: returns f(n_steps, first) that enqueues n_steps control steps of the open loop reading the PD-target
        slices (first + j) % period of the device bank at targets_ptr ([period, N, nq]); one resident launch with set_step_residency(True)."""
        fidx = _host(force_collisions, np.int32)
        aidx = _host(allowed_collisions, np.int32) if allowed_collisions is not None else None
        fn, h = self.L.rsb_control_steps, self.handle
        head = (C.c_void_p(targets_ptr), int(period))
        tail = (int(n_substeps), C.c_void_p(obs_ptr) if obs_ptr else None, int(obs_step_stride), _hp(fidx), 0 if fidx is None else fidx.shape[0],
                _hp(aidx), 0 if aidx is None else aidx.shape[0],
                C.c_void_p(gc0_ptr) if gc0_ptr else None, C.c_void_p(gv0_ptr) if gv0_ptr else None, int(rows),
                C.c_void_p(done_ptr) if done_ptr else None, int(done_step_stride))
        keep = (fidx, aidx)

        def steps(n_steps, first, _keep=keep):
            st = fn(h, int(n_steps), *head, int(first), *tail)
            if st != 0:
                check(st, "rsb_control_steps")
        return steps

    # -- peer-mapped obs exchange (rsb_obs_peer_*: no collective, no copy kernel; see include/rsb.h) ------------
    OBS_HANDLE_BYTES = 64

    def obs_peer_create(self, n_ranks, rank, force_collisions):
        """Allocates this rank's gathered buffer; returns its IPC handle (64 bytes; zeros when the system has no hipIpc)."""
        fidx = _host(force_collisions, np.int32)
        buf = C.create_string_buffer(self.OBS_HANDLE_BYTES)
        check(self.L.rsb_obs_peer_create(self.handle, int(n_ranks), int(rank), _hp(fidx), 0 if fidx is None else fidx.shape[0], buf),
              "rsb_obs_peer_create")
        return buf.raw

    def obs_peer_connect(self, handles):
        """handles: the n_ranks IPC handles in rank order (bytes of n_ranks * 64), e.g. from dist.all_gather_object."""
        check(self.L.rsb_obs_peer_connect(self.handle, bytes(handles)), "rsb_obs_peer_connect")

    def obs_peer_connect_ptrs(self, bases):
        """Within one process: base pointers (obs_peer_base()) of all ranks' worlds in rank order (own entry ignored)."""
        arr = (C.c_void_p * len(bases))(*[C.c_void_p(b) for b in bases])
        check(self.L.rsb_obs_peer_connect_ptrs(self.handle, arr), "rsb_obs_peer_connect_ptrs")

    def obs_peer_base(self):
        return self.L.rsb_obs_peer_base(self.handle)

    def obs_peer_wait(self):
        """Stream-side wait for every rank's rows of the last control step issued; returns the gathered block's device pointer."""
        out = C.c_void_p()
        check(self.L.rsb_obs_peer_wait(self.handle, C.byref(out)), "rsb_obs_peer_wait")
        return out.value

    def obs_peer_destroy(self):
        check(self.L.rsb_obs_peer_destroy(self.handle), "rsb_obs_peer_destroy")

    def last_kernel_ms(self):
        ms = C.c_float()
        check(self.L.rsb_last_kernel_ms(self.handle, C.byref(ms)), "rsb_last_kernel_ms")
        return ms.value

    # -- debug aid: one env's contact problem (G, c, lam) of the last sub-step -----------------------
    def debug_select_env(self, env):
        check(self.L.rsb_debug_select_env(self.handle, int(env)), "rsb_debug_select_env")

    def debug_contact_problem(self):
        K = RSB_MAX_CONTACTS
        G = np.zeros(9 * K * K, np.float32)
        c = np.zeros(3 * K, np.float32)
        lam = np.zeros(3 * K, np.float32)
        nc = C.c_int()
        check(self.L.rsb_debug_read_contact_problem(self.handle, C.byref(nc), _hp(G), _hp(c), _hp(lam)),
              "rsb_debug_read_contact_problem")
        n3 = 3 * nc.value
        return nc.value, G[:n3 * n3].reshape(n3, n3).copy(), c[:n3].copy(), lam[:n3].copy()

    def debug_phase_cycles(self, enable=True, read=True):
        out = np.zeros(16, np.int64)
        check(self.L.rsb_debug_phase_cycles(self.handle, 1 if enable else 0, _hp(out) if read else None),
              "rsb_debug_phase_cycles")
        return out

    def debug_wave_profile(self):
        nb = (self.N * self.lanes_per_env() + 63) // 64
        out = np.zeros((nb, 16), np.int64)
        check(self.L.rsb_debug_wave_profile(self.handle, _hp(out), nb), "rsb_debug_wave_profile")
        return out
