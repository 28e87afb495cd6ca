"""ctypes binding of the C-ABI declared in include/rsb.h (plumbing only — no physics here).

The shared library `raisimlib_amd/lib/librsb.so` is built in-tree by `__graft_entry__.build()`
(hipcc, gfx950).  Loading fails loudly if it is missing: there is no Python or CPU fallback.
"""
import ctypes as C
import os

RSB_MAX_BODIES = 64
RSB_MAX_DOF = 6 + RSB_MAX_BODIES - 1
RSB_MAX_COLLISIONS = 64
RSB_MAX_CONTACTS = 16
RSB_NAME_LEN = 48

RSB_HOST, RSB_DEVICE = 0, 1
RSB_CONTACT_SELF_A, RSB_CONTACT_SELF_B, RSB_CONTACT_SECOND, RSB_CONTACT_CAPSULE = 0x10000, 0x20000, 0x40000, 0x80000
RSB_FORCE_AND_TORQUE, RSB_PD_PLUS_FEEDFORWARD_TORQUE = 0, 1
(RSB_F_GC, RSB_F_GV, RSB_F_PTARGET, RSB_F_DTARGET, RSB_F_TAU_FF, RSB_F_CONTACT_COUNT, RSB_F_CONTACTS,
 RSB_F_FLAGS, RSB_F_GENERALIZED_FORCE) = range(9)

_B, _S = RSB_MAX_BODIES, RSB_MAX_COLLISIONS


class ModelBlob(C.Structure):
    _fields_ = [
        ("nb", C.c_int32), ("nq", C.c_int32), ("nv", C.c_int32), ("ncol", C.c_int32), ("depth", C.c_int32),
        ("fixed_base", C.c_int32),
        ("parent", C.c_int32 * _B), ("level", C.c_int32 * _B), ("jtype", C.c_int32 * _B),
        ("axis", (C.c_double * 3) * _B), ("ptree", (C.c_double * 3) * _B), ("rtree", (C.c_double * 9) * _B),
        ("mass", C.c_double * _B), ("com", (C.c_double * 3) * _B), ("inertia", (C.c_double * 6) * _B),
        ("armature", C.c_double * _B), ("damping", C.c_double * _B),
        ("q_lower", C.c_double * _B), ("q_upper", C.c_double * _B), ("effort", C.c_double * _B),
        ("col_body", C.c_int32 * _S), ("col_pos", (C.c_double * 3) * _S), ("col_radius", C.c_double * _S),
        ("body_name", (C.c_char * RSB_NAME_LEN) * _B), ("joint_name", (C.c_char * RSB_NAME_LEN) * _B),
        ("col_name", (C.c_char * RSB_NAME_LEN) * _S),
        ("col_axis", (C.c_double * 3) * _S), ("col_rim", C.c_double * _S), ("col_material", (C.c_char * RSB_NAME_LEN) * _S),
        ("col_capsule", C.c_int32 * _S),
    ]


class Contact(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("normal", C.c_float * 3), ("impulse", C.c_float * 3),
                ("depth", C.c_float), ("body", C.c_int32), ("collision", C.c_int32)]


class TerrainProperties(C.Structure):
    """rsb_terrain_properties (raisim::TerrainProperties field meaning)"""
    _fields_ = [("frequency", C.c_double), ("z_scale", C.c_double), ("x_size", C.c_double), ("y_size", C.c_double),
                ("x_samples", C.c_int32), ("y_samples", C.c_int32), ("fractal_octaves", C.c_int32), ("seed", C.c_uint32),
                ("fractal_lacunarity", C.c_double), ("fractal_gain", C.c_double), ("step_size", C.c_double),
                ("height_offset", C.c_double)]


class EnvConfig(C.Structure):
    """rsb_env_config"""
    _fields_ = [("n_substeps", C.c_int32), ("action_std", C.c_float), ("forward_vel_coeff", C.c_float),
                ("forward_vel_clip", C.c_float), ("torque_coeff", C.c_float), ("terminal_reward", C.c_float),
                ("n_foot", C.c_int32), ("foot_collisions", C.c_int32 * RSB_MAX_COLLISIONS)]


class LinearPolicy(C.Structure):
    """rsb_linear_policy (include/rsb_pipeline.h): device pointers"""
    _fields_ = [("W", C.c_void_p), ("bias", C.c_void_p), ("noise", C.c_void_p), ("noise_period", C.c_int32), ("clip", C.c_float),
                ("rollout_ob", C.c_void_p), ("rollout_act", C.c_void_p), ("rollout_reward", C.c_void_p), ("rollout_done", C.c_void_p)]


class MlpPolicy(C.Structure):
    """rsb_mlp_policy (include/rsb_pipeline.h): device pointers"""
    _fields_ = [("n_layers", C.c_int32), ("dims", C.c_int32 * 5), ("Wt", C.c_void_p * 4), ("bias", C.c_void_p * 4), ("activation", C.c_int32), ("leaky_slope", C.c_float),
                ("ob_mean", C.c_void_p), ("ob_inv_std", C.c_void_p), ("ob_clip", C.c_float), ("noise", C.c_void_p), ("noise_period", C.c_int32), ("clip", C.c_float),
                ("rollout_ob", C.c_void_p), ("rollout_act", C.c_void_p), ("rollout_reward", C.c_void_p), ("rollout_done", C.c_void_p)]


RSB_E_PIPELINE = -7

LIB_PATH = os.environ.get("RSB_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "librsb.so")   # RSB_LIB_PATH: kernel experiments built by build.build(extra_flags=...)

# name -> (restype, argtypes); mirrors include/rsb.h + include/rsb_pipeline.h one-to-one (tests check they stay in sync)
_VP, _I, _D, _FP, _CP = C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_char_p
PROTOTYPES = {
    "rsb_last_error": (C.c_char_p, []),
    "rsb_version": (C.c_char_p, []),
    "rsb_source_hash": (C.c_char_p, []),
    "rsb_model_from_urdf_file": (_I, [_CP, C.POINTER(_VP)]),
    "rsb_model_from_urdf_string": (_I, [_CP, C.POINTER(_VP)]),
    "rsb_model_from_urdf_file_sampled": (_I, [_CP, _D, C.POINTER(_VP)]),
    "rsb_model_from_urdf_string_sampled": (_I, [_CP, _D, C.POINTER(_VP)]),
    "rsb_model_from_blob": (_I, [C.POINTER(ModelBlob), C.POINTER(_VP)]),
    "rsb_set_mesh_point_budget": (_I, [_I]),
    "rsb_model_destroy": (_I, [_VP]),
    "rsb_model_get_blob": (_I, [_VP, C.POINTER(ModelBlob)]),
    "rsb_model_body_index": (_I, [_VP, _CP]),
    "rsb_model_joint_index": (_I, [_VP, _CP]),
    "rsb_model_total_mass": (_D, [_VP]),
    "rsb_model_collision_material": (C.c_char_p, [_VP, _I]),
    "rsb_model_skipped_collisions": (_I, [_VP]),
    "rsb_device_count": (_I, []),
    "rsb_create": (_I, [_VP, _I, _I, C.POINTER(_VP)]),
    "rsb_destroy": (_I, [_VP]),
    "rsb_set_stream": (_I, [_VP, _VP]),
    "rsb_get_stream": (_VP, [_VP]),
    "rsb_synchronize": (_I, [_VP]),
    "rsb_num_envs": (_I, [_VP]),
    "rsb_dims": (_I, [_VP] + [C.POINTER(C.c_int)] * 5),
    "rsb_set_timestep": (_I, [_VP, _D]),
    "rsb_get_timestep": (_D, [_VP]),
    "rsb_get_world_time": (_D, [_VP]),
    "rsb_set_gravity": (_I, [_VP, C.POINTER(C.c_double)]),
    "rsb_set_erp": (_I, [_VP, _D]),
    "rsb_set_friction": (_I, [_VP, _D]),
    "rsb_set_material": (_I, [_VP, _D, _D, _D]),
    "rsb_set_collision_materials": (_I, [_VP, _VP, _VP, _VP]),
    "rsb_set_self_collision": (_I, [_VP, _I]),
    "rsb_ignore_collision_between": (_I, [_VP, _I, _I]),
    "rsb_self_collision_pairs": (_I, [_VP, _FP, _I]),
    "rsb_set_self_collision_materials": (_I, [_VP, _VP, _VP, _VP]),
    "rsb_set_contact_solver_param": (_I, [_VP, _D, _D, _D, _I, _D]),
    "rsb_set_solver_stagnation_exit": (_I, [_VP, _I, _D]),
    "rsb_set_solver_friction_lag": (_I, [_VP, _I, _I, _D]),
    "rsb_set_solver_multi_contact": (_I, [_VP, _I, _I, _I, _I]),
    "rsb_set_solver_anderson": (_I, [_VP, _I, _D]),
    "rsb_set_heightmap_contacts": (_I, [_VP, _I, _D]),
    "rsb_set_capsule_contacts": (_I, [_VP, _I]),
    "rsb_set_step_pipelining": (_I, [_VP, _I]),
    "rsb_step_pipelining_enabled": (_I, [_VP]),
    "rsb_step_pipeline_publish": (_I, [_VP, _VP]),
    "rsb_step_pipeline_wait_event": (_I, [_VP, _VP]),
    "rsb_step_pipelining_stats": (_I, [_VP, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "rsb_step_pipeline_join": (_I, [_VP]),
    "rsb_step_pipeline_fault": (_I, [_VP, C.POINTER(_I), C.POINTER(_I)]),
    "rsb_debug_pipeline_fault": (_I, [_VP, _I]),
    "rsb_debug_pipeline_wait_stats": (_I, [_VP, C.POINTER(_D), C.POINTER(_D)]),
    "rsb_closed_loop_run": (_I, [_VP, _I, _VP, _VP]),
    "rsb_closed_loop_run_linear": (_I, [_VP, _I, C.POINTER(LinearPolicy)]),
    "rsb_closed_loop_run_mlp": (_I, [_VP, _I, C.POINTER(MlpPolicy)]),
    "rsb_closed_loop_buffers": (_I, [_VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP)]),
    "rsb_closed_loop_set_stage_grid": (_I, [_VP, _I]),
    "rsb_set_slip_rule": (_I, [_VP, _I]),
    "rsb_set_integration_scheme": (_I, [_VP, _I]),
    "rsb_set_early_termination": (_I, [_VP, _I]),
    "rsb_set_solver_warm_start": (_I, [_VP, _I]),
    "rsb_set_max_contacts": (_I, [_VP, _I]),
    "rsb_set_lanes_per_env": (_I, [_VP, _I]),
    "rsb_get_lanes_per_env": (_I, [_VP]),
    "rsb_set_ground": (_I, [_VP, _D]),
    "rsb_set_heightmap": (_I, [_VP, _I, _I, _D, _D, _D, _D, _FP]),
    "rsb_set_heightmaps": (_I, [_VP, _I, _I, _I, _D, _D, _D, _D, _FP, _FP]),
    "rsb_heightmap_png_size": (_I, [_CP, C.POINTER(_I), C.POINTER(_I)]),
    "rsb_heightmap_png_read": (_I, [_CP, _D, _D, _FP, _I]),
    "rsb_heightmap_perlin": (_I, [C.POINTER(TerrainProperties), _FP]),
    "rsb_heightmap_text_size": (_I, [_CP, C.POINTER(_I), C.POINTER(_I), C.POINTER(_D), C.POINTER(_D)]),
    "rsb_heightmap_text_read": (_I, [_CP, _FP, _I]),
    "rsb_set_state": (_I, [_VP, _FP, _FP, _FP, _I]),
    "rsb_get_state": (_I, [_VP, _FP, _FP, _I]),
    "rsb_set_env_row": (_I, [_VP, _I, _I, _FP]),
    "rsb_get_env_row": (_I, [_VP, _I, _I, _FP]),
    "rsb_get_field": (_I, [_VP, _I, _FP, _I]),
    "rsb_enable_generalized_force_output": (_I, [_VP, _I]),
    "rsb_set_control_mode": (_I, [_VP, _I]),
    "rsb_set_pd_gains": (_I, [_VP, _FP, _FP]),
    "rsb_set_pd_target": (_I, [_VP, _FP, _FP, _I]),
    "rsb_set_generalized_force": (_I, [_VP, _FP, _I]),
    "rsb_integrate": (_I, [_VP, _I]),
    "rsb_integrate_masked": (_I, [_VP, _I, _VP, _I]),
    "rsb_view_exchange": (_I, [_VP, _VP]),
    "rsb_host_alloc": (_I, [C.c_size_t, _VP]),
    "rsb_host_free": (_I, [_VP]),
    "rsb_device_alloc": (_I, [_VP, C.c_size_t, _VP]),
    "rsb_device_free": (_I, [_VP, _VP]),
    "rsb_device_copy": (_I, [_VP, _VP, _VP, C.c_size_t, _I]),
    "rsb_set_done_output": (_I, [_VP, _VP]),
    "rsb_comm_rccl_version": (_I, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rsb_comm_get_unique_id": (_I, [C.c_char_p]),
    "rsb_comm_init": (_I, [_VP, _I, _I, C.c_char_p]),
    "rsb_comm_destroy": (_I, [_VP]),
    "rsb_allgather_obs": (_I, [_VP, _VP, _I, _FP, _I]),
    "rsb_obs_peer_create": (_I, [_VP, _I, _I, _FP, _I, C.c_char_p]),
    "rsb_obs_peer_connect": (_I, [_VP, C.c_char_p]),
    "rsb_obs_peer_connect_ptrs": (_I, [_VP, C.POINTER(_VP)]),
    "rsb_obs_peer_base": (_VP, [_VP]),
    "rsb_obs_peer_wait": (_I, [_VP, C.POINTER(_VP)]),
    "rsb_obs_peer_destroy": (_I, [_VP]),
    "rsb_integrate1": (_I, [_VP]),
    "rsb_integrate2": (_I, [_VP]),
    "rsb_get_contacts": (_I, [_VP, _FP, _FP, _I]),
    "rsb_get_mass_matrix": (_I, [_VP, _FP, _I]),
    "rsb_get_nonlinearities": (_I, [_VP, _FP, _I]),
    "rsb_get_inverse_mass_matrix": (_I, [_VP, _FP, _I]),
    "rsb_get_flags": (_I, [_VP, _FP, _I]),
    "rsb_get_solver_iterations": (_I, [_VP, _FP, _I]),
    "rsb_obs_dim": (_I, [_VP, _I]),
    "rsb_gather_obs": (_I, [_VP, _FP, _FP, _I, _I]),
    "rsb_reset_terminated": (_I, [_VP, _FP, _I, _FP, _FP, _I, _FP, _I]),
    "rsb_env_configure": (_I, [_VP, C.POINTER(EnvConfig), _FP, _FP, _FP]),
    "rsb_env_dims": (_I, [_VP, C.POINTER(_I), C.POINTER(_I)]),
    "rsb_env_set_reset_states": (_I, [_VP, _FP, _FP, _I]),
    "rsb_env_reset": (_I, [_VP]),
    "rsb_env_observe": (_I, [_VP, _FP, _I]),
    "rsb_env_step": (_I, [_VP, _FP, _FP, _FP, _FP, _I]),
    "rsb_device_ptr": (_VP, [_VP, _I]),
    "rsb_last_kernel_ms": (_I, [_VP, C.POINTER(C.c_float)]),
    "rsb_enable_timing": (_I, [_VP, _I]),
    "rsb_set_timing_stride": (_I, [_VP, _I]),
    "rsb_read_kernel_ms": (_I, [_VP, _FP, _I]),
    "rsb_control_step": (_I, [_VP, _FP, _FP, _I, _FP, _FP, _I, _FP, _I, _FP, _FP, _I]),
    "rsb_control_steps": (_I, [_VP, _I, _FP, _I, C.c_longlong, _I, _FP, C.c_longlong, _FP, _I, _FP, _I, _FP, _FP, _I, _VP, C.c_longlong]),
    "rsb_set_step_residency": (_I, [_VP, _I]),
    "rsb_step_residency_enabled": (_I, [_VP]),
    "rsb_step_residency_status": (_I, [_VP, _I]),
    "rsb_step_residency_launches": (C.c_longlong, [_VP]),
    "rsb_debug_resident_full_writes": (_I, [_VP, _I]),
    "rsb_model_lds_bytes": (_I, [_VP, _I, _I, _I]),
    "rsb_set_specialization": (_I, [_VP, _I]),
    "rsb_specialization_status": (_I, [_VP, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "rsb_spec_dir": (C.c_char_p, []),
    "rsb_spec_compile": (_I, [C.c_char_p]),
    "rsb_spec_file_name": (_I, [C.c_char_p, C.c_char_p, _I]),
    "rsb_debug_view_profile": (_I, [_VP, C.POINTER(C.c_longlong), _I]),
    "rsb_debug_select_env": (_I, [_VP, _I]),
    "rsb_debug_phase_cycles": (_I, [_VP, _I, _FP]),
    "rsb_debug_wave_profile": (_I, [_VP, _FP, _I]),
    "rsb_debug_read_contact_problem": (_I, [_VP, C.POINTER(C.c_int), _FP, _FP, _FP]),
}

_lib = None


def lib():
    """Load librsb.so once; raise (never fall back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). raisimlib_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError here = header/library drift; fail loudly
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class RsbError(RuntimeError):
    pass


def check(status, what=""):
    if status != 0:
        msg = lib().rsb_last_error()
        raise RsbError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")
