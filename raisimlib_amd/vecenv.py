"""Device-resident vectorised env over the C-ABI's rsb_env_* entry points (plumbing only).

Mirrors raisimGymTorch's Python wrapper `RaisimGymVecEnv` [RECALL raisimGymTorch/env/RaisimGymVecEnv.py, absent from
/root/reference]: `reset()`, `observe()`, `step(action) -> (reward, done)`, `num_obs`, `num_acts`, `num_envs`, with
rsg_anymal's task semantics computed on the GPU (see include/rsb.h, "device-resident vectorised env").  Actions,
observations, rewards and dones are torch CUDA tensors when torch tensors are passed (zero-copy, on the caller's
stream) and numpy arrays otherwise (staged through PCIe by the library).  All device work is enqueued on the HIP stream
given as `stream` (a raw stream handle), by default torch's current stream at construction time.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import RSB_DEVICE, RSB_HOST, check
from .world import BatchedWorld, Model, _hp


class VecEnv:
    def __init__(self, model, num_envs, device=0, simulation_dt=0.0025, control_dt=0.01, action_std=0.3, p_gain=50.0,
                 d_gain=0.2, forward_vel_coeff=0.3, forward_vel_clip=4.0, torque_coeff=-4e-5, terminal_reward=-10.0,
                 gc_init=None, foot_suffix="_foot", stream=None, early_termination=False):
        self.model = model if isinstance(model, Model) else Model(urdf_path=model)
        self.world = BatchedWorld(self.model, num_envs, device=device)
        w, m = self.world, self.model
        if stream is None:
            # torch tensors are produced / consumed on torch's current stream: run the world on it, otherwise the
            # world's own (non-blocking) stream would race with the caller's tensor ops
            try:
                import torch
                if torch.cuda.is_available():
                    stream = torch.cuda.current_stream(device).cuda_stream
            except ImportError:
                pass
        if stream is not None:
            w.set_stream(stream)
        self.num_envs, self.nq, self.nv = num_envs, m.nq, m.nv
        self.device_index = int(device)
        self.num_acts = m.nv - 6
        self.num_obs = 10 + 2 * self.num_acts
        w.set_time_step(simulation_dt)
        if early_termination:      # NOT upstream's rule (last sub-step only): see rsb_set_early_termination in include/rsb.h
            w.set_early_termination(True)
        kp = np.zeros(m.nv, np.float32); kd = np.zeros(m.nv, np.float32)
        kp[6:] = p_gain; kd[6:] = d_gain
        w.set_pd_gains(kp, kd)
        if gc_init is None:
            gc_init = np.zeros(m.nq, np.float32); gc_init[2] = 0.6; gc_init[3] = 1.0
        self.gc_init = np.ascontiguousarray(gc_init, np.float32)
        self.gv_init = np.zeros(m.nv, np.float32)
        self.action_mean = np.ascontiguousarray(self.gc_init[7:], np.float32)
        cfg = _capi.EnvConfig()
        cfg.n_substeps = int(round(control_dt / simulation_dt))
        cfg.action_std, cfg.forward_vel_coeff, cfg.forward_vel_clip = action_std, forward_vel_coeff, forward_vel_clip
        cfg.torque_coeff, cfg.terminal_reward = torque_coeff, terminal_reward
        feet = m.collision_indices(foot_suffix)
        cfg.n_foot = len(feet)
        for i, f in enumerate(feet):
            cfg.foot_collisions[i] = f
        self.cfg = cfg
        w.set_pd_target(np.tile(np.r_[np.zeros(3), 1, np.zeros(m.nq - 4)], (num_envs, 1)), np.zeros((num_envs, m.nv)))
        check(w.L.rsb_env_configure(w.handle, C.byref(cfg), _hp(self.action_mean), _hp(self.gc_init), _hp(self.gv_init)),
              "rsb_env_configure")
        self.ob_mean = self.ob_var = None      # running observation statistics (created on the env's device at first use)
        self.ob_count = 1e-4
        self.reset()

    def _check_tensor(self, t, shape, dtype, name):
        """A wrongly shaped / typed / placed tensor would make the kernels read or write out of bounds on the device."""
        if not (t.is_cuda and t.is_contiguous() and t.dtype == dtype and tuple(t.shape) == tuple(shape)
                and t.device.index == self.device_index):
            raise ValueError(f"VecEnv: {name} must be a contiguous {dtype} CUDA tensor of shape {tuple(shape)} on cuda:{self.device_index}, "
                             f"got {t.dtype} {tuple(t.shape)} on {t.device}")

    def _stats(self, device=None):
        import torch
        if self.ob_mean is None:
            dev = device if device is not None else torch.device("cuda", self.device_index)
            self.ob_mean = torch.zeros(self.num_obs, dtype=torch.float32, device=dev)
            self.ob_var = torch.ones(self.num_obs, dtype=torch.float32, device=dev)

    # a torch tensor travels as its device pointer, anything else as a host array
    @staticmethod
    def _is_torch(x):
        return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")

    def reset(self):
        check(self.world.L.rsb_env_reset(self.world.handle), "rsb_env_reset")

    def observe(self, out=None):
        w = self.world
        if out is not None and self._is_torch(out):
            import torch
            self._check_tensor(out, (self.num_envs, self.num_obs), torch.float32, "out")
            check(w.L.rsb_env_observe(w.handle, C.c_void_p(out.data_ptr()), RSB_DEVICE), "rsb_env_observe")
            return out
        ob = np.zeros((self.num_envs, self.num_obs), np.float32) if out is None else out
        if not (isinstance(ob, np.ndarray) and ob.dtype == np.float32 and ob.flags.c_contiguous and ob.shape == (self.num_envs, self.num_obs)):
            raise ValueError("VecEnv.observe: out must be a C-contiguous float32 array [num_envs, num_obs]")
        check(w.L.rsb_env_observe(w.handle, _hp(ob), RSB_HOST), "rsb_env_observe")
        return ob

    def step(self, action, reward=None, done=None, ob_next=None):
        """action [num_envs, num_acts] -> (reward [num_envs] float32, done [num_envs] uint8).  ob_next (optional,
        [num_envs, num_obs]) receives the observation the next step starts from (after the resets) from the same
        launch, which saves the separate observe() call."""
        w = self.world
        if self._is_torch(action):
            import torch
            self._check_tensor(action, (self.num_envs, self.num_acts), torch.float32, "action")
            reward = torch.empty(self.num_envs, dtype=torch.float32, device=action.device) if reward is None else reward
            done = torch.empty(self.num_envs, dtype=torch.uint8, device=action.device) if done is None else done
            self._check_tensor(reward, (self.num_envs,), torch.float32, "reward")
            self._check_tensor(done, (self.num_envs,), torch.uint8, "done")
            obp = None
            if ob_next is not None:
                self._check_tensor(ob_next, (self.num_envs, self.num_obs), torch.float32, "ob_next")
                obp = C.c_void_p(ob_next.data_ptr())
            check(w.L.rsb_env_step(w.handle, C.c_void_p(action.data_ptr()), C.c_void_p(reward.data_ptr()),
                                   C.c_void_p(done.data_ptr()), obp, RSB_DEVICE), "rsb_env_step")
            return reward, done
        a = np.ascontiguousarray(action, np.float32)
        assert a.shape == (self.num_envs, self.num_acts)
        reward = np.zeros(self.num_envs, np.float32) if reward is None else reward
        done = np.zeros(self.num_envs, np.uint8) if done is None else done
        for arr, dt, shp, nm in ((reward, np.float32, (self.num_envs,), "reward"), (done, np.uint8, (self.num_envs,), "done"),
                                 (ob_next, np.float32, (self.num_envs, self.num_obs), "ob_next")):
            if arr is not None and not (isinstance(arr, np.ndarray) and arr.dtype == dt and arr.flags.c_contiguous and arr.shape == shp):
                raise ValueError(f"VecEnv.step: {nm} must be a C-contiguous {np.dtype(dt).name} array of shape {shp}")
        check(w.L.rsb_env_step(w.handle, _hp(a), _hp(reward), _hp(done), _hp(ob_next), RSB_HOST), "rsb_env_step")
        return reward, done

    # -- the closed loop on the device (include/rsb_pipeline.h): K steps with a policy between them, handed over env block by env block ----
    def set_reset_states(self, gc0=None, gv0=None):
        """per-env reset states [num_envs, nq] / [num_envs, nv] (numpy; None, None: back to the single gc_init / gv_init)"""
        w = self.world
        if gc0 is None:
            check(w.L.rsb_env_set_reset_states(w.handle, None, None, RSB_HOST), "rsb_env_set_reset_states")
            return
        a, b = np.ascontiguousarray(gc0, np.float32), np.ascontiguousarray(gv0, np.float32)
        assert a.shape == (self.num_envs, self.nq) and b.shape == (self.num_envs, self.nv)
        check(w.L.rsb_env_set_reset_states(w.handle, _hp(a), _hp(b), RSB_HOST), "rsb_env_set_reset_states")

    def closed_loop_buffers(self):
        """device pointers (ob [N, num_obs], act [N, num_acts], reward [N], done [N]) of the world's own env-task buffers"""
        w = self.world
        ps = [C.c_void_p(0) for _ in range(4)]
        check(w.L.rsb_closed_loop_buffers(w.handle, *[C.byref(p) for p in ps]), "rsb_closed_loop_buffers")
        return tuple(int(p.value or 0) for p in ps)

    def set_stage_grid(self, workgroups):
        check(self.world.L.rsb_closed_loop_set_stage_grid(self.world.handle, int(workgroups)), "rsb_closed_loop_set_stage_grid")

    def rollout_linear(self, n_steps, W, bias=None, noise=None, clip=0.0, rollout=None):
        """n_steps control steps with the in-repo reference stage in the loop: action = clip(bias + W ob + noise[step % period])  (torch CUDA
        tensors: W [num_acts, num_obs], bias [num_acts], noise [period, num_envs, num_acts]).  `rollout` (optional): dict of torch CUDA tensors
        ob [K + 1, N, num_obs], act [K, N, num_acts], reward [K, N] float32, done [K, N] uint8, filled on the device.  Nothing synchronises;
        with world.set_step_pipelining(True) the steps overlap at env-block granularity, bit-identical to lock-step."""
        import torch
        w = self.world
        K, N = int(n_steps), self.num_envs
        self._check_tensor(W, (self.num_acts, self.num_obs), torch.float32, "W")
        p = _capi.LinearPolicy()
        p.W = W.data_ptr()
        if bias is not None:
            self._check_tensor(bias, (self.num_acts,), torch.float32, "bias")
            p.bias = bias.data_ptr()
        if noise is not None:
            self._check_tensor(noise, (noise.shape[0], N, self.num_acts), torch.float32, "noise")
            p.noise, p.noise_period = noise.data_ptr(), int(noise.shape[0])
        p.clip = float(clip)
        if rollout is not None:
            for key, shape, dt in (("ob", (K + 1, N, self.num_obs), torch.float32), ("act", (K, N, self.num_acts), torch.float32),
                                   ("reward", (K, N), torch.float32), ("done", (K, N), torch.uint8)):
                if rollout.get(key) is not None:
                    self._check_tensor(rollout[key], shape, dt, f"rollout[{key}]")
                    setattr(p, "rollout_" + key, rollout[key].data_ptr())
        self._keep = (W, bias, noise, rollout)       # the launches are asynchronous: keep the tensors alive until the next call
        check(w.L.rsb_closed_loop_run_linear(w.handle, K, C.byref(p)), "rsb_closed_loop_run_linear")

    def rollout_mlp(self, n_steps, layers, activation="leaky_relu", leaky_slope=0.01, ob_mean=None, ob_var=None, ob_clip=10.0, eps=1e-8,
                    noise=None, clip=0.0, rollout=None):
        """n_steps control steps with the in-repo MLP stage in the loop (rsb_closed_loop_run_mlp): the actor of a raisimGymTorch-style PPO run.
        `layers`: [(weight [out, in], bias [out] or None), ...] torch CUDA tensors in torch.nn.Linear's layout (e.g. [(l.weight, l.bias) for l in
        actor if isinstance(l, torch.nn.Linear)]); hidden layers use `activation` ("tanh" | "relu" | "leaky_relu"), the last one is linear.
        ob_mean / ob_var [num_obs]: frozen running statistics, the network sees clamp((ob - mean) / sqrt(var + eps), +-ob_clip) (this class's
        observe_normalized); None: the raw observation.  noise / clip / rollout as rollout_linear.  Nothing synchronises."""
        import torch
        w = self.world
        K, N = int(n_steps), self.num_envs
        p = _capi.MlpPolicy()
        p.n_layers = len(layers)
        if not 1 <= len(layers) <= 4:
            raise ValueError("rollout_mlp: 1 .. 4 layers")
        keep = []
        dims = [self.num_obs]
        for l, (W, b) in enumerate(layers):
            self._check_tensor(W, (W.shape[0], dims[-1]), torch.float32, f"layers[{l}].weight")
            Wt = W.detach().t().contiguous()                # [in, out]: a row = the weights of one input for all units (one coalesced load)
            keep.append(Wt)
            p.Wt[l] = Wt.data_ptr()
            if b is not None:
                self._check_tensor(b, (W.shape[0],), torch.float32, f"layers[{l}].bias")
                bb = b.detach().contiguous()
                keep.append(bb)
                p.bias[l] = bb.data_ptr()
            dims.append(int(W.shape[0]))
        if dims[-1] != self.num_acts:
            raise ValueError(f"rollout_mlp: the last layer must have {self.num_acts} outputs")
        for i, d in enumerate(dims):
            p.dims[i] = d
        p.activation = {"tanh": 0, "relu": 1, "leaky_relu": 2}[activation]
        p.leaky_slope = float(leaky_slope)
        if ob_mean is not None:
            self._check_tensor(ob_mean, (self.num_obs,), torch.float32, "ob_mean")
            self._check_tensor(ob_var, (self.num_obs,), torch.float32, "ob_var")
            inv = torch.rsqrt(ob_var + eps).contiguous()
            keep += [ob_mean, inv]
            p.ob_mean, p.ob_inv_std, p.ob_clip = ob_mean.data_ptr(), inv.data_ptr(), float(ob_clip)
        if noise is not None:
            self._check_tensor(noise, (noise.shape[0], N, self.num_acts), torch.float32, "noise")
            p.noise, p.noise_period = noise.data_ptr(), int(noise.shape[0])
        p.clip = float(clip)
        if rollout is not None:
            for key, shape, dt in (("ob", (K + 1, N, self.num_obs), torch.float32), ("act", (K, N, self.num_acts), torch.float32),
                                   ("reward", (K, N), torch.float32), ("done", (K, N), torch.uint8)):
                if rollout.get(key) is not None:
                    self._check_tensor(rollout[key], shape, dt, f"rollout[{key}]")
                    setattr(p, "rollout_" + key, rollout[key].data_ptr())
        self._keep = (keep, noise, rollout)          # the launches are asynchronous: keep the tensors alive until the next call
        check(w.L.rsb_closed_loop_run_mlp(w.handle, K, C.byref(p)), "rsb_closed_loop_run_mlp")

    # -- running observation statistics (RaisimGymVecEnv's normalize_ob / RunningMeanStd [RECALL]) -------------------
    def observe_normalized(self, out, update_statistics=True, clip=10.0, eps=1e-8):
        """Observation tensor [num_envs, num_obs] (torch CUDA), normalised in place with running mean / variance kept on
        the device (batched Welford update, as upstream's RunningMeanStd); returns `out`."""
        import torch
        self.observe(out)
        self._stats(out.device)
        if update_statistics:
            bm, bv, n = out.mean(0), out.var(0, unbiased=False), float(out.shape[0])
            delta, tot = bm - self.ob_mean, self.ob_count + n
            self.ob_mean = self.ob_mean + delta * (n / tot)
            self.ob_var = (self.ob_var * self.ob_count + bv * n + delta * delta * (self.ob_count * n / tot)) / tot
            self.ob_count = tot
        out.sub_(self.ob_mean).div_(torch.sqrt(self.ob_var + eps)).clamp_(-clip, clip)
        return out

    # -- the rest of RaisimGymVecEnv's surface [RECALL]: scaling files, seeds, no-op hooks ------------------------------
    def save_scaling(self, dir_name, iteration):
        """mean<iteration>.csv / var<iteration>.csv of the running observation statistics (upstream file names)."""
        import os
        self._stats()
        np.savetxt(os.path.join(dir_name, f"mean{iteration}.csv"), self.ob_mean.cpu().numpy())
        np.savetxt(os.path.join(dir_name, f"var{iteration}.csv"), self.ob_var.cpu().numpy())

    def load_scaling(self, dir_name, iteration, count=1e5, device=None):
        import os
        import torch
        device = torch.device("cuda", self.device_index) if device is None else device
        self.ob_mean = torch.from_numpy(np.loadtxt(os.path.join(dir_name, f"mean{iteration}.csv")).astype(np.float32)).to(device)
        self.ob_var = torch.from_numpy(np.loadtxt(os.path.join(dir_name, f"var{iteration}.csv")).astype(np.float32)).to(device)
        self.ob_count = float(count)

    def seed(self, seed=None):
        """The simulation itself is deterministic; randomness lives in the caller's actions."""
        return seed

    def curriculum_callback(self):
        pass

    def turn_on_visualization(self):
        pass

    def turn_off_visualization(self):
        pass

    def close(self):
        self.world.close()
