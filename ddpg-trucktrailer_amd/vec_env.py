"""TruckTrailerVecEnv: N truck-trailer backing envs stepped by one HIP kernel (libttenv.so).

The vector form of the reference's `Truck_trailer_Env_2` (truck_trailer_sim/simv2.py:20-545):
same reset / step / observe / pose-override surface, but every array is a torch tensor resident
on the GPU and one call advances all N envs.  torch is used for device memory and streams only;
all arithmetic happens in ddpg-trucktrailer_amd/csrc/ttenv.hip behind the C ABI of include/ttenv.h.
"""
import ctypes as C

import torch

from ddpg_trucktrailer_amd import _lib as L


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class TruckTrailerVecEnv:
    """N independent envs on one GPU.

    reset(seed, mask)            -> obs [N,23] f32            (simv2.py:459-498)
    step(action, auto_reset)     -> obs, reward, done, info   (simv2.py:499-545)
    set_pose / set_attrs / set_state / set_max_steps          (DDPG/test.py:96-115 pattern)
    observe(steering)            -> obs                       (simv2.py:103-181)
    """
    observation_dim = L.OBS_DIM

    def __init__(self, n_envs, device=None, variant=0, params=None):
        if not torch.cuda.is_available():
            raise RuntimeError("TruckTrailerVecEnv needs a GPU: the env step is a HIP kernel, there is no CPU path")
        self.lib = L.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("device must be a cuda (HIP) device")
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        self.n_envs = int(n_envs)
        self.params = params if params is not None else L.default_params(variant)
        self.variant = int(self.params.variant)
        h = C.c_void_p()
        L.check(self.lib.tt_env_create(self.n_envs, index, C.byref(self.params), C.byref(h)))
        self._h = h
        n = self.n_envs
        with torch.cuda.device(self.device):
            self.obs = torch.zeros((n, L.OBS_DIM), dtype=torch.float32, device=self.device)
            self.reward = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.done = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self._info_bufs = None
        self.max_steer = float(self.params.max_steer)
        # bumped whenever something a captured step launch bakes in BY VALUE changes (reset seed, per-env-goal mode, pose
        # pool, step counter): holders of hipGraphs of step launches (DDPGRollout) re-capture when it moves
        self.graph_epoch = 0

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, "_h", None):
            self.lib.tt_env_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc):
        L.check(rc, self._h)

    def _as(self, x, dtype, shape=None):
        if x is None:
            return None
        t = torch.as_tensor(x, dtype=dtype, device=self.device).contiguous()
        if shape is not None:
            t = t.reshape(shape)
        return t

    def _info(self):
        if self._info_bufs is None:
            n = self.n_envs
            comp = torch.zeros((L.NINFO, n), dtype=torch.float64, device=self.device)
            viol = torch.zeros(n, dtype=torch.uint8, device=self.device)
            flags = torch.zeros(n, dtype=torch.uint8, device=self.device)
            self._info_bufs = (comp, viol, flags, L.TTInfo(comp.data_ptr(), viol.data_ptr(), flags.data_ptr()))
        return self._info_bufs

    # ------------------------------------------------------------------ reset family
    def reset(self, seed=0, mask=None, out=None):
        """Philox-sampled start poses for all (or masked) envs; returns obs [N,23]."""
        obs = self.obs if out is None else out
        m = self._as(mask, torch.uint8)
        self.graph_epoch += 1
        self._check(self.lib.tt_env_reset(self._h, _ptr(m), int(seed) & (2 ** 64 - 1), _ptr(obs), self._stream()))
        return obs

    def set_reset_pool(self, poses):
        """Resets (explicit and in-kernel) draw start poses from this [m,3] pool (x, y, yaw) instead of the box."""
        self.graph_epoch += 1
        if poses is None:
            self._pool = None
            self._check(self.lib.tt_env_set_reset_pool(self._h, None, 0))
            return
        self._pool = self._as(poses, torch.float64).reshape(-1, 3)      # kept alive here: the library only borrows it
        self._check(self.lib.tt_env_set_reset_pool(self._h, _ptr(self._pool), self._pool.shape[0]))

    def set_pose(self, start, goal=None, L2=None, idx=None, out=None):
        """Pose override: start [k,3] (x, y, yaw), optional goal [k,3], L2 [k], idx [k] (default 0..k-1)."""
        start = self._as(start, torch.float64).reshape(-1, 3)
        k = start.shape[0]
        goal = self._as(goal, torch.float64, (k, 3)) if goal is not None else None
        L2 = self._as(L2, torch.float64, (k,)) if L2 is not None else None
        idx = self._as(idx, torch.int32, (k,)) if idx is not None else None
        obs = self.obs if out is None else out
        if goal is not None or L2 is not None:
            self.graph_epoch += 1          # switches the handle to per-env goals / trailer lengths
        self._check(self.lib.tt_env_set_pose(self._h, _ptr(idx), k, _ptr(start), _ptr(goal), _ptr(L2), _ptr(obs),
                                             self._stream()))
        return obs

    def set_attrs(self, start=None, goal=None, L2=None, idx=None):
        k = None
        for x, w in ((start, 3), (goal, 3), (L2, 1)):
            if x is not None:
                k = torch.as_tensor(x).numel() // w
        if k is None:
            return
        start = self._as(start, torch.float64, (k, 3)) if start is not None else None
        goal = self._as(goal, torch.float64, (k, 3)) if goal is not None else None
        L2 = self._as(L2, torch.float64, (k,)) if L2 is not None else None
        idx = self._as(idx, torch.int32, (k,)) if idx is not None else None
        if goal is not None or L2 is not None:
            self.graph_epoch += 1
        self._check(self.lib.tt_env_set_attrs(self._h, _ptr(idx), k, _ptr(start), _ptr(goal), _ptr(L2), self._stream()))

    def set_state(self, state, idx=None):
        state = self._as(state, torch.float64).reshape(-1, 6)
        k = state.shape[0]
        idx = self._as(idx, torch.int32, (k,)) if idx is not None else None
        self._check(self.lib.tt_env_set_state(self._h, _ptr(idx), k, _ptr(state), self._stream()))

    def set_max_steps(self, max_steps, idx=None):
        m = self._as(max_steps, torch.int32).reshape(-1)
        k = m.shape[0]
        if k and (int(m.min()) < 0 or int(m.max()) > L.MAX_EPISODE_STEPS):
            raise ValueError(f"max_episode_steps must lie in [0, {L.MAX_EPISODE_STEPS}] (12-bit packed counters)")
        idx = self._as(idx, torch.int32, (k,)) if idx is not None else None
        self._check(self.lib.tt_env_set_max_steps(self._h, _ptr(idx), k, _ptr(m), self._stream()))

    def set_steps(self, steps, idx=None):
        """`env.episode_steps = ...` for envs idx (default 0..k-1): the step counter alone, the reward carry stays
        (include/ttenv.h: tt_env_set_steps)."""
        m = self._as(steps, torch.int32).reshape(-1)
        k = m.shape[0]
        if k and (int(m.min()) < 0 or int(m.max()) > L.MAX_EPISODE_STEPS):
            raise ValueError(f"episode_steps must lie in [0, {L.MAX_EPISODE_STEPS}] (12-bit packed counters)")
        idx = self._as(idx, torch.int32, (k,)) if idx is not None else None
        self._check(self.lib.tt_env_set_steps(self._h, _ptr(idx), k, _ptr(m), self._stream()))

    # ------------------------------------------------------------------ read-back
    @property
    def state(self):
        """[N,6] f64 (psi1, psi2, x1, y1, x2, y2), a fresh copy."""
        buf = torch.empty((6, self.n_envs), dtype=torch.float64, device=self.device)
        self._check(self.lib.tt_env_get_state(self._h, _ptr(buf), self._stream()))
        return buf.t().contiguous()

    def episode(self):
        n = self.n_envs
        steps = torch.empty(n, dtype=torch.int32, device=self.device)
        maxs = torch.empty(n, dtype=torch.int32, device=self.device)
        start = torch.empty((3, n), dtype=torch.float64, device=self.device)
        goal = torch.empty((3, n), dtype=torch.float64, device=self.device)
        L2 = torch.empty(n, dtype=torch.float64, device=self.device)
        self._check(self.lib.tt_env_get_episode(self._h, _ptr(steps), _ptr(maxs), _ptr(start), _ptr(goal), _ptr(L2),
                                                self._stream()))
        return dict(steps=steps, max_episode_steps=maxs, start=start.t().contiguous(), goal=goal.t().contiguous(), L2=L2)

    def observe(self, steering=None, out=None):
        obs = self.obs if out is None else out
        s = self._as(steering, torch.float32, (self.n_envs,)) if steering is not None else None
        self._check(self.lib.tt_env_observe(self._h, _ptr(s), _ptr(obs), self._stream()))
        return obs

    # ------------------------------------------------------------------ step
    def step(self, action, auto_reset=True, info=False, obs_out=None, reward_out=None, done_out=None):
        """action [N] f32 radians (already scaled by action_space.high, trainv2.py:516).

        Returns (obs [N,23] f32, reward [N] f32, done [N] u8, info).  The returned tensors are the
        env's own buffers (or the *_out tensors, e.g. slots of a replay ring) and are overwritten
        by the next step.  info=True adds the per-component f64 SoA of reward_functionv1.py:489-504."""
        a = action if (torch.is_tensor(action) and action.dtype == torch.float32 and action.is_contiguous()
                       and action.device == self.device) else self._as(action, torch.float32)
        if a.numel() != self.n_envs:
            raise ValueError(f"action has {a.numel()} elements, expected {self.n_envs}")
        obs = self.obs if obs_out is None else obs_out
        rew = self.reward if reward_out is None else reward_out
        done = self.done if done_out is None else done_out
        inf = None
        ti = None
        if info:
            comp, viol, flags, ti = self._info()
            inf = dict(comp=comp, violation=viol, flags=flags)
        self._check(self.lib.tt_env_step(self._h, _ptr(a), _ptr(obs), _ptr(rew), _ptr(done),
                                         C.byref(ti) if ti is not None else None, 1 if auto_reset else 0, self._stream()))
        return obs, rew, done, inf

    def step_ring(self, action, ring_view, auto_reset=True):
        """step() with its outputs addressed through a trajectory ring's device cursor (tt_env_step_ring): obs into slot
        t+1, reward and done into slot t -- one captured launch serves every ring position."""
        self._check(self.lib.tt_env_step_ring(self._h, _ptr(action), C.byref(ring_view), 1 if auto_reset else 0, self._stream()))

    def set_step_counter(self, counter):
        """counter: device int64 scalar tensor that every step launch advances by 1 (None detaches); see
        include/ttenv.h: tt_env_set_step_counter.  The tensor is kept alive by this object."""
        if counter is not None:
            assert counter.dtype == torch.int64 and counter.device == self.device and counter.numel() == 1
        self._step_counter = counter
        self.graph_epoch += 1
        self._check(self.lib.tt_env_set_step_counter(self._h, _ptr(counter) if counter is not None else None))

    def step_random(self, policy_seed=123, auto_reset=True, info=False, action_out=None, obs_out=None, reward_out=None,
                    done_out=None):
        """step() with the random policy of BASELINE.json config 2 drawn inside the kernel (graph-capturable)."""
        obs = self.obs if obs_out is None else obs_out
        rew = self.reward if reward_out is None else reward_out
        done = self.done if done_out is None else done_out
        inf, ti = None, None
        if info:
            comp, viol, flags, ti = self._info()
            inf = dict(comp=comp, violation=viol, flags=flags)
        self._check(self.lib.tt_env_step_random(self._h, int(policy_seed) & (2 ** 64 - 1), _ptr(action_out), _ptr(obs),
                                                _ptr(rew), _ptr(done), C.byref(ti) if ti is not None else None,
                                                1 if auto_reset else 0, self._stream()))
        return obs, rew, done, inf

    def rollout_random(self, k_steps, policy_seed=123, obs_out=None, reward_sum=None, episodes_done=None):
        """k_steps random-policy steps in one launch (state stays in registers); returns the last obs buffer."""
        obs = self.obs if obs_out is None else obs_out
        self._check(self.lib.tt_env_rollout_random(self._h, int(k_steps), int(policy_seed) & (2 ** 64 - 1), _ptr(obs),
                                                   _ptr(reward_sum), _ptr(episodes_done), self._stream()))
        return obs

    def state_dict(self):
        """Everything needed to resume this env batch (device blob copied to the host + host-side mode)."""
        nbytes = int(self.lib.tt_env_state_bytes(self._h))
        blob = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        meta = (C.c_uint64 * 4)()
        self._check(self.lib.tt_env_export(self._h, _ptr(blob), C.byref(meta), self._stream()))
        return {"blob": blob.cpu(), "meta": [int(x) for x in meta], "variant": self.variant,
                "pool": None if getattr(self, "_pool", None) is None else self._pool.cpu()}

    def load_state_dict(self, sd):
        blob = sd["blob"].to(self.device)
        meta = (C.c_uint64 * 4)(*sd["meta"])
        self.graph_epoch += 1              # reset seed and per-env-goal mode come back with the blob
        self._check(self.lib.tt_env_import(self._h, _ptr(blob), C.byref(meta), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()      # blob is a temporary
        if sd.get("pool") is not None:
            self.set_reset_pool(sd["pool"])

    def profile(self, max_launches):
        """Time the next `max_launches` step-kernel dispatches with per-dispatch HIP events (0 = off)."""
        self._check(self.lib.tt_env_profile(self._h, int(max_launches)))

    def profile_read(self):
        """-> (sum of step-kernel durations in ms, number of launches timed)."""
        ms, cnt = C.c_double(), C.c_int64()
        self._check(self.lib.tt_env_profile_read(self._h, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def random_actions(self, seed, step, out=None):
        out = torch.empty(self.n_envs, dtype=torch.float32, device=self.device) if out is None else out
        L.check(self.lib.tt_random_actions(self.n_envs, int(seed), int(step), _ptr(out), self._stream()))
        return out
