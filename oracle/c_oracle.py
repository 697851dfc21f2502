"""ORACLE (test infrastructure, not product): ctypes view of oracle/libtt_oracle.so,
the plain-C restatement in tt_oracle.c.  Used only by tests/, smoke() and bench.py's
cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libtt_oracle.so")
OBS_DIM, NINFO = 23, 12
INFO_KEYS = ("total_reward", "progress_reward", "heading_reward", "orientation_reward", "staged_success",
             "safety_penalty", "exploration_bonus", "final_success_bonus", "backward_penalty",
             "smoothness_penalty", "cumulative_backward", "movement_budget")
F_JACKKNIFE, F_OUT_OF_MAP, F_MAX_STEPS, F_GOAL_REACHED, F_GOAL_PASSED, F_EXCESSIVE_BACK, F_SUCCESS = (1 << i for i in range(7))


class Params(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("L1", "L2", "hitch_offset", "v1x", "dt", "map_min", "map_max", "max_steer",
                                          "max_expected_distance", "position_threshold", "orientation_threshold",
                                          "step_length")] + \
               [("extra_steps", C.c_int32), ("fixed_max_steps", C.c_int32), ("term_mask", C.c_uint32),
                ("variant", C.c_int32), ("stateless_reward", C.c_int32), ("reserved_", C.c_int32),
                ("goal", C.c_double * 3)]


class Env(C.Structure):
    _fields_ = [("y", C.c_double * 6), ("start", C.c_double * 3), ("goal", C.c_double * 3), ("L2", C.c_double),
                ("prev_dist", C.c_double), ("cum_back", C.c_double), ("closest", C.c_double), ("hist", C.c_double * 5),
                ("prev_steer", C.c_float), ("steps", C.c_int32), ("max_steps", C.c_int32), ("bt_steps", C.c_int32),
                ("stages", C.c_uint8 * 3), ("has_carry", C.c_uint8), ("flags", C.c_uint8), ("violation", C.c_uint8)]


def build(force=False):
    src = [os.path.join(HERE, f) for f in ("tt_oracle.c", "tt_oracle.h", "Makefile")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.tto_env_size.restype = C.c_int
        assert L.tto_env_size() == C.sizeof(Env), (L.tto_env_size(), C.sizeof(Env))
        L.tto_rollout_random.restype = C.c_long
        L.tto_rollout_random.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


class COracle:
    """N scalar envs stepped on the CPU by the C restatement."""

    def __init__(self, n, variant=0):
        self.n = n
        self.params = Params()
        lib().tto_params_default(C.byref(self.params), variant)
        self.envs = (Env * n)()

    def place(self, start, goal=None, L2=None):
        start = np.ascontiguousarray(np.broadcast_to(np.asarray(start, np.float64), (self.n, 3)))
        goal = None if goal is None else np.ascontiguousarray(np.broadcast_to(np.asarray(goal, np.float64), (self.n, 3)))
        L2 = None if L2 is None else np.ascontiguousarray(np.broadcast_to(np.asarray(L2, np.float64), (self.n,)))
        obs = np.zeros((self.n, OBS_DIM), np.float32)
        lib().tto_place_batch(C.byref(self.params), self.envs, self.n, _p(start, C.c_double), _p(goal, C.c_double),
                              _p(L2, C.c_double), _p(obs, C.c_float))
        return obs

    def set_state(self, i, y):
        y = np.ascontiguousarray(y, np.float64)
        lib().tto_set_state(C.byref(self.envs[i]), _p(y, C.c_double))

    def set_max_steps(self, i, m):
        self.envs[i].max_steps = int(m)

    def observe(self, i, steering=0.0):
        obs = np.zeros(OBS_DIM, np.float32)
        lib().tto_observe(C.byref(self.params), C.byref(self.envs[i]), C.c_double(steering), _p(obs, C.c_float))
        return obs

    def state(self):
        return np.array([list(e.y) for e in self.envs], np.float64)

    def flags(self):
        return np.array([e.flags for e in self.envs], np.uint8)

    def violation(self):
        return np.array([e.violation for e in self.envs], np.uint8)

    def step(self, actions, nthreads=1):
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(actions, np.float32), (self.n,)))
        obs = np.zeros((self.n, OBS_DIM), np.float32)
        rew = np.zeros(self.n, np.float64)
        done = np.zeros(self.n, np.uint8)
        info = np.zeros((self.n, NINFO), np.float64)
        lib().tto_step_batch(C.byref(self.params), self.envs, self.n, _p(a, C.c_float), _p(obs, C.c_float),
                             _p(rew, C.c_double), _p(done, C.c_uint8), _p(info, C.c_double), nthreads)
        return obs, rew, done.astype(bool), info


def rhs(y, steering, variant=0, L2=None):
    """The ODE right-hand side of the restatement for one state (fixture F6 pins it to the reference's kinematic_model)."""
    p = Params()
    lib().tto_params_default(C.byref(p), variant)
    y = np.ascontiguousarray(y, np.float64)
    d = np.zeros(6, np.float64)
    lib().tto_rhs(C.byref(p), C.c_double(p.L2 if L2 is None else L2), C.c_double(steering), _p(y, C.c_double), _p(d, C.c_double))
    return d


def rollout_random(n_envs, n_steps, seed=0, nthreads=1, variant=0):
    p = Params()
    lib().tto_params_default(C.byref(p), variant)
    rs = C.c_double()
    n = lib().tto_rollout_random(C.byref(p), n_envs, n_steps, seed, nthreads, C.byref(rs))
    return int(n), rs.value
