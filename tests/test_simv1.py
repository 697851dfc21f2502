"""simv1 variant (BASELINE config 5).

Pinned to the reference (fixture F6, tests/golden/make_golden_simv1.py): the constants, the ODE right-hand side, the
observation, the in-map predicate and free-running trajectories through the reference's own step() up to its reward call --
state, observation and the four termination flags of simv1.py:422-432, incl. the 300-step cap reached free-running.
STILL UNPINNED: the reward call of simv1.step (it raises TypeError in the reference, simv1.py:435) and the Dubins planner
behind reset() (not in the repository): there these tests pin the build's own restatement to its stated assumptions --
textbook Dubins geometry on the CPU, and on the GPU the variant-1 kernel against the variant-1 C oracle (the same shared
reward code that IS pinned through simv2)."""
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN
from ddpg_trucktrailer_amd import simv1_reset as S

F6 = os.path.join(GOLDEN, "f6_simv1.npz")


def _f6():
    return np.load(F6, allow_pickle=False)


def _f6_trajectories():
    z = _f6()
    return [pytest.param(str(n), id=str(n)) for n in z["traj/names"]]


def test_f6_constants_ode_and_observation_pin_the_variant_1_oracle():
    """oracle/tt_oracle.c, variant 1, against the reference's simv1: constructor constants (simv1.py:23-99), kinematic_model
    (:180-214) and compute_observation (:101-179)."""
    from oracle import c_oracle
    z = _f6()
    ora = c_oracle.COracle(1, variant=1)
    p = ora.params
    for mine, ref in ((p.L1, "L1"), (p.L2, "L2"), (p.hitch_offset, "hitch_offset"), (p.v1x, "v1x"), (p.dt, "dt"),
                      (p.map_min, "min_map_x"), (p.map_min, "min_map_y"), (p.map_max, "max_map_x"), (p.map_max, "max_map_y"),
                      (p.max_steer, "max_steering_angle"), (-p.max_steer, "min_steering_angle"),
                      (p.max_expected_distance, "max_expected_distance"), (p.position_threshold, "position_threshold"),
                      (p.orientation_threshold, "orientation_threshold"), (p.fixed_max_steps, "max_episode_steps")):
        assert mine == float(z["const/" + ref]), ref
    assert tuple(p.goal) == tuple(z["const/goal"]) and float(z["const/time"]) == 0.0
    assert p.term_mask == (c_oracle.F_JACKKNIFE | c_oracle.F_OUT_OF_MAP | c_oracle.F_MAX_STEPS | c_oracle.F_GOAL_REACHED)
    for y, d, xd in zip(z["ode/x"], z["ode/delta"], z["ode/xd"]):
        assert np.abs(c_oracle.rhs(y, float(d), variant=1) - xd).max() <= 1e-13 * max(1.0, np.abs(xd).max())
    ora.place(np.array([[0.0, 0.0, 1.0]]))
    for st, steer, ob in zip(z["obs/state"], z["obs/steer"], z["obs/out"]):
        ora.set_state(0, st)
        assert np.abs(ora.observe(0, float(steer)) - ob).max() <= 1e-6


def test_f6_in_map_predicate():
    """simv1_reset.points_out_of_map == the reference's check_path_out_of_Map on hand-made paths (simv1.py:239-253) and its
    check_out_of_Map truth table (:225-237): strict comparisons, a point ON an edge is inside."""
    z = _f6()
    for name, want in zip(z["path/names"], z["path/out"]):
        assert S.points_out_of_map(z[f"path/{name}/x"], z[f"path/{name}/y"]) == bool(want), name
    for (x1, y1, x2, y2), want in zip(z["oom/xy"], z["oom/out"]):
        assert S.points_out_of_map([x1, x2], [y1, y2]) == bool(want)
    assert z["oom/out"].any() and not z["oom/out"].all()
    # the other two truth tables, as the reference answered them: strict > 90 degrees, >= 300 steps
    r90 = np.deg2rad(90)
    assert (z["jk/out"] == (np.abs(z["jk/psi1"] - z["jk/psi2"]) > r90)).all() and z["jk/out"].any() and not z["jk/out"].all()
    assert (z["ms/out"] == (z["ms/step"] >= 300)).all()


@pytest.mark.parametrize("name", _f6_trajectories())
def test_f6_trajectories_pin_the_variant_1_oracle(name):
    """Free-running against the reference's step() (up to its reward call): state, observation <= 1e-5, flags and done exact."""
    from oracle import c_oracle
    z = _f6()
    g = lambda k: z[f"traj/{name}/{k}"]
    ora = c_oracle.COracle(1, variant=1)
    obs0 = ora.place(g("start")[None])
    assert np.abs(obs0[0] - g("obs0")).max() <= 1e-6 and np.abs(ora.state()[0] - g("state0")).max() == 0.0
    ora.envs[0].steps = int(g("steps_before"))
    keys = (c_oracle.F_JACKKNIFE, c_oracle.F_OUT_OF_MAP, c_oracle.F_MAX_STEPS, c_oracle.F_GOAL_REACHED)
    for t, a in enumerate(g("actions")):
        obs, rew, done, info = ora.step(np.array([a], np.float32))
        assert np.abs(ora.state()[0] - g("states")[t]).max() <= 1e-5, t
        assert np.abs(obs[0] - g("obs")[t]).max() <= 1e-5, t
        assert [bool(ora.flags()[0] & k) for k in keys] == list(g("flags")[t]), t
        assert bool(done[0]) == bool(g("done")[t]), t
    assert g("reward_call_raised_typeerror").all()      # what stays unpinned, as the fixture recorded it


@pytest.mark.gpu
@pytest.mark.parametrize("name", _f6_trajectories())
def test_f6_trajectories_on_the_hip_kernel(gpu_device, name):
    """Variant 1 of the HIP env (through the C ABI) against the reference's simv1 trajectories of F6, free-running: state and
    observation <= 1e-5, the four flag bits and done (mask of simv1.py:432) exact, the 300-step cap reached at step 300."""
    import torch
    from ddpg_trucktrailer_amd import _lib as L
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    z = _f6()
    g = lambda k: z[f"traj/{name}/{k}"]
    n = 64                                       # one wave: every lane runs the same episode
    env = TruckTrailerVecEnv(n, variant=1)
    p = env.params
    zc = lambda k: float(z["const/" + k])
    assert (p.L1, p.L2, p.v1x, p.dt, p.fixed_max_steps) == (zc("L1"), zc("L2"), zc("v1x"), zc("dt"), int(zc("max_episode_steps")))
    obs0 = env.set_pose(torch.tensor(np.broadcast_to(g("start"), (n, 3)).copy(), device="cuda"))
    assert np.abs(obs0.cpu().numpy() - g("obs0")).max() <= 1e-6
    assert np.abs(env.state.cpu().numpy() - g("state0")).max() == 0.0
    if int(g("steps_before")):                   # the generator wrote env.episode_steps on the reference: tt_env_set_steps
        env.set_steps([int(g("steps_before"))] * n)
        assert (env.episode()["steps"].cpu().numpy() == int(g("steps_before"))).all()
    bits = (L.F_JACKKNIFE, L.F_OUT_OF_MAP, L.F_MAX_STEPS, L.F_GOAL_REACHED)
    for t, a in enumerate(g("actions")):
        obs, rew, done, info = env.step(torch.full((n,), float(a), dtype=torch.float32, device="cuda"), auto_reset=False, info=True)
        assert np.abs(env.state.cpu().numpy() - g("states")[t]).max() <= 1e-5, t
        assert np.abs(obs.cpu().numpy() - g("obs")[t]).max() <= 1e-5, t
        fl = info["flags"].cpu().numpy()
        assert (fl == fl[0]).all()
        assert [bool(fl[0] & b) for b in bits] == list(g("flags")[t]), t
        assert bool(done[0].item()) == bool(g("done")[t]), t
    env.close()


@pytest.mark.gpu
def test_f6_observation_on_the_hip_kernel(gpu_device):
    """compute_observation of the reference's simv1 (256 known answers of F6) through tt_env_set_state + tt_env_observe."""
    import torch
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    z = _f6()
    st, steer, want = z["obs/state"], z["obs/steer"], z["obs/out"]
    env = TruckTrailerVecEnv(len(st), variant=1)
    env.set_pose(torch.tensor(np.tile([0.0, 0.0, 1.0], (len(st), 1)), device="cuda"))
    env.set_state(torch.tensor(st, device="cuda"))
    got = env.observe(steering=torch.tensor(steer, dtype=torch.float32, device="cuda")).cpu().numpy()
    # the steering enters as f32 here (the C ABI's action type): sin / cos of it differ from the f64 ones by <= 3e-8
    assert np.abs(got - want).max() <= 1e-6
    env.close()


def test_dubins_known_answers():
    c = 1.0 / 6
    # straight ahead: pure S of length 20
    x, y, yaw, word, L = S.plan_dubins_path(0, 0, 0, 20, 0, 0, c)
    assert abs(L - 20) < 1e-9 and np.abs(y).max() < 1e-9 and abs(x[-1] - 20) < 1e-9
    # quarter-circle left (radius 6), then 5 m straight up
    x, y, yaw, word, L = S.plan_dubins_path(0, 0, 0, 6, 11, math.pi / 2, c)
    assert abs(L - (6 * math.pi / 2 + 5)) < 1e-9 and abs(x[-1] - 6) < 1e-6 and abs(y[-1] - 11) < 1e-6
    # quarter-circle alone: the degenerate case where both turning circles coincide
    x, y, yaw, word, L = S.plan_dubins_path(0, 0, 0, 6, 6, math.pi / 2, c)
    assert abs(L - 6 * math.pi / 2) < 1e-6 and abs(x[-1] - 6) < 1e-6 and abs(y[-1] - 6) < 1e-6
    # every path ends at the goal pose and is never shorter than the straight line
    rng = np.random.RandomState(0)
    for _ in range(200):
        s = rng.uniform(-40, 40, 2); g = rng.uniform(-40, 40, 2); a, b = rng.uniform(0, 2 * math.pi, 2)
        x, y, yaw, word, L = S.plan_dubins_path(s[0], s[1], a, g[0], g[1], b, c)
        assert abs(x[-1] - g[0]) < 1e-6 and abs(y[-1] - g[1]) < 1e-6
        assert abs(math.remainder(yaw[-1] - b, 2 * math.pi)) < 1e-6
        assert L >= np.hypot(*(g - s)) - 1e-9
        seg = np.hypot(np.diff(x), np.diff(y))
        assert seg.max() <= 0.1 + 1e-9 and abs(seg.sum() - L) < 1e-2 * max(1, L)
        # curvature bound: heading change per metre <= 1/6
        dyaw = np.abs(np.diff(np.unwrap(yaw)))
        assert (dyaw <= seg * c * (1 + 1e-3) + 1e-9).all()      # seg is the chord, slightly shorter than the arc


def test_backward_path_and_pose_generation():
    # a reversing vehicle heading +y at (0, -10) reaches the goal (0,-30, +90deg) by backing straight down
    px, py, pyaw = S.plan_dubins_path_backward(0.0, -10.0, math.pi / 2, 0.0, -30.0, math.pi / 2, 1.0 / 6)
    assert np.abs(px).max() < 1e-9 and (np.diff(py) < 0).all() and np.allclose(pyaw, math.pi / 2)
    assert not S.path_out_of_map(0.0, -10.0, math.pi / 2, 0.0, -30.0, math.pi / 2)
    assert S.path_out_of_map(39.5, 0.0, math.pi, 0.0, -30.0, math.pi / 2)     # travelling +x at x = 39.5: any radius-6 turn leaves the map
    pool = S.generate_pose_pool(64, seed=3)
    assert pool.shape == (64, 3) and (np.abs(pool[:, :2]) <= 40).all()
    assert (np.hypot(pool[:, 0] - 0.0, pool[:, 1] + 30.0) >= 15).all()     # simv1.py:270-272
    for sx, sy, syaw in pool[:16]:
        assert not S.path_out_of_map(sx, sy, syaw, 0.0, -30.0, math.pi / 2)
    assert np.array_equal(pool, S.generate_pose_pool(64, seed=3))


@pytest.mark.gpu
def test_simv1_kernel_vs_oracle_and_pool_resets(gpu_device):
    import torch
    from ddpg_trucktrailer_amd import _lib as L
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    from oracle import c_oracle
    n = 2048
    env = TruckTrailerVecEnv(n, variant=1)
    p = env.params
    assert (p.L1, p.L2, p.fixed_max_steps, p.term_mask, p.stateless_reward) == (5.74, 10.192, 300, L.F_JACKKNIFE |
            L.F_OUT_OF_MAP | L.F_MAX_STEPS | L.F_GOAL_REACHED, 1)
    pool = S.generate_pose_pool(256, seed=1)
    env.set_reset_pool(pool)
    env.reset(seed=9)
    start = env.episode()["start"].cpu().numpy()
    key = {tuple(r) for r in pool.round(12)}
    assert all(tuple(r) in key for r in start.round(12)), "reset poses must come from the pool"
    assert (env.episode()["max_episode_steps"] == 300).all()
    ora = c_oracle.COracle(n, variant=1)
    ora.place(start)
    alive = np.ones(n, bool)
    for t in range(120):
        a = env.random_actions(5, t)
        obs, rew, done, info = env.step(a, auto_reset=False, info=True)
        o_obs, o_rew, o_done, o_info = ora.step(a.cpu().numpy(), nthreads=8)
        m = alive
        assert np.abs(obs.cpu().numpy()[m] - o_obs[m]).max() <= 1e-5
        assert np.abs(info["comp"].cpu().numpy().T[m] - o_info[m]).max() <= 1e-5
        assert (done.cpu().numpy().astype(bool)[m] == o_done[m]).all()
        fl = info["flags"].cpu().numpy()
        # goal_passed / excessive_backward are still reported but do not end a simv1 episode (simv1.py:432)
        assert not (done.cpu().numpy().astype(bool) & ((fl & 0x0F) == 0)).any()
        alive &= ~o_done
    # stateless reward: smoothness is always 0 and the progress term is the constant 0.2*15 (simv1.py:435)
    assert (info["comp"][L.INFO_ROWS.index("smoothness_penalty")] == 0).all()
    assert torch.allclose(info["comp"][L.INFO_ROWS.index("progress_reward")], torch.tensor(3.0, dtype=torch.float64, device="cuda"))
    # in-kernel auto-reset also draws from the pool
    env.reset(seed=10)
    for t in range(60):
        env.step(torch.full((n,), 0.78, device="cuda"), auto_reset=True)
    start = env.episode()["start"].cpu().numpy()
    assert all(tuple(r) in key for r in start.round(12))
    env.close()


@pytest.mark.gpu
def test_simv1_facade(gpu_device):
    from ddpg_trucktrailer_amd.env import Truck_trailer_Env_1
    import random
    random.seed(4)
    env = Truck_trailer_Env_1()
    obs, info = env.reset()
    assert obs.shape == (23,) and env.max_episode_steps == 300 and env.L2 == 10.192
    assert len(env.path_x) == len(env.path_y) == len(env.path_yaw) > 10
    assert abs(env.path_x[-1] - env.goalx) < 1e-6 and abs(env.path_y[-1] - env.goaly) < 1e-6
    o, r, d, i = env.step(np.array([0.1], np.float32))
    assert isinstance(r, np.float64) and i["smoothness_penalty"] == 0
    env.close()


@pytest.mark.gpu
def test_simv1_at_bench_size(gpu_device):
    """BASELINE config 5 at its stated size, N = 65536 (PARITY UNPINNED, see the module docstring): properties over all
    envs -- the termination mask of simv1.py:432 (goal-passed / excessive-backward are reported but never end an
    episode), the fixed 300-step cap, stateless reward, pool resets -- and the variant-1 C oracle on a 4096-env slice."""
    import torch
    from ddpg_trucktrailer_amd import _lib as L
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    from oracle import c_oracle
    n, m = 65536, 4096
    env = TruckTrailerVecEnv(n, variant=1)
    pool = S.generate_pose_pool(512, seed=2)
    env.set_reset_pool(pool)
    env.reset(seed=3)
    ep = env.episode()
    assert (ep["max_episode_steps"] == 300).all()
    start = ep["start"].cpu().numpy()
    ora = c_oracle.COracle(m, variant=1)
    ora.place(start[:m])
    alive = np.ones(m, bool)
    ended_by = np.zeros(8, np.int64)
    masked_only = 0
    for t in range(60):
        a = env.random_actions(7, t)
        obs, rew, done, info = env.step(a, auto_reset=False, info=True)
        fl = info["flags"]
        d = done.bool()
        # done <=> a cause of the simv1 mask; the two unmasked causes alone never end an episode
        assert torch.equal(d, (fl & L.default_params(1).term_mask) != 0)
        masked_only += int(((fl & 0x30) != 0).logical_and((fl & 0x0F) == 0).sum())
        # (finished envs keep moving here -- auto_reset is off -- so positions / 40 leave [-1, 1]; everything else is bounded)
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
        assert (obs[:, [2, 3, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 20, 21, 22]].abs() <= 1.0 + 1e-6).all()
        o_obs, o_rew, o_done, o_info = ora.step(a[:m].cpu().numpy(), nthreads=8)
        k = alive
        assert np.abs(obs[:m].cpu().numpy()[k] - o_obs[k]).max() <= 1e-5
        assert np.abs(info["comp"][:, :m].cpu().numpy().T[k] - o_info[k]).max() <= 1e-5
        assert (d[:m].cpu().numpy()[k] == o_done[k]).all()
        alive &= ~o_done
        for b in range(7):
            ended_by[b] += int(((fl >> b) & 1).logical_and(d).sum())
    assert masked_only > 0, "the run must contain steps where only an unmasked cause fired"
    assert ended_by[0] > 0 and ended_by[1] > 0            # jackknife and out-of-map both occur under the random policy
    assert (info["comp"][L.INFO_ROWS.index("smoothness_penalty")] == 0).all()
    # auto-reset at full size draws from the pool
    env.reset(seed=4)
    for t in range(40):
        env.step(env.random_actions(8, t), auto_reset=True)
    key = {tuple(r) for r in pool.round(12)}
    st = env.episode()["start"].cpu().numpy()
    assert all(tuple(r) in key for r in st[::97].round(12))
    env.close()
