"""GPU (-m gpu): the HIP env step, called through the C ABI of libttenv.so, against
 (1) the golden fixtures captured from the real reference (F1 recorded episode, F2 seeded, F3 branches, F4 obs),
 (2) the plain-C oracle on the same seeded inputs at N = 4096 (BASELINE.json config 2),
 (3) size-independent properties at N = 65536 (the bench size).
Tolerance: 1e-5 absolute on state, observation and reward (BASELINE.json north_star), free-running;
flags / done / violation labels exact.  Nothing here reads /root/reference."""
import numpy as np
import pytest

from conftest import GOLDEN, all_trajectories, needs_raw_state

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def torch_mod(gpu_device):
    import torch
    return torch


def _facade_replay(t):
    from ddpg_trucktrailer_amd.env import Truck_trailer_Env_2
    env = Truck_trailer_Env_2()
    env.reset(seed=0)
    # callers' override pattern (DDPG/test.py:96-115)
    env.goalx, env.goaly, env.goalyaw = float(t["goal"][0]), float(t["goal"][1]), float(t["goal"][2])
    env.L2 = float(t["L2"])
    env.startx, env.starty, env.startyaw = (float(x) for x in t["start"])
    env.max_episode_steps = env.compute_max_steps()
    if needs_raw_state(t):
        env.state = t["state0"]
    else:
        x1 = env.startx + env.L2 * np.cos(env.startyaw)
        y1 = env.starty + env.L2 * np.sin(env.startyaw)
        env.state = np.array([env.startyaw, env.startyaw, x1, y1, env.startx, env.starty], dtype=np.float32)
    if int(t["max_episode_steps"]) != env.max_episode_steps:
        env.max_episode_steps = int(t["max_episode_steps"])
    obs0 = env.compute_observation(env.state, 0.0)
    out = dict(obs0=obs0, states=[], obs=[], reward=[], done=[], viol=[], flags=[], info=[])
    for a in t["actions"]:
        o, r, d, info = env.step(np.array([a], np.float32))
        assert isinstance(r, np.float64) and o.dtype == np.float32 and o.shape == (23,) and isinstance(d, bool)
        out["states"].append(env.state.copy()); out["obs"].append(o); out["reward"].append(r); out["done"].append(d)
        out["viol"].append(info["violation_type"])
        out["flags"].append([env.jackknife, env.out_of_map, env.max_steps_reached, env.goal_reached, env.goal_passed,
                             env.excessive_backward])
        out["info"].append(info)
    env.close()
    return out


VIOL = ("none", "jackknife", "jackknife_warning", "major_boundary", "minor_boundary", "past_the_goal", "max_step",
        "excessive_backward")


@pytest.mark.parametrize("t", all_trajectories())
def test_fixture_trajectory_through_gym_facade(torch_mod, t):
    r = _facade_replay(t)
    assert np.abs(r["obs0"] - t["obs0"]).max() <= TOL
    assert np.abs(np.array(r["states"]) - t["states"]).max() <= TOL
    assert np.abs(np.array(r["obs"]) - t["obs"]).max() <= TOL
    assert np.abs(np.array(r["reward"]) - t["reward"]).max() <= TOL
    assert (np.array(r["done"]) == t["done"]).all()
    assert [VIOL[v] for v in t["violation"]] == r["viol"]
    assert (np.array(r["flags"]) == t["flags"]).all()
    cols = dict(progress_reward=2, heading_reward=3, orientation_reward=4, staged_success=5, safety_penalty=6,
                exploration_bonus=7, final_success_bonus=8, backward_penalty=9, smoothness_penalty=10)
    for key, col in cols.items():
        got = np.array([float(i[key]) for i in r["info"]])
        assert np.abs(got - t["info"][:, col]).max() <= TOL, key
    got = np.array([float(i["backward_movement_info"]["cumulative_backward"]) for i in r["info"]])
    assert np.abs(got - t["info"][:, 11]).max() <= TOL
    assert [bool(i["success"]) for i in r["info"]] == t["success"].tolist()


def _f7():
    from conftest import load_group
    return [pytest.param(t, id=f"f7-{n}") for n, t in load_group("f7_episode_steps.npz").items()]


@pytest.mark.parametrize("t", _f7())
def test_episode_steps_written_through_the_facade(torch_mod, t):
    """Fixture F7 (the reference with `env.episode_steps = k` written between steps): the facade's attribute write goes through
    tt_env_set_steps and moves the device step counter alone -- exploration tiers, max-step penalty and `max_steps_reached`
    follow it, the reward carry (movement budget, progress window, first steering) does not.  The `no_reset` case is the caller
    pattern of DDPG/episode_replay_collectorv2.py:258-269 on an env object that was never reset."""
    from ddpg_trucktrailer_amd.env import Truck_trailer_Env_2
    writes = {int(i): int(v) for i, v in t["writes"]}
    env = Truck_trailer_Env_2()
    if not bool(t["no_reset"]):
        env.reset(seed=0)
    env.state = t["state0"].astype(np.float32) if bool(t["no_reset"]) else t["state0"]
    env.startx, env.starty, env.startyaw = (float(x) for x in t["start"])
    env.goalx, env.goaly, env.goalyaw = (float(x) for x in t["goal"])
    env.max_episode_steps = env.compute_max_steps()
    if int(t["max_episode_steps"]) != env.max_episode_steps:
        env.max_episode_steps = int(t["max_episode_steps"])
    if bool(t["no_reset"]):
        env.episode_steps = 0
    assert np.abs(env.compute_observation(env.state, 0.0) - t["obs0"]).max() <= TOL
    steps = 0
    for k, a in enumerate(t["actions"]):
        if k in writes:
            env.episode_steps = steps = writes[k]
        o, r, d, info = env.step(np.array([a], np.float32))
        steps += 1
        assert env.episode_steps == steps and int(env._vec.episode()["steps"][0].item()) == steps, k
        assert np.abs(env.state - t["states"][k]).max() <= TOL and np.abs(o - t["obs"][k]).max() <= TOL, k
        assert abs(r - t["reward"][k]) <= TOL and d == bool(t["done"][k]) and info["violation_type"] == VIOL[t["violation"][k]], k
        assert [env.jackknife, env.out_of_map, env.max_steps_reached, env.goal_reached, env.goal_passed,
                env.excessive_backward] == t["flags"][k].tolist(), k
        assert abs(info["exploration_bonus"] - t["info"][k, 7]) <= TOL and abs(info["safety_penalty"] - t["info"][k, 6]) <= TOL, k
        assert abs(info["backward_movement_info"]["movement_budget"] - t["info"][k, 12]) <= TOL, k
        assert abs(info["backward_penalty"] - t["info"][k, 9]) <= TOL and abs(info["smoothness_penalty"] - t["info"][k, 10]) <= TOL, k
    with pytest.raises(ValueError):
        env.episode_steps = 5000
        env.step(np.array([0.0], np.float32))
    env.close()


def test_set_steps_is_range_checked_at_the_c_abi(torch_mod):
    import ctypes as C
    torch = torch_mod
    from ddpg_trucktrailer_amd import _lib as L
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    env = TruckTrailerVecEnv(8)
    env.reset(seed=3)
    bad = torch.tensor([1, 4096], dtype=torch.int32, device="cuda")
    rc = env.lib.tt_env_set_steps(env._h, None, 2, C.c_void_p(bad.data_ptr()), None)
    assert rc == L.TT_EINVAL and b"4096" in env.lib.tt_last_error(env._h)
    assert (env.episode()["steps"].cpu().numpy() == 0).all()                 # nothing was written
    env.set_steps([7, 4095], idx=[5, 2])
    assert env.episode()["steps"].cpu().numpy().tolist() == [0, 0, 4095, 0, 0, 7, 0, 0]
    env.close()


def test_golden_episode_against_the_authors_recording(torch_mod):
    """F1 against what the reference's author recorded (not only our replay of it)."""
    from conftest import load_group
    t = load_group("f1_golden_episode.npz")["golden"]
    r = _facade_replay(t)
    assert np.abs(np.array(r["states"]) - t["recorded_states"][1:]).max() <= TOL
    assert np.abs(np.array(r["reward"]) - t["recorded_info"][:, 0]).max() <= TOL
    assert abs(float(np.sum(r["reward"])) - 4792.9998) < 1e-3
    assert r["info"][-1]["success"] and r["done"][-1] and r["info"][-1]["final_success_bonus"] == 200.0


def test_all_fixtures_batched_in_one_handle(torch_mod):
    """Every fixture trajectory in its own lane of ONE vector env (ragged lengths, per-env goal/L2)."""
    torch = torch_mod
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    ts = [p.values[0] for p in all_trajectories()]
    n = len(ts)
    env = TruckTrailerVecEnv(n)
    env.set_pose(np.stack([t["start"] for t in ts]), goal=np.stack([t["goal"] for t in ts]),
                 L2=np.array([float(t["L2"]) for t in ts]))
    raw = [i for i, t in enumerate(ts) if needs_raw_state(t)]
    env.set_state(np.stack([ts[i]["state0"] for i in raw]), idx=raw)
    env.set_max_steps([int(t["max_episode_steps"]) for t in ts])
    assert (env.episode()["max_episode_steps"].cpu().numpy() == [int(t["max_episode_steps"]) for t in ts]).all()
    obs0 = env.observe().cpu().numpy()
    for i, t in enumerate(ts):
        assert np.abs(obs0[i] - t["obs0"]).max() <= TOL
    T = max(len(t["actions"]) for t in ts)
    for k in range(T):
        a = np.array([t["actions"][k] if k < len(t["actions"]) else 0.0 for t in ts], np.float32)
        obs, rew, done, info = env.step(torch.from_numpy(a).cuda(), auto_reset=False, info=True)
        st = env.state.cpu().numpy(); ob = obs.cpu().numpy(); tot = info["comp"][0].cpu().numpy()
        dn = done.cpu().numpy(); fl = info["flags"].cpu().numpy(); vi = info["violation"].cpu().numpy()
        r32 = rew.cpu().numpy()
        for i, t in enumerate(ts):
            if k >= len(t["actions"]):
                continue
            assert np.abs(st[i] - t["states"][k]).max() <= TOL
            assert np.abs(ob[i] - t["obs"][k]).max() <= TOL
            assert abs(tot[i] - t["reward"][k]) <= TOL
            assert r32[i] == np.float32(tot[i])                      # the f32 reward is the rounded f64 total
            assert bool(dn[i]) == bool(t["done"][k]) and vi[i] == t["violation"][k]
            assert [(fl[i] >> b) & 1 for b in range(6)] == t["flags"][k].astype(int).tolist()
    env.close()


def test_observation_known_answers(torch_mod):
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    z = np.load(f"{GOLDEN}/f4_observation.npz", allow_pickle=False)
    k = len(z["states"])
    env = TruckTrailerVecEnv(k)
    env.set_pose(np.zeros((k, 3)), goal=z["goals"])
    env.set_state(z["states"])
    obs = env.observe(steering=z["steer"].astype(np.float32)).cpu().numpy()
    assert np.abs(obs - z["obs_f64_state"]).max() <= 1e-6   # f32 steering input costs ~3e-8
    env.set_state(z["states"].astype(np.float32).astype(np.float64))
    assert np.abs(env.observe().cpu().numpy() - z["obs_f32_state_steer0"]).max() <= 1e-6
    env.close()


def _oracle_pair(n, seed):
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    from oracle import c_oracle
    env = TruckTrailerVecEnv(n)
    obs0 = env.reset(seed=seed).cpu().numpy()
    start = env.episode()["start"].cpu().numpy()
    ora = c_oracle.COracle(n)
    o0 = ora.place(start)
    assert np.abs(obs0 - o0).max() <= TOL
    return env, ora, start


def test_n4096_random_policy_vs_c_oracle(torch_mod):
    """BASELINE.json config 2: N=4096, random policy, obs/reward/state/done parity vs the CPU oracle,
    every env followed until its own episode ends."""
    torch = torch_mod
    n = 4096
    env, ora, start = _oracle_pair(n, seed=27)
    assert start[:, 0].min() >= -27 and start[:, 0].max() <= 27 and start[:, 1].min() >= 0 and start[:, 1].max() <= 27
    assert start[:, 2].min() >= np.pi / 4 and start[:, 2].max() <= 2 * np.pi / 3
    alive = np.ones(n, bool)
    causes = np.zeros(8, int)
    for t in range(260):
        a = env.random_actions(seed=123, step=t)
        obs, rew, done, info = env.step(a, auto_reset=False, info=True)
        o_obs, o_rew, o_done, o_info = ora.step(a.cpu().numpy(), nthreads=8)
        m = alive
        assert np.abs(obs.cpu().numpy()[m] - o_obs[m]).max() <= TOL
        assert np.abs(env.state.cpu().numpy()[m] - ora.state()[m]).max() <= TOL
        assert np.abs(info["comp"].cpu().numpy().T[m] - o_info[m]).max() <= TOL
        assert (done.cpu().numpy().astype(bool)[m] == o_done[m]).all()
        assert (info["violation"].cpu().numpy()[m] == ora.violation()[m]).all()
        assert (info["flags"].cpu().numpy()[m] == ora.flags()[m]).all()
        fl = info["flags"].cpu().numpy()
        for b in range(7):
            causes[b] += int(((fl[m & o_done] >> b) & 1).sum())
        alive &= ~o_done
        if not alive.any():
            break
    assert not alive.any(), "some episode never ended within 260 steps"
    assert causes[0] > 0 and causes[1] > 0, f"termination causes seen: {causes}"   # jackknife, out of map at least
    env.close()


def test_auto_reset_starts_a_fresh_episode(torch_mod):
    torch = torch_mod
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    n = 2048
    env = TruckTrailerVecEnv(n)
    env.reset(seed=5)
    full = torch.full((n,), 0.9, device="cuda")
    resets = 0
    for t in range(40):
        before = env.episode()
        obs, rew, done, _ = env.step(full, auto_reset=True)
        d = done.bool()
        after = env.episode()
        if d.any():
            resets += int(d.sum())
            assert (after["steps"][d] == 0).all()
            st = env.state[d]
            s = after["start"][d]
            assert torch.equal(st[:, 4], s[:, 0].float().double()) and torch.equal(st[:, 5], s[:, 1].float().double())
            assert (s[:, 0].abs() <= 27).all() and (s[:, 1] >= 0).all() and (s[:, 1] <= 27).all()
            fresh = env.observe(out=torch.empty_like(obs))
            assert torch.equal(fresh[d], obs[d])            # the obs row handed back is the new episode's first obs
            assert not torch.equal(after["start"][d], before["start"][d])
        assert (after["steps"][~d] == before["steps"][~d] + 1).all()
    assert resets > n // 2
    env.close()


def test_bench_size_properties_n65536(torch_mod):
    """N = 65536: lane independence (same input in every lane -> same output), determinism of a
    replayed launch, and obs invariants (unit sin/cos pairs, clipped ranges)."""
    torch = torch_mod
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    from conftest import load_group
    n = 65536
    t = load_group("f2_seeded.npz")["seed27"]
    env = TruckTrailerVecEnv(n)
    env.set_pose(np.tile(t["start"], (n, 1)))
    for k in range(len(t["actions"])):
        a = torch.full((n,), float(t["actions"][k]), device="cuda")
        obs, rew, done, info = env.step(a, auto_reset=False, info=True)
        assert (obs == obs[0]).all() and (info["comp"] == info["comp"][:, :1]).all() and (done == done[0]).all()
    assert np.abs(obs[0].cpu().numpy() - t["obs"][-1]).max() <= TOL
    assert abs(info["comp"][0, 0].item() - t["reward"][-1]) <= TOL and bool(done[0]) == bool(t["done"][-1])
    # random states: invariants
    env.reset(seed=99)
    for k in range(20):
        obs, rew, done, _ = env.step(env.random_actions(7, k), auto_reset=True)
    o = obs.double()
    for s, c in ((2, 3), (6, 7), (8, 9), (10, 11), (14, 15), (19, 20), (21, 22)):
        assert ((o[:, s] ** 2 + o[:, c] ** 2 - 1).abs() < 1e-6).all()
    assert (o[:, 16] >= 0).all() and (o[:, 16:19].abs() <= 1).all() and torch.isfinite(rew).all()
    # two handles, same seed, same actions -> bitwise equal
    env2 = TruckTrailerVecEnv(n)
    env.reset(seed=3); env2.reset(seed=3)
    for k in range(5):
        a = env.random_actions(11, k)
        o1, r1, d1, _ = env.step(a, auto_reset=True)
        o2, r2, d2, _ = env2.step(a, auto_reset=True)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    env.close(); env2.close()


def test_tail_block_and_unaligned_obs_pointer(torch_mod):
    """N not a multiple of the workgroup size and an obs pointer that is only 4-byte aligned."""
    torch = torch_mod
    n = 1000
    env, ora, _ = _oracle_pair(n, seed=8)
    backing = torch.zeros(n * 23 + 1, dtype=torch.float32, device="cuda")
    out = backing[1:].view(n, 23)
    a = env.random_actions(1, 0)
    obs, rew, done, _ = env.step(a, auto_reset=False, obs_out=out)
    o_obs, o_rew, o_done, _ = ora.step(a.cpu().numpy())
    assert obs.data_ptr() % 16 != 0
    assert np.abs(obs.cpu().numpy() - o_obs).max() <= TOL and backing[0].item() == 0.0
    env.close()


def test_exact_multiples_of_the_step_length(gpu_device):
    """int(d / 0.40096) at distances that sit ON a multiple of 0.40096 (reward_functionv1.py:38 with d0 + 1e-6,
    simv2.py:265 with d0): a quotient formed by multiplying with 1/0.40096 truncates differently from the division numpy
    does for about a third of the multiples.  Starts straight above the goal at such distances; the env's
    max_episode_steps and the reward's exploration / max-step thresholds must follow the C oracle (which divides)."""
    import torch
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    from oracle import c_oracle
    js = np.arange(40, 170)
    d = np.concatenate([js * 0.40096, js * 0.40096 - 1e-6, np.nextafter(js * 0.40096 - 1e-6, 0), np.nextafter(js * 0.40096, 1e9)])
    n = d.size
    start = np.stack([np.zeros(n), -30.0 + d, np.full(n, np.pi / 2)], 1)
    # the case exists: the two truncations disagree for some of these distances
    assert (((d + 1e-6) * (1.0 / 0.40096)).astype(int) != ((d + 1e-6) / 0.40096).astype(int)).any() or \
           ((d * (1.0 / 0.40096)).astype(int) != (d / 0.40096).astype(int)).any()
    env = TruckTrailerVecEnv(n)
    env.set_pose(start)
    ora = c_oracle.COracle(n)
    ora.place(start)
    want_max = np.array([e.max_steps for e in ora.envs])
    assert np.array_equal(env.episode()["max_episode_steps"].cpu().numpy(), want_max)
    assert np.array_equal(want_max, (d / 0.40096).astype(int) + 75)
    alive = np.ones(n, bool)
    i_exp, i_saf = c_oracle.INFO_KEYS.index("exploration_bonus"), c_oracle.INFO_KEYS.index("safety_penalty")
    seen_switch = False
    for t in range(200):
        a = np.zeros(n, np.float32)
        obs, rew, done, info = env.step(torch.from_numpy(a).cuda(), auto_reset=False, info=True)
        o_obs, o_rew, o_done, o_info = ora.step(a, nthreads=4)
        comp = info["comp"].cpu().numpy().T
        m = alive
        assert np.array_equal(comp[m][:, i_exp], o_info[m][:, i_exp]), t          # 4 / 2 / 0 switch at 0.5 and 0.8 of rmax
        assert np.array_equal(comp[m][:, i_saf], o_info[m][:, i_saf]), t
        assert (done.cpu().numpy().astype(bool)[m] == o_done[m]).all()
        seen_switch |= bool((o_info[m][:, i_exp] == 2.0).any())
        alive &= ~o_done
        if not alive.any():
            break
    assert seen_switch
    env.close()


def test_max_episode_steps_beyond_the_packed_counters_is_refused(gpu_device):
    """steps / max_episode_steps are 12-bit packed counters: 4096 and more is TT_EINVAL (C ABI) / ValueError (facade),
    and nothing is written."""
    import ctypes as C
    import torch
    from ddpg_trucktrailer_amd import _lib as L
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    env = TruckTrailerVecEnv(8)
    env.reset(seed=1)
    before = env.episode()["max_episode_steps"].clone()
    with pytest.raises(ValueError):
        env.set_max_steps([4096] * 8)
    bad = torch.tensor([10, 4096, 20, 30, 40, 50, 60, 70], dtype=torch.int32, device="cuda")
    rc = env.lib.tt_env_set_max_steps(env._h, None, 8, C.c_void_p(bad.data_ptr()), None)
    assert rc == L.TT_EINVAL and b"4096" in env.lib.tt_last_error(env._h)
    assert torch.equal(env.episode()["max_episode_steps"], before)
    env.set_max_steps([4095] * 8)
    assert (env.episode()["max_episode_steps"] == 4095).all()
    p = L.default_params(0)
    p.fixed_max_steps = 5000
    h = C.c_void_p()
    assert env.lib.tt_env_create(4, 0, C.byref(p), C.byref(h)) == L.TT_EINVAL
    env.close()
