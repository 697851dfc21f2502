"""CPU: pin the oracle (oracle/) to the golden fixtures generated from the real reference.

Tolerance is BASELINE.json's: 1e-5 on state, observation and reward, free-running (the
oracle's own state is fed back, never the fixture's).  The scipy twin is expected to be
far tighter (it repeats the reference's arithmetic), and that is asserted too."""
import numpy as np
import pytest

from conftest import all_trajectories, load_group, needs_raw_state, GOLDEN
from oracle import c_oracle
from oracle.simv2_twin import FixedStepTwin, Simv2Twin, observation, Params

TOL = 1e-5
FLAG_KEYS = ("jackknife", "out_of_map", "max_steps_reached", "goal_reached", "goal_passed", "excessive_backward")


def replay_twin(cls, t):
    env = cls()
    obs0 = env.set_pose(t["start"], goal=tuple(t["goal"]), L2=float(t["L2"]),
                        state=t["state0"] if needs_raw_state(t) else None, max_steps=int(t["max_episode_steps"]))
    out = dict(obs0=obs0, states=[], obs=[], reward=[], done=[], viol=[], flags=[], info=[])
    for a in t["actions"]:
        o, r, d, info = env.step(np.array([a], np.float32))
        out["states"].append(env.state.copy()); out["obs"].append(o); out["reward"].append(r); out["done"].append(d)
        out["viol"].append(info["violation"]); out["flags"].append([env.flags[k] for k in FLAG_KEYS]); out["info"].append(info)
    return {k: (np.array(v) if k not in ("obs0", "info") else v) for k, v in out.items()}


@pytest.mark.parametrize("t", all_trajectories())
def test_scipy_twin_reproduces_reference(t):
    r = replay_twin(Simv2Twin, t)
    assert np.abs(r["obs0"] - t["obs0"]).max() <= 2e-7
    assert np.abs(r["states"] - t["states"]).max() <= 1e-12
    assert np.abs(r["obs"] - t["obs"]).max() <= 1e-7
    assert np.abs(r["reward"] - t["reward"]).max() <= 1e-9
    assert (r["done"] == t["done"]).all() and (r["viol"] == t["violation"]).all()
    assert (r["flags"] == t["flags"]).all()
    for key, col in (("progress_reward", 2), ("heading_reward", 3), ("orientation_reward", 4), ("staged_success", 5),
                     ("safety_penalty", 6), ("exploration_bonus", 7), ("final_success_bonus", 8),
                     ("backward_penalty", 9), ("smoothness_penalty", 10), ("cumulative_backward", 11)):
        got = np.array([float(i[key]) for i in r["info"]])
        assert np.abs(got - t["info"][:, col]).max() <= 1e-9, key


@pytest.mark.parametrize("t", all_trajectories())
def test_fixed_step_twin_within_tolerance(t):
    r = replay_twin(FixedStepTwin, t)
    assert np.abs(r["states"] - t["states"]).max() <= TOL
    assert np.abs(r["obs"] - t["obs"]).max() <= TOL
    assert np.abs(r["reward"] - t["reward"]).max() <= TOL
    assert (r["done"] == t["done"]).all() and (r["viol"] == t["violation"]).all() and (r["flags"] == t["flags"]).all()


def replay_c(t):
    o = c_oracle.COracle(1)
    obs0 = o.place(t["start"], goal=t["goal"], L2=float(t["L2"]))[0]
    if needs_raw_state(t):
        o.set_state(0, t["state0"])
        obs0 = o.observe(0)
    o.set_max_steps(0, int(t["max_episode_steps"]))
    out = dict(obs0=obs0, states=[], obs=[], reward=[], done=[], viol=[], flags=[], info=[])
    for a in t["actions"]:
        ob, r, d, info = o.step([a])
        out["states"].append(o.state()[0]); out["obs"].append(ob[0]); out["reward"].append(r[0]); out["done"].append(d[0])
        out["viol"].append(o.violation()[0]); out["flags"].append([(o.flags()[0] >> b) & 1 for b in range(6)])
        out["info"].append(info[0])
    return {k: np.array(v) for k, v in out.items()}


@pytest.mark.parametrize("t", all_trajectories())
def test_c_oracle_within_tolerance(t):
    r = replay_c(t)
    assert np.abs(r["obs0"] - t["obs0"]).max() <= TOL
    assert np.abs(r["states"] - t["states"]).max() <= TOL
    assert np.abs(r["obs"] - t["obs"]).max() <= TOL
    assert np.abs(r["reward"] - t["reward"]).max() <= TOL
    assert (r["done"] == t["done"]).all()
    assert (r["viol"] == t["violation"]).all()
    assert (r["flags"].astype(bool) == t["flags"]).all()
    # components: fixture info columns are [total, distance(=0), progress, heading, orientation, staged, safety,
    # exploration, final, backward, smoothness, cum_backward, budget, excess]
    fix = t["info"][:, [0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]]
    assert np.abs(r["info"] - fix).max() <= TOL


def _f7():
    return [pytest.param(t, id=f"f7-{n}") for n, t in load_group("f7_episode_steps.npz").items()]


@pytest.mark.parametrize("t", _f7())
def test_episode_steps_written_by_the_caller(t):
    """Fixture F7: `env.episode_steps = k` between steps of the reference (episode_replay_collectorv2.py:258-269) moves the step
    counter alone; the reward carry keeps its own step count.  Twin (bit-faithful) and C oracle, free-running."""
    writes = {int(i): int(v) for i, v in t["writes"]}
    env = Simv2Twin()
    env.set_pose(t["start"], goal=tuple(t["goal"]), L2=float(t["L2"]), state=t["state0"] if needs_raw_state(t) else None,
                 max_steps=int(t["max_episode_steps"]))
    o = c_oracle.COracle(1)
    o.place(t["start"], goal=t["goal"], L2=float(t["L2"]))
    if needs_raw_state(t):
        o.set_state(0, t["state0"])
    o.set_max_steps(0, int(t["max_episode_steps"]))
    for k, a in enumerate(t["actions"]):
        if k in writes:
            env.episode_steps = writes[k]
            o.envs[0].steps = writes[k]
        ob, r, d, info = env.step(np.array([a], np.float32))
        assert np.abs(env.state - t["states"][k]).max() <= 1e-12 and np.abs(ob - t["obs"][k]).max() <= 1e-7, k
        assert abs(r - t["reward"][k]) <= 1e-9 and d == t["done"][k] and info["violation"] == t["violation"][k], k
        assert [env.flags[f] for f in FLAG_KEYS] == t["flags"][k].tolist(), k
        assert abs(info["movement_budget"] - t["info"][k, 12]) <= 1e-12 and abs(info["exploration_bonus"] - t["info"][k, 7]) <= 1e-12
        cob, cr, cd, cinfo = o.step([a])
        assert np.abs(o.state()[0] - t["states"][k]).max() <= TOL and np.abs(cob[0] - t["obs"][k]).max() <= TOL, k
        assert abs(cr[0] - t["reward"][k]) <= TOL and bool(cd[0]) == bool(t["done"][k]) and o.violation()[0] == t["violation"][k], k
        assert [(o.flags()[0] >> b) & 1 for b in range(6)] == t["flags"][k].astype(int).tolist(), k
        assert np.abs(cinfo[0] - t["info"][k, [0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]]).max() <= TOL, k
    # the fixture really separates the two counters: somewhere the budget differs from what the step counter would give
    if len(writes):
        steps, sep = 0, False
        for k in range(len(t["actions"])):
            steps = writes.get(k, steps) + 1
            sep |= abs(5.0 * min(1.0, steps / 50) - t["info"][k, 12]) > 1e-9
        assert sep


def test_golden_episode_recording_is_the_authority():
    """The fixture's replayed states/rewards agree with what the reference's author recorded."""
    t = load_group("f1_golden_episode.npz")["golden"]
    assert t["recorded_states"].shape == (194, 6) and t["actions"].shape == (193,)
    assert np.abs(t["states"] - t["recorded_states"][1:]).max() <= 1e-12
    assert np.abs(t["reward"] - t["recorded_info"][:, 0]).max() <= 2e-6
    assert abs(t["recorded_info"][:, 0].sum() - 4792.9998) < 1e-3
    assert t["recorded_success"][-1] and t["done"][-1] and not t["done"][:-1].any()


def test_observation_known_answers():
    z = np.load(f"{GOLDEN}/f4_observation.npz", allow_pickle=False)
    p = Params()
    co = c_oracle.COracle(1)
    for i in range(len(z["states"])):
        g = z["goals"][i]
        goal = tuple(g)  # np.float64 scalars, exactly what make_golden.py handed the reference
        o64 = observation(z["states"][i].astype(np.float64), z["steer"][i], goal, p)
        o32 = observation(z["states"][i].astype(np.float32), np.deg2rad(0), goal, p)
        assert np.abs(o64 - z["obs_f64_state"][i]).max() <= 1e-7
        assert np.abs(o32 - z["obs_f32_state_steer0"][i]).max() <= 2e-7
        co.place([0.0, 0.0, 0.0], goal=g)
        co.set_state(0, z["states"][i])
        assert np.abs(co.observe(0, float(z["steer"][i])) - z["obs_f64_state"][i]).max() <= 1e-6


def test_reset_seed_reproduces_numpy_legacy_stream():
    """reset(seed) = np.random.seed(seed) then three uniforms in the order x, y, yaw (simv2.py:331-333)."""
    for name, t in load_group("f2_seeded.npz").items():
        env = Simv2Twin()
        obs0, info = env.reset(seed=int(t["seed"]))
        assert info == {}
        assert np.allclose([env.startx, env.starty, env.startyaw], t["start"], rtol=0, atol=0)
        assert env.max_episode_steps == int(t["max_episode_steps"])
        assert np.array_equal(env.state.astype(np.float64), t["state0"])
        assert np.abs(obs0 - t["obs0"]).max() <= 2e-7


def test_c_oracle_threads_agree():
    n = 257
    rng = np.random.RandomState(5)
    start = np.stack([rng.uniform(-27, 27, n), rng.uniform(0, 27, n), rng.uniform(np.pi / 4, 2 * np.pi / 3, n)], 1)
    a, b = c_oracle.COracle(n), c_oracle.COracle(n)
    a.place(start); b.place(start)
    for _ in range(30):
        act = rng.uniform(-1, 1, n).astype(np.float32)
        ra, rb = a.step(act, nthreads=1), b.step(act, nthreads=4)
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)
