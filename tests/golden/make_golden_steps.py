#!/usr/bin/env python3
"""Fixture F7: `env.episode_steps` WRITTEN BY THE CALLER, captured from the REAL reference.

`episode_steps` is a public attribute of the reference's env (truck_trailer_sim/simv2.py:94, 523-530) that a caller
writes (DDPG/episode_replay_collectorv2.py:258-269: `state`, start / goal, `max_episode_steps = compute_max_steps()`,
`episode_steps = 0` on an env that was never reset).  Such a write moves the step counter alone: the reward carry
(`reward_state`: previous distance, backward-movement step count, stage latches) stays -- or stays None.  F7 pins both
halves of that: counter-dependent terms (exploration tiers, the max-step penalty, `max_steps_reached`) follow the written
value, carry-dependent terms (movement budget, progress window, smoothness reference) do not.

Runs only in the build container (needs /root/reference); writes tests/golden/f7_episode_steps.npz.  Trajectory fields
are make_golden.py's plus `writes` [[i, value], ...]: `env.episode_steps = value` right before action i is stepped, and
`no_reset` (the env object was never reset: the collector's pattern)."""
import os
import sys

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def run_with_writes(env, obs0, actions, writes):
    """make_golden.run_trajectory with attribute writes between steps; one env.step per action."""
    pieces, at = [], dict(writes)
    state0 = np.asarray(env.state, dtype=np.float64).copy()
    for i, a in enumerate(actions):
        if i in at:
            env.episode_steps = int(at[i])
        t = mg.run_trajectory(env, obs0, [a])
        pieces.append(t)
        if t["done"][-1]:
            break
    out = dict(pieces[0])
    for k in ("actions", "states", "obs", "reward", "done", "info", "violation", "success", "flags", "carry"):
        out[k] = np.concatenate([p[k] for p in pieces], axis=0)
    out["state0"] = state0
    out["max_episode_steps"] = np.int32(env.max_episode_steps)
    out["writes"] = np.array([[i, v] for i, v in writes if i < len(out["actions"])], dtype=np.int32).reshape(-1, 2)
    return out


def main():
    Env = mg.import_simv2()
    trajs, names = [], []

    def add(name, t, no_reset=False):
        t["no_reset"] = np.bool_(no_reset)
        trajs.append(t); names.append(name)
        print(f"F7 {name}: len {len(t['actions'])}, writes {t['writes'].tolist()}, done {t['done'][-1]}, "
              f"flags {t['flags'][-1].astype(int)}, max_steps {int(t['max_episode_steps'])}, return {t['reward'].sum():.3f}")

    wig = lambda n, w=0.21, amp=0.3: (amp * np.sin(np.arange(n) * w)).astype(np.float32)

    # 1. fresh episode (reward_state None) started at a late step count: the carry is new (budget 0.1) while the exploration
    #    tier is already 0 and the cap is a few steps away
    env = Env()
    obs0 = mg.override_pose(env, (-4.0, 22.0, 1.75))
    cap = int(env.max_episode_steps)
    add("fresh_late_count", run_with_writes(env, obs0, wig(60), [(0, cap - 9)]))

    # 2. the counter set back to 0 in the middle of an episode: tiers restart, the carry (budget, window, first steering) goes on
    env = Env()
    obs0 = mg.override_pose(env, (10.0, 20.0, 1.4))
    add("mid_episode_zero", run_with_writes(env, obs0, wig(120, 0.17, 0.1), [(30, 0)]))

    # 3. two jumps: forward past the 0.5 / 0.8 tiers and the reward's own max-step penalty, then back
    env = Env()
    obs0 = mg.override_pose(env, (-15.0, 24.0, 1.2), max_steps=400)      # (the env's own cap out of the way: the reward keeps its own)
    rmax = int((np.sqrt((0.0 + 15.0) ** 2 + (-30.0 - 24.0) ** 2) + 1e-6) / 0.40096) + 75
    add("jump_forward_and_back", run_with_writes(env, obs0, wig(90, 0.13, 0.1), [(20, rmax - 5), (32, 3)]))

    # 4. backward-moving start (cumulative backward movement above its budget) with the counter written: the budget follows the
    #    carry's own step count, not the counter
    env = Env()
    obs0 = mg.override_pose(env, (0.0, -20.0, -np.pi / 2))
    add("backward_budget_vs_counter", run_with_writes(env, obs0, np.zeros(40, np.float32), [(0, 60), (10, 0)]))

    # 5. the collector's pattern (episode_replay_collectorv2.py:258-269): an env that was NEVER reset, `state`, start / goal,
    #    max_episode_steps = compute_max_steps(), episode_steps = 0, then the recorded actions
    f1 = np.load(os.path.join(HERE, "f1_golden_episode.npz"), allow_pickle=False)
    env = Env()
    env.state = f1["golden/state0"].astype(np.float32)
    env.startx, env.starty, env.startyaw = (float(x) for x in f1["golden/start"])
    env.goalx, env.goaly, env.goalyaw = (float(x) for x in f1["golden/goal"])
    env.max_episode_steps = env.compute_max_steps()
    env.episode_steps = 0
    obs0 = env.compute_observation(env.state, 0.0)
    t = run_with_writes(env, obs0, f1["golden/actions"], [])
    t["raw_state_override"] = np.bool_(True)
    add("collector_pattern_no_reset", t, no_reset=True)
    d = np.abs(t["states"] - f1["golden/states"]).max()
    print(f"   collector pattern vs F1 (reset + override): state diff {d:.2e}, reward diff "
          f"{np.abs(t['reward'] - f1['golden/reward']).max():.2e}")

    mg.save_group(os.path.join(HERE, "f7_episode_steps.npz"), trajs, names,
                  "reference simv2 with env.episode_steps written by the caller between steps "
                  f"(episode_replay_collectorv2.py:258-269) ({mg.versions()})")


if __name__ == "__main__":
    main()
