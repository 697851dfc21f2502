#!/usr/bin/env python3
"""Generate tests/golden/f6_simv1.npz: the PINNABLE half of the reference's simv1 (BASELINE config 5).

Runs only in the build container (needs /root/reference).  The reference's simv1 imports a Dubins planner that is not in
the repository (`PythonRobotics...dubins_path_backward_planner`, simv1.py:13) and its step() ends in a 7-argument
RewardFunction call that raises TypeError (simv1.py:435).  Everything step() does BEFORE that call is the reference's own
code and runs unmodified: clip (:401-402), one solve_ivp(RK45) step with simv1's constants (:407-414), observation (:416),
step counter (:419), the four termination predicates (:422-430).  So F6 holds

  const/*   the constructor's constants (simv1.py:23-99)
  ode/*     kinematic_model derivatives on a seeded state / steering grid (simv1.py:180-214)
  obs/*     compute_observation known answers (simv1.py:101-179), goal = the fixed goal of :265-267
  jk/* oom/* ms/*   truth tables of check_jackknife / check_out_of_Map / check_max_steps_reached (:216-237)
  path/*    check_path_out_of_Map (:239-253) on hand-made paths handed to it by the stand-in planner
  traj*/    free-running trajectories through the reference's step() up to its reward call: state, observation
            (recomputed by the reference's compute_observation on the state step() left), the four flags, done

The stand-in planner module only returns the hand-made paths of path/*; nothing else in F6 goes through it (reset(), the
one caller of the real planner, is never called).  The reward call and the planner stay UNPINNED, and F6 says so
(`unpinned`).  Nothing here is imported by the product or by the tests."""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, install_gym_standin, versions  # noqa: E402

_PATHS = {}          # (sx, sy) -> (path_x, path_y, path_yaw): what the stand-in planner hands back


def install_planner_standin():
    names = ["PythonRobotics", "PythonRobotics.PathPlanning", "PythonRobotics.PathPlanning.DubinsPath",
             "PythonRobotics.PathPlanning.DubinsPath.dubins_path_backward_planner"]
    mods = [types.ModuleType(n) for n in names]
    for m, n in zip(mods, names):
        sys.modules[n] = m

    def plan_dubins_path_backward(sx, sy, syaw, gx, gy, gyaw, curvature):
        assert curvature == 1.0 / 6          # simv1.py:247
        return _PATHS[(float(sx), float(sy))]
    mods[-1].plan_dubins_path_backward = plan_dubins_path_backward


def import_simv1():
    install_gym_standin()
    install_planner_standin()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from truck_trailer_sim.simv1 import Truck_trailer_Env_1
    return Truck_trailer_Env_1


def fresh(Env):
    env = Env()
    env.goalx, env.goaly, env.goalyaw = 0, -30, np.deg2rad(90)      # what reset() assigns (simv1.py:265-267, 370)
    return env


def place(env, sx, sy, syaw):
    """reset()'s state construction (simv1.py:373-392) for a given start pose."""
    env.startx, env.starty, env.startyaw = float(sx), float(sy), float(syaw)
    x1 = env.startx + env.L2 * np.cos(env.startyaw)
    y1 = env.starty + env.L2 * np.sin(env.startyaw)
    env.state = np.array([env.startyaw, env.startyaw, x1, y1, env.startx, env.starty], dtype=np.float32)
    env.episode_steps = 0
    return env.compute_observation(env.state, np.deg2rad(0))


def run(env, obs0, actions, steps_before=0):
    env.episode_steps = int(steps_before)
    state0 = np.asarray(env.state, dtype=np.float64).copy()
    states, obs, flags, raised = [], [], [], []
    for a in actions:
        try:
            env.step(np.array([a], dtype=np.float32))
            raised.append(False)
        except TypeError as exc:                      # simv1.py:435: RewardFunction() takes 9 arguments, gets 7
            assert "RewardFunction" in str(exc) or "positional argument" in str(exc), exc
            raised.append(True)
        states.append(np.asarray(env.state, dtype=np.float64).copy())
        obs.append(np.asarray(env.compute_observation(env.state, np.clip(np.float32(a), env.min_steering_angle,
                                                                           env.max_steering_angle)), dtype=np.float32))
        f = [bool(env.jackknife), bool(env.out_of_map), bool(env.max_steps_reached), bool(env.goal_reached)]
        flags.append(f)
        if any(f):
            break
    n = len(states)
    return {"start": np.array([env.startx, env.starty, env.startyaw], dtype=np.float64), "state0": state0,
            "obs0": np.asarray(obs0, dtype=np.float32), "steps_before": np.int32(steps_before),
            "actions": np.asarray(actions[:n], dtype=np.float32), "states": np.array(states), "obs": np.array(obs),
            "flags": np.array(flags, dtype=np.bool_), "done": np.array([any(f) for f in flags], dtype=np.bool_),
            "reward_call_raised_typeerror": np.array(raised, dtype=np.bool_)}


def main():
    Env = import_simv1()
    env = fresh(Env)
    out = {"provenance": np.array(
        "reference truck_trailer_sim/simv1.py imported with stand-in modules for gym and for the un-vendored Dubins planner "
        "(which only hands back the hand-made paths of path/*); trajectories = the reference's step() up to the "
        f"RewardFunction call that raises TypeError (simv1.py:435) ({versions()})"),
        "unpinned": np.array("the reward call of simv1.step (simv1.py:435, TypeError in the reference) and the Dubins planner "
                             "behind reset() / check_path_out_of_Map (not in the repository)"),
        "flag_keys": np.array(["jackknife", "out_of_map", "max_steps_reached", "goal_reached"])}
    # ---- constants
    for k in ("min_map_x", "min_map_y", "max_map_x", "max_map_y", "L1", "L2", "hitch_offset", "v1x", "dt", "time",
              "max_hitch_angle", "min_steering_angle", "max_steering_angle", "max_expected_distance", "observation_dim",
              "max_episode_steps", "position_threshold", "orientation_threshold"):
        out[f"const/{k}"] = np.float64(getattr(env, k))
    out["const/action_low"] = np.asarray(env.action_space.low, dtype=np.float32)
    out["const/action_high"] = np.asarray(env.action_space.high, dtype=np.float32)
    out["const/goal"] = np.array([env.goalx, env.goaly, env.goalyaw], dtype=np.float64)
    # ---- ODE right-hand side
    rng = np.random.RandomState(61)
    m = 256
    x = np.stack([rng.uniform(-np.pi, 2 * np.pi, m), rng.uniform(-np.pi, 2 * np.pi, m), rng.uniform(-40, 40, m),
                  rng.uniform(-40, 40, m), rng.uniform(-40, 40, m), rng.uniform(-40, 40, m)], axis=1)
    x[:8, 1] = x[:8, 0]                               # straight rigs
    delta = rng.uniform(-1, 1, m) * np.pi / 4
    delta[:4] = [0.0, np.radians(45), np.radians(-45), 1e-3]
    xd = []
    for xi, di in zip(x, delta):
        env.steering_angle = float(di)
        xd.append(env.kinematic_model(0.0, xi, float(di)))
    out["ode/x"], out["ode/delta"], out["ode/xd"] = x, delta, np.array(xd, dtype=np.float64)
    # ---- observation known answers (f64 states and freshly-reset f32 states)
    st = np.stack([rng.uniform(-np.pi, 2 * np.pi, m), rng.uniform(-np.pi, 2 * np.pi, m), rng.uniform(-45, 45, m),
                   rng.uniform(-45, 45, m), rng.uniform(-45, 45, m), rng.uniform(-45, 45, m)], axis=1)
    steer = rng.uniform(-1, 1, m) * np.pi / 4
    steer[:3] = 0.0
    ob = [env.compute_observation(s, d) for s, d in zip(st, steer)]
    st[-1, 4:6] = [env.goalx, env.goaly]              # trailer exactly on the goal: atan2(0, 0)
    ob[-1] = env.compute_observation(st[-1], steer[-1])
    out["obs/state"], out["obs/steer"], out["obs/out"] = st, steer, np.array(ob, dtype=np.float32)
    # ---- truth tables
    r90 = np.deg2rad(90)
    th = np.array([0.0, 1.0, r90 - 1e-9, r90, np.nextafter(r90, 4), r90 + 1e-9, -r90, -np.nextafter(r90, 4), 2.0, -3.0, 7.0])
    base = rng.uniform(-3, 3, len(th))
    out["jk/psi1"], out["jk/psi2"] = base + th, base
    out["jk/out"] = np.array([env.check_jackknife(a, b) for a, b in zip(base + th, base)], dtype=np.bool_)
    pts = []
    for v in (-40.0, np.nextafter(-40.0, -50), -40.000001, 40.0, np.nextafter(40.0, 50), 40.000001, 0.0, 39.999999):
        for slot in range(4):
            p = [1.0, -2.0, 3.0, -4.0]
            p[slot] = v
            pts.append(p)
    pts = np.array(pts, dtype=np.float64)
    out["oom/xy"] = pts
    out["oom/out"] = np.array([env.check_out_of_Map(*p) for p in pts], dtype=np.bool_)
    ms = np.array([0, 1, 150, 298, 299, 300, 301, 4000], dtype=np.int64)
    out["ms/step"] = ms
    out["ms/out"] = np.array([env.check_max_steps_reached(int(s)) for s in ms], dtype=np.bool_)
    # ---- check_path_out_of_Map on hand-made paths
    t = np.linspace(0.0, 1.0, 41)
    paths = [
        ("inside_line", 10 + 0 * t, 30 - 60 * t),
        ("touches_right_edge", 30 + 10 * np.sin(np.pi * t), 20 - 50 * t),               # max x == 40.0 exactly: inside (:250 uses >)
        ("one_point_past_right_edge", np.where(np.arange(41) == 20, np.nextafter(40.0, 50), 30.0), 20 - 50 * t),
        ("leaves_bottom", 5 + 0 * t, -20 - 25 * t),
        ("leaves_left_then_returns", -35 - 8 * np.sin(np.pi * t), 10 - 40 * t),
        ("starts_outside_top", 0 * t, 40.5 - 70 * t),
        ("corner_inside", 40 - 80 * t, -40 + 0 * t),                                     # runs along y == -40: inside
        ("single_point", np.array([12.0]), np.array([-3.0])),
    ]
    names, res = [], []
    for i, (name, px, py) in enumerate(paths):
        px, py = np.asarray(px, dtype=np.float64), np.asarray(py, dtype=np.float64)
        key = (float(1000 + i), float(-1000 - i))
        _PATHS[key] = (px, py, np.zeros_like(px))
        res.append(env.check_path_out_of_Map(key[0], key[1], 0.3, env.goalx, env.goaly, env.goalyaw))
        out[f"path/{name}/x"], out[f"path/{name}/y"] = px, py
        names.append(name)
    out["path/names"], out["path/out"] = np.array(names), np.array(res, dtype=np.bool_)
    # ---- trajectories through step() up to the reward call
    R45 = np.float32(np.pi / 4)
    sc = [("random_far", (-20.0, 25.0, 1.0), rng.uniform(-0.3, 0.3, 150).astype(np.float32), 0),
          ("random_wide", (12.0, 18.0, 4.0), (rng.uniform(-1, 1, 150) * np.pi / 4).astype(np.float32), 0),
          ("jackknife_full_lock", (-5.0, 10.0, np.pi / 2), np.full(200, R45, np.float32), 0),
          ("jackknife_oversaturated", (8.0, 0.0, 1.2), np.full(200, -1.5, np.float32), 0),
          ("out_of_map_right", (36.0, 0.0, np.pi), np.zeros(60, np.float32), 0),
          ("out_of_map_truck_first", (30.5, 5.0, 0.0), np.zeros(40, np.float32), 0),
          ("goal_straight", (0.0, -24.0, np.pi / 2), np.zeros(60, np.float32), 0),
          ("past_goal_keeps_running", (3.0, -29.5, np.pi / 2), np.zeros(25, np.float32), 0),     # simv1 has no goal-passed end
          ("max_steps_cap", (-15.0, 30.0, 1.3), (0.1 * np.sin(np.arange(40) * 0.3)).astype(np.float32), 290),
          ("sine_long", (20.0, 30.0, 2.0), (0.3 * np.sin(np.arange(299) * 0.11)).astype(np.float32), 0)]
    # a closed-loop circle that survives to the fixed cap: the hitch angle is held at 0.5 rad by feedback computed on the
    # REFERENCE's own state (the recorded actions are then plain data); ends by max_steps_reached at step 300 exactly
    e = fresh(Env)
    place(e, 18.0, -14.0, np.pi / 2)
    acts = []
    for _ in range(300):
        th = float(e.state[0] - e.state[1])
        d = np.float32(np.clip(np.arctan((e.L1 / e.L2) * np.sin(th) - (e.L1 * 2.0 / e.v1x) * (th - 0.5)), -0.78, 0.78))
        acts.append(d)
        try:
            e.step(np.array([d], dtype=np.float32))
        except TypeError:
            pass
        assert not (e.jackknife or e.out_of_map or e.goal_reached), (len(acts), e.state)
    assert e.max_steps_reached and len(acts) == 300
    sc.append(("circle_to_the_300_step_cap", (18.0, -14.0, np.pi / 2), np.array(acts, dtype=np.float32), 0))
    tn = []
    for name, start, actions, before in sc:
        e = fresh(Env)
        obs0 = place(e, *start)
        tr = run(e, obs0, actions, before)
        assert tr["reward_call_raised_typeerror"].all(), "the reference's simv1 reward call is expected to raise"
        for k, v in tr.items():
            out[f"traj/{name}/{k}"] = v
        tn.append(name)
        print(f"F6 {name}: {len(tr['actions'])} steps, flags at the end {tr['flags'][-1].astype(int)}")
    out["traj/names"] = np.array(tn)
    path = os.path.join(HERE, "f6_simv1.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: L1 {out['const/L1']}, L2 {out['const/L2']}, cap {out['const/max_episode_steps']}, "
          f"{m} ODE rows, {m} observation rows, {len(names)} paths {res}, {len(tn)} trajectories")


if __name__ == "__main__":
    main()
