"""Start-pose generation of the simv1 variant (truck_trailer_sim/simv1.py:239-282, 367-396).

simv1 draws a start pose uniformly over the whole map and over 0..360 deg of heading (python's `random`
module, seed ignored), rejects it when it is closer than 15 m to the goal or when the Dubins path
(curvature 1/6) from start to goal leaves the map, and repeats (up to 1000 attempts).

PARITY UNPINNED: the reference imports `plan_dubins_path_backward` from a local, un-vendored
PythonRobotics file that is not part of the repository, so neither its planner nor simv1 itself can run.
What is implemented here is the textbook shortest Dubins path (six words LSL, RSR, LSR, RSL, RLR, LRL;
Shkel & Lumelsky 2001), and "backward" is read as: the vehicle reverses along the path, i.e. its direction
of travel is heading + pi at both ends.  Both are assumptions, stated here and in DESIGN.md.

This is host logic (rejection sampling with a geometric test); the poses it produces are uploaded as the
env's reset pool (`TruckTrailerVecEnv.set_reset_pool`), from which explicit and in-kernel resets draw."""
import math
import random

import numpy as np

TWO_PI = 2.0 * math.pi


def _mod2pi(x):
    r = x - TWO_PI * math.floor(x / TWO_PI)
    return 0.0 if TWO_PI - r < 1e-9 else r      # -1e-17 must wrap to 0, not to a full extra turn


def _words(alpha, beta, d):
    """(t, p, q, word) candidates in units of the turning radius, start (0,0,alpha) -> (d,0,beta)."""
    sa, sb, ca, cb, cab = math.sin(alpha), math.sin(beta), math.cos(alpha), math.cos(beta), math.cos(alpha - beta)
    out = []
    p2 = 2 + d * d - 2 * cab + 2 * d * (sa - sb)
    if p2 < 1e-18 and p2 > -1e-12:   # both circles coincide: one left arc (atan2(0, 0) below would be arbitrary)
        out.append((_mod2pi(beta - alpha), 0.0, 0.0, "LSL"))
    elif p2 >= 0:
        tmp = math.atan2(cb - ca, d + sa - sb)
        out.append((_mod2pi(-alpha + tmp), math.sqrt(p2), _mod2pi(beta - tmp), "LSL"))
    p2 = 2 + d * d - 2 * cab + 2 * d * (sb - sa)
    if p2 < 1e-18 and p2 > -1e-12:
        out.append((_mod2pi(alpha - beta), 0.0, 0.0, "RSR"))
    elif p2 >= 0:
        tmp = math.atan2(ca - cb, d - sa + sb)
        out.append((_mod2pi(alpha - tmp), math.sqrt(p2), _mod2pi(-beta + tmp), "RSR"))
    p2 = -2 + d * d + 2 * cab + 2 * d * (sa + sb)
    if p2 >= 0:
        p = math.sqrt(p2)
        tmp = math.atan2(-ca - cb, d + sa + sb) - math.atan2(-2.0, p)
        out.append((_mod2pi(-alpha + tmp), p, _mod2pi(-_mod2pi(beta) + tmp), "LSR"))
    p2 = d * d - 2 + 2 * cab - 2 * d * (sa + sb)
    if p2 >= 0:
        p = math.sqrt(p2)
        tmp = math.atan2(ca + cb, d - sa - sb) - math.atan2(2.0, p)
        out.append((_mod2pi(alpha - tmp), p, _mod2pi(beta - tmp), "RSL"))
    tmp = (6.0 - d * d + 2 * cab + 2 * d * (sa - sb)) / 8.0
    if abs(tmp) <= 1:
        p = _mod2pi(TWO_PI - math.acos(tmp))
        t = _mod2pi(alpha - math.atan2(ca - cb, d - sa + sb) + p / 2.0)
        out.append((t, p, _mod2pi(alpha - beta - t + p), "RLR"))
    tmp = (6.0 - d * d + 2 * cab + 2 * d * (-sa + sb)) / 8.0
    if abs(tmp) <= 1:
        p = _mod2pi(TWO_PI - math.acos(tmp))
        t = _mod2pi(-alpha - math.atan2(ca - cb, d + sa - sb) + p / 2.0)
        out.append((t, p, _mod2pi(_mod2pi(beta) - alpha - t + p), "LRL"))
    return out


def plan_dubins_path(sx, sy, syaw, gx, gy, gyaw, curvature, step_size=0.1):
    """Shortest Dubins path; returns (x[], y[], yaw[], word, total_length)."""
    dx, dy = gx - sx, gy - sy
    dist = math.hypot(dx, dy)
    theta = math.atan2(dy, dx)
    d = dist * curvature
    best = min(_words(_mod2pi(syaw - theta), _mod2pi(gyaw - theta), d), key=lambda w: w[0] + w[1] + w[2])
    t, p, q, word = best
    r = 1.0 / curvature
    xs, ys, yaws = [sx], [sy], [syaw]
    x, y, yaw = sx, sy, syaw
    for seg_len, kind in zip((t, p, q), word):
        length = seg_len * r                          # metres along this segment
        if length < 1e-12:
            continue
        n = max(1, int(math.ceil(length / step_size)))
        s = np.linspace(length / n, length, n)
        if kind == "S":
            px, py, pyaw = x + s * math.cos(yaw), y + s * math.sin(yaw), np.full(n, yaw)
        else:
            sign = 1.0 if kind == "L" else -1.0
            ang = yaw + sign * s * curvature
            px = x + sign * r * (np.sin(ang) - math.sin(yaw))
            py = y - sign * r * (np.cos(ang) - math.cos(yaw))
            pyaw = ang
        xs.extend(px.tolist()); ys.extend(py.tolist()); yaws.extend(pyaw.tolist())
        x, y, yaw = float(px[-1]), float(py[-1]), float(pyaw[-1])
    return np.array(xs), np.array(ys), np.array(yaws), word, (t + p + q) * r


def plan_dubins_path_backward(sx, sy, syaw, gx, gy, gyaw, curvature, step_size=0.1):
    """Path a REVERSING vehicle follows: travel direction = heading + pi (assumption, see module docstring).
    Returns (path_x, path_y, path_yaw) with path_yaw the vehicle heading along the path."""
    x, y, yaw, _, _ = plan_dubins_path(sx, sy, syaw + math.pi, gx, gy, gyaw + math.pi, curvature, step_size)
    return x, y, yaw - math.pi


def points_out_of_map(px, py, lo=-40.0, hi=40.0):
    """The in-map test of simv1.py:249-253 (== check_out_of_Map, :225-237, per point): a point ON an edge is inside.
    Pinned to the reference by fixture F6 (path/*, oom/*)."""
    px, py = np.asarray(px, dtype=np.float64), np.asarray(py, dtype=np.float64)
    return bool(((px < lo) | (px > hi) | (py < lo) | (py > hi)).any())


def path_out_of_map(sx, sy, syaw, gx, gy, gyaw, lo=-40.0, hi=40.0, curvature=1.0 / 6):
    """simv1.py:239-253."""
    px, py, _ = plan_dubins_path_backward(sx, sy, syaw, gx, gy, gyaw, curvature)
    return points_out_of_map(px, py, lo, hi)


def generate_valid_random_pose(rng=random, goal=(0.0, -30.0, math.pi / 2), lo=-40.0, hi=40.0, max_attempts=1000):
    """simv1.py:255-282: uniform over the map and 0..2pi, >= 15 m from the goal, Dubins path inside the map.
    Returns None when no attempt succeeds (the reference then fails to unpack)."""
    gx, gy, gyaw = goal
    for _ in range(max_attempts):
        sx = rng.uniform(lo, hi)
        sy = rng.uniform(lo, hi)
        syaw = rng.uniform(0.0, TWO_PI)
        if math.hypot(gx - sx, gy - sy) < 15:
            continue
        if not path_out_of_map(sx, sy, syaw, gx, gy, gyaw, lo, hi):
            return (sx, sy, syaw)
    return None


def generate_pose_pool(m, seed=0, **kw):
    """m valid start poses [m,3] f64 for TruckTrailerVecEnv.set_reset_pool."""
    rng = random.Random(seed)
    out = []
    while len(out) < m:
        pose = generate_valid_random_pose(rng, **kw)
        if pose is not None:
            out.append(pose)
    return np.array(out, dtype=np.float64)
