"""Grid evaluation of a policy: the data generation of DDPG/heatmap.py:39-193 on the vector env.

heatmap.py runs, for every 2 m cell of the map and 5 trials per cell, one deterministic-policy episode from a
start pose at the cell (yaw ~ U(45,120) deg, trailer length L2 ~ U(5,7): heatmap.py:84-89), one after the
other (40 x 35 x 5 = 7000 episodes, "may take several hours").  Here every episode is one lane of ONE
TruckTrailerVecEnv: pose override for all lanes, then actor -> step until every lane is done.  Returns the
same seven objects as `generate_heatmap_data` (no plotting)."""
import numpy as np
import torch

from ddpg_trucktrailer_amd import _lib as L
from ddpg_trucktrailer_amd import fused
from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv


def _violation_label(flags, success):
    """heatmap.py:157-171 priority: success, jackknife, out_of_map, goal_passed, max_steps, other."""
    if success or flags & L.F_GOAL_REACHED:
        return 'success'
    if flags & L.F_JACKKNIFE:
        return 'jackknife'
    if flags & L.F_OUT_OF_MAP:
        return 'out_of_map'
    if flags & L.F_GOAL_PASSED:
        return 'goal_passed'
    if flags & L.F_MAX_STEPS:
        return 'max_steps'
    return 'other_failure'


@torch.no_grad()
def generate_heatmap_data(actor, grid_resolution=2.0, trials_per_cell=5, map_x_range=(-40, 40), map_y_range=(-30, 40),
                          start_orientation_range_deg=(45, 120), goal_pose=(0.0, -30.0, 90.0), L2_range=(5, 7),
                          seed=66, device=None, max_steps_cap=1024):
    x_coords = np.arange(map_x_range[0], map_x_range[1], grid_resolution)
    y_coords = np.arange(map_y_range[0], map_y_range[1], grid_resolution)
    ny, nx, nt = len(y_coords), len(x_coords), trials_per_cell
    n = ny * nx * nt
    rng = np.random.RandomState(seed)                      # heatmap.py seeds numpy once (set_seed(SEED))
    yaw_deg = rng.uniform(start_orientation_range_deg[0], start_orientation_range_deg[1], n)
    l2 = rng.uniform(L2_range[0], L2_range[1], n)
    iy, ix, it = np.meshgrid(np.arange(ny), np.arange(nx), np.arange(nt), indexing="ij")
    start = np.stack([x_coords[ix.ravel()], y_coords[iy.ravel()], np.deg2rad(yaw_deg)], 1)
    goal = np.tile(np.array([goal_pose[0], goal_pose[1], np.deg2rad(goal_pose[2])]), (n, 1))

    env = TruckTrailerVecEnv(n, device=device)
    dev = env.device
    obs = env.set_pose(start, goal=goal, L2=l2)
    high = float(np.float32(np.pi / 4))
    score = torch.zeros(n, dtype=torch.float64, device=dev)
    finished = torch.zeros(n, dtype=torch.bool, device=dev)
    end_flags = torch.zeros(n, dtype=torch.uint8, device=dev)
    end_xy = torch.zeros((n, 2), dtype=torch.float64, device=dev)
    first = torch.arange(0, n, nt, device=dev)             # lane of trial 0 of every cell (its trajectory is kept)
    traj = [env.state[first][:, 4:6].cpu().numpy()]
    traj_len = torch.ones(len(first), dtype=torch.int64, device=dev)
    # the reference-shaped actor on the GPU goes through the fused forward (csrc/ttnet_split.hip: all 7,000 lanes in one
    # launch at f32 accuracy); any other module, or the CPU, through torch
    use_fused = dev.type == "cuda" and fused.supported(actor)
    mu = torch.empty(n, dtype=torch.float32, device=dev)
    for _ in range(max_steps_cap):
        mu_now = fused.actor_forward(actor, obs, mu).view(-1) if use_fused else actor(obs).view(-1)
        action = torch.clamp(mu_now, -1.0, 1.0) * high                     # evaluate=True: no noise (heatmap.py:138)
        obs, _, done, info = env.step(action, auto_reset=False, info=True)
        live = ~finished
        score += torch.where(live, info["comp"][0], torch.zeros_like(score))
        newly = live & done.bool()
        st = env.state
        end_flags = torch.where(newly, info["flags"], end_flags)
        end_xy = torch.where(newly.unsqueeze(1), st[:, 4:6], end_xy)
        traj.append(st[first][:, 4:6].cpu().numpy())
        traj_len += (~finished[first]).long()
        finished |= done.bool()
        if bool(finished.all()):
            break
    env.close()

    score = score.cpu().numpy().reshape(ny, nx, nt)
    flags = end_flags.cpu().numpy()
    success = ((flags & L.F_SUCCESS) != 0).reshape(ny, nx, nt)
    reward_grid = score.mean(axis=2)
    success_grid = success.mean(axis=2)
    orientations_data = [{'x': float(start[k, 0]), 'y': float(start[k, 1]), 'yaw_deg': float(yaw_deg[k]),
                          'yaw_rad': float(start[k, 2])} for k in range(n)]
    end_xy = end_xy.cpu().numpy()
    trajectory_endpoints = [{'end_x': float(end_xy[k, 0]), 'end_y': float(end_xy[k, 1]), 'start_x': float(start[k, 0]),
                             'start_y': float(start[k, 1]),
                             'violation_type': _violation_label(int(flags[k]), bool(flags[k] & L.F_SUCCESS)),
                             'score': float(score.reshape(-1)[k])} for k in range(n)]
    traj = np.stack(traj, 1)                                # [cells, T+1, 2]
    lens = traj_len.cpu().numpy()
    trajectories_data = []
    for c, k in enumerate(range(0, n, nt)):
        trajectories_data.append({'trailer_x': traj[c, :lens[c], 0].tolist(), 'trailer_y': traj[c, :lens[c], 1].tolist(),
                                  'start_x': float(start[k, 0]), 'start_y': float(start[k, 1]),
                                  'start_yaw_deg': float(yaw_deg[k]),
                                  'success': bool(flags[k] & (L.F_SUCCESS | L.F_GOAL_REACHED))})
    return reward_grid, success_grid, x_coords, y_coords, orientations_data, trajectory_endpoints, trajectories_data
