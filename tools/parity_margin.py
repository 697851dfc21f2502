#!/usr/bin/env python3
"""GPU diagnostic: max |HIP - fixture| per golden trajectory and per quantity (how much of the 1e-5
tolerance each fixture uses), and the same against the C oracle at N=4096."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import all_trajectories, needs_raw_state
from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv

ts = [(p.id, p.values[0]) for p in all_trajectories()]
n = len(ts)
env = TruckTrailerVecEnv(n)
env.set_pose(np.stack([t["start"] for _, t in ts]), goal=np.stack([t["goal"] for _, t in ts]),
             L2=np.array([float(t["L2"]) for _, t in ts]))
raw = [i for i, (_, t) in enumerate(ts) if needs_raw_state(t)]
env.set_state(np.stack([ts[i][1]["state0"] for i in raw]), idx=raw)
env.set_max_steps([int(t["max_episode_steps"]) for _, t in ts])
err = np.zeros((n, 3))
T = max(len(t["actions"]) for _, t in ts)
for k in range(T):
    a = np.array([t["actions"][k] if k < len(t["actions"]) else 0.0 for _, t in ts], np.float32)
    obs, rew, done, info = env.step(torch.from_numpy(a).cuda(), auto_reset=False, info=True)
    st = env.state.cpu().numpy(); ob = obs.cpu().numpy(); tot = info["comp"][0].cpu().numpy()
    for i, (_, t) in enumerate(ts):
        if k < len(t["actions"]):
            err[i] = np.maximum(err[i], [np.abs(st[i] - t["states"][k]).max(), np.abs(ob[i] - t["obs"][k]).max(),
                                         abs(tot[i] - t["reward"][k])])
print(f"{'fixture':40s} {'state':>9s} {'obs':>9s} {'reward':>9s}")
for (name, _), e in zip(ts, err):
    print(f"{name:40s} {e[0]:9.2e} {e[1]:9.2e} {e[2]:9.2e}")
print(f"{'MAX':40s} {err[:,0].max():9.2e} {err[:,1].max():9.2e} {err[:,2].max():9.2e}  (tolerance 1e-5)")
