"""GPU: time per vector step against the number of learn() updates per step (pipelined order, graphs): the slope is the cost of
one more update, the intercept what a step costs besides its updates.  usage: updates_scaling.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddpg_trucktrailer_amd.rollout import DDPGRollout
from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    pts = []
    for U in (1, 2, 4, 8, 16, 32, 64):
        env = TruckTrailerVecEnv(n); env.reset(seed=27)
        G = 20 if U <= 4 else 4
        loop = DDPGRollout(env, batch_size=256, replay_slots=64, seed=27, graph_steps=G, updates_per_step=U)
        loop.run(4 + G + 4 + 1)
        torch.cuda.synchronize()
        k = max(G, (200 // U) // G * G)
        loop.run(k)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); loop.run(k); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / k * 1e6
            best = min(best, dt)
        pts.append((U, best))
        print(f"N = {n}, {U:2d} updates per step ({loop.policy_edge()} hand-over): {best:8.1f} us per step = {best / U:6.2f} per update", flush=True)
        del loop, env
    for (u0, t0), (u1, t1) in zip(pts, pts[1:]):
        print(f"  {u0:2d} -> {u1:2d} updates: {(t1 - t0) / (u1 - u0):6.2f} us per additional update")


if __name__ == "__main__":
    main()
