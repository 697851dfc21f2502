#!/usr/bin/env python3
"""GPU: training behaviour of the N-env loop at a chosen data/update ratio: per block of vector steps, the mean return
and length of the episodes that ended in it and how many reached the goal (+200 success bonus: last reward > 150).
Usage: train_vector.py n_envs ring_slots updates_per_step batch vector_steps report_every [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpg_trucktrailer_amd.rollout import DDPGRollout
from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv

n, slots, upd, batch, total, every = (int(x) for x in sys.argv[1:7])
seed = int(sys.argv[7]) if len(sys.argv) > 7 else 27
env = TruckTrailerVecEnv(n)
env.reset(seed=seed)
loop = DDPGRollout(env, batch_size=batch, replay_slots=slots, seed=seed, updates_per_step=upd, graph_steps=1 if slots <= 1024 else 0)
print(f"N = {n}, ring {slots} steps ({slots * n:.2e} transitions), {upd} learn() per vector step = {n / upd:.1f} env-steps per update, "
      f"batch {batch}, pipeline={loop.pipeline}", flush=True)
ret = torch.zeros(n, device=env.device, dtype=torch.float64)
length = torch.zeros(n, device=env.device, dtype=torch.int64)
t0 = time.time()
acc = torch.zeros(4, device=env.device, dtype=torch.float64)        # sum of returns, episodes, sum of lengths, goals
for s in range(1, total + 1):
    t = loop.ring.slot()
    loop.run(1)
    r, d = loop.ring.rew[t].double(), loop.ring.done[t].bool()
    ret += r; length += 1
    acc[0] += ret[d].sum(); acc[1] += d.sum(); acc[2] += length[d].sum(); acc[3] += (d & (r > 150)).sum()
    ret[d] = 0; length[d] = 0
    if s % every == 0:
        a = acc.tolist(); acc.zero_()
        e = max(1.0, a[1])
        print(f"vector steps {s:7d} ({s * n:.2e} env-steps, {int(loop.learner.step_dev.item())} updates): episodes {int(a[1]):7d}  "
              f"mean length {a[2] / e:6.1f}  mean return {a[0] / e:9.1f}  goals {int(a[3]):6d} ({100 * a[3] / e:4.1f} %)  {time.time() - t0:.0f}s", flush=True)
