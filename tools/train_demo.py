#!/usr/bin/env python3
"""GPU: an actual training run of the N-env DDPG loop (the reference's hyper-parameters, trainv2.py:404-407), to see the
policy improve: mean reward per env-step, episodes finished, share of them that reach the goal (final reward > 150: the
+200 success bonus of reward_functionv1.py:466-470 is the only way to get there), per block of vector steps.
Usage: train_demo.py [n_envs] [vector_steps] [report_every] [ring_slots] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpg_trucktrailer_amd.rollout import DDPGRollout
from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
total = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
ring_slots = int(sys.argv[4]) if len(sys.argv) > 4 else 64
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 256
env = TruckTrailerVecEnv(n)
env.reset(seed=27)
loop = DDPGRollout(env, batch_size=batch, replay_slots=ring_slots, seed=27, fused_learn=os.environ.get('TT_TORCH_LEARN') != '1')
loop.prepare()
slots = loop.ring.slots
t0 = time.time()
done_steps = 0
print(f"N = {n}, batch {batch}, alpha 1e-4, beta 1e-3, tau 1e-3, gamma 0.99, OU(0.2, 0.15, 0.01); one learn() per vector step")
print("vector steps | env-steps | mean reward/step | episodes ended | reached goal | jackknife-like (r < -400) | s")
while done_steps < total:
    block_r = torch.zeros((), device=env.device, dtype=torch.float64)
    ended = torch.zeros((), device=env.device, dtype=torch.int64)
    good = torch.zeros((), device=env.device, dtype=torch.int64)
    bad = torch.zeros((), device=env.device, dtype=torch.int64)
    steps_in_block = 0
    while steps_in_block < every:
        chunk = min(slots - 4, 60)
        loop.run(chunk)                                # fewer than the ring holds, so its newest `chunk` slots are these steps
        k = loop.ring.k
        idx = torch.tensor([(k - 1 - i) % slots for i in range(chunk)], device=env.device)
        r, d = loop.ring.rew[idx], loop.ring.done[idx].bool()
        block_r += r.double().sum()
        ended += d.sum()
        good += (d & (r > 150)).sum()
        bad += (d & (r < -400)).sum()
        steps_in_block += chunk
    done_steps += steps_in_block
    torch.cuda.synchronize()
    e = max(1, int(ended))
    print(f"{done_steps:12d} | {done_steps * n:.3e} | {float(block_r) / (steps_in_block * n):+9.3f} | {int(ended):10d} | "
          f"{int(good):9d} ({100.0 * int(good) / e:5.1f} %) | {int(bad):9d} ({100.0 * int(bad) / e:5.1f} %) | {time.time() - t0:6.1f}", flush=True)
