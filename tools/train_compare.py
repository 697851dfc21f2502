#!/usr/bin/env python3
"""GPU: training behaviour at the REFERENCE's data/update ratio (one env, batch 64, one learn() per env step), reported the
way profiles/r01_training_behaviour.md reports the reference's own run (blocks of 20 episodes: mean length, mean return):
  facade  -- the trainv2-shaped loop (DDPG/trainv2.py:488-531) on this build's reference-API Agent + gym facade of the HIP env
  vector  -- the N = 1 vector loop (DDPGRollout): serial or pipelined order, fused or torch learner
Usage: train_compare.py facade|vector-serial|vector-pipelined|vector-torch [env_steps] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

mode = sys.argv[1]
total = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 27
t0 = time.time()
lengths, returns, goals = [], [], []


def report(final=False):
    n = len(lengths)
    if n and (n % 20 == 0 or final):
        a = max(0, n - 20) if not final else (n // 20) * 20
        if a < n:
            print(f"episodes {a + 1:4d}-{n:4d}: mean length {np.mean(lengths[a:]):6.1f}  mean return {np.mean(returns[a:]):9.1f}  "
                  f"goals {int(np.sum(goals[a:])):2d}  steps {int(np.sum(lengths))}  {time.time() - t0:.0f}s", flush=True)


if mode == "facade":
    from ddpg_trucktrailer_amd.agent import Agent
    from ddpg_trucktrailer_amd.env import Truck_trailer_Env_2
    from ddpg_trucktrailer_amd.seed_utils import set_seed
    set_seed(seed)
    env = Truck_trailer_Env_2()
    agent = Agent(alpha=0.0001, beta=0.001, input_dims=env.observation_space.shape, tau=0.001, batch_size=64, fc1_dims=400,
                  fc2_dims=300, n_actions=env.action_space.shape[0])
    steps, i = 0, 0
    while steps < total:
        obs, _ = env.reset(seed=seed + i)
        agent.noise.reset()
        done, score, n = False, 0.0, 0
        while not done:
            a = agent.choose_action(obs)
            obs_, r, done, info = env.step(np.clip(a, -1, 1) * env.action_space.high)
            agent.remember(obs, a, r, obs_, done)
            agent.learn()
            score += r; obs = obs_; n += 1
        steps += n; i += 1
        lengths.append(n); returns.append(score); goals.append(bool(info.get("success", False)))
        report()
else:
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    env = TruckTrailerVecEnv(1)
    env.reset(seed=seed)
    loop = DDPGRollout(env, batch_size=64, replay_slots=65536, seed=seed, graph_steps=0,
                       pipeline=(mode == "vector-pipelined"), fused_learn=(mode != "vector-torch"))
    print(f"{mode}: pipeline={loop.pipeline} fused learner={loop.learner is not None}", flush=True)
    score, n = 0.0, 0
    for s in range(total):
        t = loop.ring.slot()
        loop.step()
        r, d = float(loop.ring.rew[t, 0]), bool(loop.ring.done[t, 0])
        score += r; n += 1
        if d:
            lengths.append(n); returns.append(score); goals.append(r > 150)
            score, n = 0.0, 0
            report()
report(final=True)
