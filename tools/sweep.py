#!/usr/bin/env python3
"""GPU: N-sweep of the step kernel (HBM asymptote) and the K-steps-per-launch rollout (SURVEY §8d ii, iii).
Prints a markdown table; run on the GPU box, copy the output to profiles/."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv

B_ALG = 313
print("| N envs | k_step kernel us (events) | env-steps/s (kernel) | algorithmic GB/s | frac of 8 TB/s | graph-replayed env-steps/s | rollout K=64 env-steps/s |")
print("|---|---|---|---|---|---|---|")
for n in (4096, 16384, 65536, 262144, 1048576, 4194304):
    env = TruckTrailerVecEnv(n)
    env.reset(seed=1)
    for _ in range(50):
        env.step_random(7, auto_reset=True)
    reps = 400 if n <= 262144 else 100
    env.profile(reps)
    for _ in range(reps):
        env.step_random(7, auto_reset=True)
    ms, cnt = env.profile_read(); env.profile(0)
    kern = ms / cnt
    # graph of 20 steps
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(20):
            env.step_random(7, auto_reset=True)
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps // 20 + 1):
        g.replay()
    torch.cuda.synchronize()
    graph_rate = n * 20 * (reps // 20 + 1) / (time.perf_counter() - t0)
    # K-step rollout
    env.rollout_random(64, 7); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        env.rollout_random(64, 7)
    e1.record(); torch.cuda.synchronize()
    roll_rate = n * 64 * 5 / (e0.elapsed_time(e1) * 1e-3)
    gbs = B_ALG * n / (kern * 1e-3) / 1e9
    print(f"| {n} | {kern*1e3:.1f} | {n/(kern*1e-3):.3e} | {gbs:.0f} | {gbs/8000:.3f} | {graph_rate:.3e} | {roll_rate:.3e} |")
    env.close()
