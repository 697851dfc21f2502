import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ddpg_trucktrailer_amd import fused
from ddpg_trucktrailer_amd.rollout import DDPGRollout
from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
n, k = 2048, 23
env = TruckTrailerVecEnv(n); env.reset(seed=5)
a = DDPGRollout(env, batch_size=256, replay_slots=8, seed=5, graph_steps=4, pipeline=True)
a.run(k); torch.cuda.synchronize()
obs = a.ring.obs[a.ring.slot()].clone()
mu_before = fused.actor_forward(a.agent.actor, obs).view(-1).clone()
mu_torch = a.agent.actor(obs).view(-1).detach()
print("fused vs torch before:", (mu_before - mu_torch).abs().max().item(), mu_before[:5], mu_torch[:5])
a.run(1); torch.cuda.synchronize()
stored = a.ring.act[a.ring.slot(a.ring.k - 1)]
mu_used = stored - a.noise.x
print("mu_used", mu_used[:5], "vs before", (mu_used - mu_before).abs().max().item(), "vs torch-before", (mu_used - mu_torch).abs().max().item())
mu_after = a.agent.actor(obs).view(-1).detach()
print("vs torch-after", (mu_used - mu_after).abs().max().item())
