"""GPU: soak run of the N-env loop -- many graph replays back to back, then check that every network is finite, the ring and
env state are finite, the counters agree and the throughput of the last block matches the first.  usage: soak.py [N] [steps] [updates per step]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpg_trucktrailer_amd.rollout import DDPGRollout
from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
updates = int(sys.argv[3]) if len(sys.argv) > 3 else 1
env = TruckTrailerVecEnv(n); env.reset(seed=27)
loop = DDPGRollout(env, batch_size=256, replay_slots=64, seed=27, graph_steps=20, updates_per_step=updates)
loop.prepare()
rates = []
block = max(20, steps // 20 // 20 * 20)
for b in range(steps // block):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loop.run(block)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rates.append(n * block / dt)
    flat = torch.cat([p.detach().reshape(-1) for net in loop.agent._nets() for p in net.parameters()])
    ok = bool(torch.isfinite(flat).all()) and bool(torch.isfinite(loop.ring.obs).all()) and bool(torch.isfinite(env.state).all())
    print(f"block {b}: {rates[-1]:.3e} env-steps/s  finite={ok}  k={loop.ring.k} k_dev={int(loop.ring.k_dev.item())} "
          f"k_pipe={int(loop.k_pipe_dev.item())} learn steps={int(loop.learner.step_dev.item())} (x{updates}) max|w|={flat.abs().max().item():.3f} "
          f"image hand-over: {loop.policy_edge()}, gave up {loop.ring.policy_gave_up()}", flush=True)
    assert loop.ring.policy_gave_up() == 0
    assert ok and loop.ring.k == int(loop.ring.k_dev.item()) == int(loop.k_pipe_dev.item())
assert loop.learner.images_current()
print("soak ok: first/last block rate ratio %.3f" % (rates[-1] / rates[0]))
