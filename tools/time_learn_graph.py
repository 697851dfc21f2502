"""GPU: learn() (sample + the 5 fused launches) as a captured hipGraph, time per replay; and the whole vector step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpg_trucktrailer_amd.rollout import DDPGRollout
from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = TruckTrailerVecEnv(n); env.reset(seed=27)
loop = DDPGRollout(env, batch_size=256, replay_slots=64, seed=27, graph_steps=0)
for _ in range(8): loop.step()
torch.cuda.synchronize()
assert loop.graph is not None
def t(fn, reps):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print(f"learn() graph replay: {t(loop.graph.replay, 200):.1f} us")
loop2 = DDPGRollout(TruckTrailerVecEnv(n), batch_size=256, replay_slots=64, seed=27, graph_steps=4)
loop2.env.reset(seed=27)
loop2.prepare()
print(f"whole vector step (graphs of 4): {t(lambda: loop2.run(4), 50) / 4:.1f} us")
