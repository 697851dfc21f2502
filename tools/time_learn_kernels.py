import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from ddpg_trucktrailer_amd import _lib as L, fused
from ddpg_trucktrailer_amd.agent import Agent
from ddpg_trucktrailer_amd.fused_learn import FusedLearner, _p
dev = torch.device("cuda:0")
B = 256
ag = Agent(1e-4, 1e-3, (23,), 1e-3, 1, batch_size=B, device=dev, replay=False)
fl = FusedLearner(ag, B)
s = torch.rand((B, 23), device=dev); a = torch.rand((B, 1), device=dev); r = torch.rand(B, device=dev)
d8 = torch.zeros(B, dtype=torch.uint8, device=dev)
def timeit(fn, name, reps=200):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1)/reps*1e3:8.1f} us")
timeit(lambda: fl._fwd(ag.actor, s, None, fl.mu), "fwd actor (no save)")
timeit(lambda: fl._fwd(ag.critic, s, a, fl.q, fl.critic.saved), "fwd critic (save)")
timeit(lambda: fl._bwd(fl.critic, 1, 2.0 / B, s, a, fl.q, y=fl.y), "bwd critic (rows+weights)")
timeit(lambda: fl._adam(fl.critic, fl.hyp_critic, 1e-3), "adam+soft critic")
timeit(lambda: fl.learn_batch(s, a, r, s, d8), "learn_batch (eager)", 100)
big = torch.rand((65536, 23), device=dev); out = torch.empty(65536, device=dev)
timeit(lambda: fused.actor_forward(ag.actor, big, out), "actor fwd N=65536", 50)
