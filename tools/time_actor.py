"""GPU: time of the N-env policy forward (pack + kernel) for the library at TT_LIB_PATH (variant builds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpg_trucktrailer_amd import fused
from ddpg_trucktrailer_amd.networks import ActorNetwork
dev = torch.device("cuda:0")
a = ActorNetwork(1e-4, (23,), 400, 300, 1, name="actor", device=dev)
for n in [int(x) for x in (sys.argv[1:] or ["65536"])]:
    obs = torch.rand((n, 23), device=dev); out = torch.empty(n, device=dev)
    for _ in range(10): fused.actor_forward(a, obs, out)
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fused.actor_forward(a, obs, out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50)
    print(f"{os.environ.get('TT_LIB_PATH','default')} N={n}: {best*1e3:.1f} us")
