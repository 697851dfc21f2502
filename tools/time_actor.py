import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpg_trucktrailer_amd import fused
from ddpg_trucktrailer_amd.networks import ActorNetwork
dev = torch.device("cuda:0")
a = ActorNetwork(1e-4, (23,), 400, 300, 1, name="actor", device=dev)
for n in (65536, 262144):
    obs = torch.rand((n, 23), device=dev); out = torch.empty(n, device=dev)
    for fn, name in ((lambda: fused.actor_forward(a, obs, out), "fused"), (lambda: a(obs), "torch")):
        with torch.no_grad():
            for _ in range(5): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): fn()
            e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print(f"N={n} {name}: {ms*1e3:.1f} us  ({n*2*(23*400+400*300+300)/ms/1e9:.1f} TFLOP/s)")
