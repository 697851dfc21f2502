"""GPU: time of the N-env actor forward (split-bf16 kernel incl. its pack launch; exact-f32 kernel) at a few N."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpg_trucktrailer_amd import fused
from ddpg_trucktrailer_amd.networks import ActorNetwork
dev = torch.device("cuda:0")
a = ActorNetwork(1e-4, (23,), 400, 300, 1, name="actor", device=dev)
for n in (65536, 262144):
    obs = torch.rand((n, 23), device=dev); out = torch.empty(n, device=dev)
    for name in ("split", "f32"):
        ctx = fused.exact_f32(a) if name == "f32" else None
        if ctx: ctx.__enter__()
        for _ in range(5): fused.actor_forward(a, obs, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fused.actor_forward(a, obs, out)
        e1.record(); torch.cuda.synchronize()
        if ctx: ctx.__exit__(None, None, None)
        ms = e0.elapsed_time(e1) / 50
        print(f"{os.path.basename(os.environ.get('TT_LIB_PATH', 'libttenv.so'))} N={n} {name}: {ms*1e3:.1f} us")
