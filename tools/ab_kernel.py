#!/usr/bin/env python3
"""GPU: time the step kernel of the library named by TT_LIB_PATH at a few N (per-dispatch events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
out = []
for n in ([int(x) for x in sys.argv[1:]] or [65536, 1048576]):
    env = TruckTrailerVecEnv(n); env.reset(seed=1)
    for _ in range(100): env.step_random(7, auto_reset=True)
    best = 1e9
    for rep in range(3):
        env.profile(300)
        for _ in range(300): env.step_random(7, auto_reset=True)
        ms, cnt = env.profile_read(); best = min(best, ms / cnt)
    env.profile(0); env.close()
    out.append(f"N={n}: {best*1e3:.2f} us")
print(os.path.basename(os.environ.get("TT_LIB_PATH", "libttenv.so")), " | ".join(out))
