"""GPU: what a 20-step timed region costs beyond 20 steady vector steps (the driver's `--steps 20 --warmup 5` form): the region
after idles of several lengths, two and five replays back to back, and the region's first kernel start seen from the host."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddpg_trucktrailer_amd.rollout import DDPGRollout
from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    env = TruckTrailerVecEnv(n); env.reset(seed=27)
    loop = DDPGRollout(env, batch_size=256, replay_slots=64, seed=27, graph_steps=G)
    loop.run(4 + G + 4 + 1)
    torch.cuda.synchronize()
    loop.run(100 * G)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); loop.run(100 * G); torch.cuda.synchronize(); t1 = time.perf_counter()
    steady = (t1 - t0) / (100 * G) * 1e3
    print(f"N = {n}, G = {G}: steady {steady:.4f} ms per step (100 replays back to back)")
    for idle in (0.0, 0.0, 0.001, 0.01, 0.1, 0.0):
        for reps in (1, 2, 5):
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                if idle:
                    time.sleep(idle)
                t0 = time.perf_counter(); loop.run(reps * G); torch.cuda.synchronize(); t1 = time.perf_counter()
                ts.append((t1 - t0) * 1e3)
            ts.sort()
            med = ts[2]
            print(f"  idle {idle * 1e3:6.1f} ms, {reps} replay(s): region {med:.4f} ms = {med / (reps * G):.4f} per step; beyond steady "
                  f"{(med - reps * G * steady) * 1e3:7.1f} us   (min {ts[0]:.4f}, max {ts[-1]:.4f})")
    # how long the GPU takes to come back to its steady clocks after an idle: replay after replay, events between them
    for idle in (0.1, 0.01):
        torch.cuda.synchronize()
        time.sleep(idle)
        R = 60
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(R + 1)]
        ev[0].record()
        for i in range(R):
            loop.run(G)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = [ev[i].elapsed_time(ev[i + 1]) / G for i in range(R)]
        print(f"  after {idle * 1e3:.0f} ms idle, ms per step of replay 1..{R}: " + " ".join(f"{x:.4f}" for x in ms))
    # host side of one replay: how long the launch call itself takes
    torch.cuda.synchronize()
    t0 = time.perf_counter(); loop.run(G); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"  host: run({G}) returned after {(t1 - t0) * 1e6:.0f} us, region {(t2 - t0) * 1e6:.0f} us")


if __name__ == "__main__":
    main()
