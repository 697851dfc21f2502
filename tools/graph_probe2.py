"""GPU probe: time per step of the pipelined whole-step graph, by variant of the cross-branch dependency."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpg_trucktrailer_amd import fused
from ddpg_trucktrailer_amd.rollout import DDPGRollout
from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
n = 65536
env = TruckTrailerVecEnv(n); env.reset(seed=27)
loop = DDPGRollout(env, batch_size=256, replay_slots=64, seed=27, graph_steps=0)
for _ in range(8): loop.step()
torch.cuda.synchronize()
main, side = torch.cuda.Stream(), loop._pipe_side

def timeit(fn, reps=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def variant(kind, k):
    cur = torch.cuda.current_stream()
    if kind == "pack_first":              # pack, THEN fork: the pack is on the critical path but the graph is a plain fork/join
        fused.pack(loop.agent.actor, 0)
        side.wait_stream(cur)
        with torch.cuda.stream(side): loop._learn_all()
        loop._act_and_step(k)
        cur.wait_stream(side)
    elif kind == "event":                 # fork, pack on the policy branch, event into the middle of learn()
        side.wait_stream(cur)
        fused.pack(loop.agent.actor, 0)
        ev = torch.cuda.Event(); ev.record(cur)
        with torch.cuda.stream(side): loop._learn_all(before_actor=lambda: side.wait_event(ev))
        loop._act_and_step(k)
        cur.wait_stream(side)
    elif kind == "no_edge":               # as "event" without the edge (NOT race-free; timing only)
        side.wait_stream(cur)
        fused.pack(loop.agent.actor, 0)
        with torch.cuda.stream(side): loop._learn_all()
        loop._act_and_step(k)
        cur.wait_stream(side)

main.wait_stream(torch.cuda.current_stream())
for kind in ("pack_first", "event", "no_edge"):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main, capture_error_mode="thread_local"):
        for i in range(4): variant(kind, 64 + 8 + i)
    torch.cuda.current_stream().wait_stream(main)
    print(f"{kind:10s}: {timeit(g.replay) / 4:.1f} us/step")
