"""GPU probe: does a hipGraph with two parallel branches (policy + env step beside learn()) overlap them, and what does
the fork/join cost?  Prints per-step times of: serial graph, forked graph (learn on the origin stream), two-stream eager."""
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
main = torch.cuda.Stream(); side = torch.cuda.Stream()

def timeit(fn, reps=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def body_serial(k):
    loop._act_and_step(k); loop._learn_all()

def body_fork(k, s_main, s_side):
    s_side.wait_stream(s_main)
    with torch.cuda.stream(s_side):
        loop._act_and_step(k)
    loop._learn_all()
    s_main.wait_stream(s_side)

G = 4
gs = torch.cuda.CUDAGraph()
main.wait_stream(torch.cuda.current_stream())
with torch.cuda.graph(gs, stream=main, capture_error_mode="thread_local"):
    for i in range(G): body_serial(64 + 8 + i)
gf = torch.cuda.CUDAGraph()
with torch.cuda.graph(gf, stream=main, capture_error_mode="thread_local"):
    for i in range(G): body_fork(64 + 8 + i, main, side)
ga = torch.cuda.CUDAGraph()
with torch.cuda.graph(ga, stream=main, capture_error_mode="thread_local"):
    for i in range(G): loop._act_and_step(64 + 8 + i)
gl = torch.cuda.CUDAGraph()
with torch.cuda.graph(gl, stream=main, capture_error_mode="thread_local"):
    for i in range(G): loop._learn_all()
torch.cuda.current_stream().wait_stream(main)
print(f"N={n}: serial graph {timeit(gs.replay) / G:.1f} us/step | forked graph {timeit(gf.replay) / G:.1f} | "
      f"policy+step alone {timeit(ga.replay) / G:.1f} | learn alone {timeit(gl.replay) / G:.1f}")

def eager_two_streams():
    cur = torch.cuda.current_stream()
    body_fork(72, cur, side)
print(f"two-stream eager {timeit(eager_two_streams, 50):.1f} us/step")

# ---- variants
def capture(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main, capture_error_mode="thread_local"):
        fn()
    return g

def body_fork_rev(k, s_main, s_side):          # learn on the forked stream, policy on the origin
    s_side.wait_stream(s_main)
    with torch.cuda.stream(s_side):
        loop._learn_all()
    loop._act_and_step(k)
    s_main.wait_stream(s_side)

main.wait_stream(torch.cuda.current_stream())
for G2 in (1, 8):
    g = capture(lambda: [body_fork(64 + 8 + i, main, side) for i in range(G2)])
    print(f"forked graph, {G2} step(s) per graph: {timeit(g.replay) / G2:.1f} us/step")
g = capture(lambda: [body_fork_rev(64 + 8 + i, main, side) for i in range(4)])
print(f"forked graph, learn on the side stream: {timeit(g.replay) / 4:.1f} us/step")
hi = torch.cuda.Stream(priority=-1)
g = capture(lambda: [body_fork_rev(64 + 8 + i, main, hi) for i in range(4)])
print(f"forked graph, learn on a high-priority side stream: {timeit(g.replay) / 4:.1f} us/step")
# host-driven: one graph per branch per step on two streams, events between them
sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
ga1 = capture(lambda: loop._act_and_step(72))
gl1 = capture(lambda: loop._learn_all())
torch.cuda.current_stream().wait_stream(main)
ea, eb = torch.cuda.Event(), torch.cuda.Event()
def two_graphs():
    with torch.cuda.stream(sa):
        sa.wait_event(eb); ga1.replay(); ea.record(sa)
    with torch.cuda.stream(sb):
        sb.wait_event(ea); gl1.replay(); eb.record(sb)
eb.record(sb); ea.record(sa)
print(f"two graphs per step on two streams (host events): {timeit(two_graphs, 100):.1f} us/step")
