"""GPU: what the data-parallel learn() costs a rank per update before any time on the wire -- the five launches + the two
optimizer launches, captured as one hipGraph and replayed back to back -- with (a) no exchange at all (k_adam_soft after a no-op),
(b) the peer-to-peer exchange at world size 1 (k_adam_soft_p2p: publish, wait, acquire, gradient read from the exchange block),
and, for reference, (c) the single-rank learn() whose Adam runs inside the weight-gradient launches -- with its last two launches apart
("fused") and as one grid ("fused+tail": tt_mlp_actor_tail)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpg_trucktrailer_amd.agent import Agent
from ddpg_trucktrailer_amd.fused_learn import FusedLearner

dev = torch.device("cuda:0")
B = 256


def make(kind):
    torch.manual_seed(0)
    ag = Agent(1e-4, 1e-3, (23,), 1e-3, 1, batch_size=B, device=dev, replay=False)
    fl = FusedLearner(ag, B)
    if kind == "p2p":
        fl.enable_p2p()
    elif kind == "fused+tail":
        fl.fuse_tail = True
    elif kind == "separate":
        fl.grad_sync_critic = fl.grad_sync_actor = lambda: None
    return fl


def time_graph(fl, reps=300):
    s = torch.rand((B, 23), device=dev); a = torch.rand((B, 1), device=dev); r = torch.rand(B, device=dev)
    d8 = torch.zeros(B, dtype=torch.uint8, device=dev)
    for _ in range(10):
        fl.learn_batch(s, a, r, s, d8)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=side):
        fl.learn_batch(s, a, r, s, d8)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps * 1e3)
    return min(out)


for kind in ("fused", "fused+tail", "separate", "p2p"):
    print(f"{kind:10s} learn() per update, back-to-back graph replays: {time_graph(make(kind)):.2f} us", flush=True)
