"""GPU (TT_STAMPS build of the library, TT_LIB_PATH): where the time of ONE learn() goes, from wall-clock stamps (100 MHz) at
the begin and end of EVERY workgroup of its five launches -- alone on the chip (eager launches back to back) and inside the
N-env loop (eager pipelined steps: the policy's capped grids run beside it).  Prints, per launch and relative to the first
workgroup of the first launch: first / last workgroup start, first / last workgroup end, median workgroup duration, and the
gap from the previous launch's last end to this launch's first start (the kernel boundary as the workgroups see it)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ddpg_trucktrailer_amd import _lib as L
from ddpg_trucktrailer_amd.agent import Agent
from ddpg_trucktrailer_amd.fused_learn import FusedLearner

NAMES = ["k_fwd_multi", "k_bwd_rows_pair", "k_bwd_weights<critic>", "k_fwd_small<critic>", "k_bwd_weights<actor>"]
GRID = [64, 32, 133 + 25 + 52, 16, 133 + 25 + 42]


def read(lib):
    buf = (C.c_ulonglong * (6 * 1024))()
    assert lib.tt_debug_kblocks(buf) == 0
    return np.array(buf, dtype=np.int64).reshape(6, 512, 2)


def phases(lib):
    """Phase stamps of workgroup 0 (a dW2 block) of the two weight-gradient launches and of workgroup 0 of the row kernels."""
    buf = (C.c_ulonglong * 32)()
    if not hasattr(lib, "tt_debug_wst") or lib.tt_debug_wst(buf) != 0:
        return
    w = np.array(buf, dtype=np.int64).reshape(2, 16)
    names = ["loads issued -> factors", "factors -> MFMAs done", "pow + partial stores", "barrier", "sums + Adam", "image patch", "end barrier"]
    for k, net in enumerate(("critic", "actor")):
        d = np.diff(w[k, :8]) / 100.0
        print(f"  k_bwd_weights<{net}> workgroup 0 phases (us): entry -> operand loads issued {(w[k, 9] - w[k, 8]) / 100.0:.2f}, -> Adam state "
              f"loads issued {(w[k, 0] - w[k, 9]) / 100.0:.2f},", ", ".join(f"{n} {x:.2f}" for n, x in zip(names, d)))
    st = (C.c_ulonglong * 32)()
    lib.tt_debug_stamps(st)
    f = list(st)
    print("  row kernels, workgroup 0 (us): fwd layer1 %.2f LN1 %.2f layer2 %.2f epilogue %.2f | bwd phaseA %.2f phaseB %.2f phaseC %.2f" % (
        tuple((f[i + 1] - f[i]) / 100 for i in range(4)) + tuple((f[i + 1] - f[i]) / 100 for i in (8, 9, 10))))
    if hasattr(lib, "tt_debug_substamps"):
        sb = (C.c_ulonglong * 16)()
        lib.tt_debug_substamps(sb)
        g = list(sb)
        us = lambda a, b: (b - a) / 100
        print("  row forward, workgroup 0, inside the phases (us): entry -> layer-1 loads issued %.2f, fc1 into LDS (= loads landed) %.2f, barrier %.2f, "
              "products %.2f | accumulators into LDS %.2f, barrier %.2f, rows read (+ barrier) %.2f, statistics + normalise + split + saves issued %.2f, "
              "barrier %.2f" % (us(f[0], g[0]), us(g[0], g[1]), us(g[1], g[2]), us(g[2], f[1]), us(f[1], g[3]), us(g[3], g[4]), us(g[4], g[5]),
                                us(g[5], g[6]), us(g[6], f[2])))


def report(a, title, lib=None):
    print(title)
    t0 = a[0, :GRID[0], 0].min()
    prev_end = None
    for k, (name, g) in enumerate(zip(NAMES, GRID)):
        st, en = (a[k, :g, 0] - t0) / 100.0, (a[k, :g, 1] - t0) / 100.0
        gap = "" if prev_end is None else f"  gap after previous launch {st.min() - prev_end:5.2f}"
        print(f"  {name:24s} {g:3d} wgs  start {st.min():6.2f}..{st.max():6.2f}  end {en.min():6.2f}..{en.max():6.2f}  "
              f"median wg {np.median(en - st):5.2f}  max wg {np.max(en - st):5.2f}{gap}")
        prev_end = en.max()
        if k in (2, 4):      # the weight-gradient launches hold three kinds of workgroups: which kind ends the launch?
            kinds = (("dW2 tiles", 0, 133), ("dW1 tiles", 133, 158), ("column sums", 158, g))
            print("      by kind (end relative to the launch's first start; median / max workgroup duration): " + "; ".join(
                f"{nm} [{a}, {b}): end {en[a:b].min() - st.min():.2f}..{en[a:b].max() - st.min():.2f}, {np.median((en - st)[a:b]):.2f} / {(en - st)[a:b].max():.2f}"
                for nm, a, b in kinds if b > a))
    print(f"  chain: first start -> last end {prev_end:6.2f} us")
    if a[5, 0, 0] and 0 < t0 - a[5, 0, 0] < 100000:
        print(f"  the previous learn()'s last workgroup ended {(t0 - a[5, 0, 0]) / 100.0:5.2f} us before this one's first began")


def main():
    dev = torch.device("cuda:0")
    lib = L.load()
    B = 256
    ag = Agent(1e-4, 1e-3, (23,), 1e-3, 1, batch_size=B, device=dev, replay=False)
    fl = FusedLearner(ag, B)
    s = torch.rand((B, 23), device=dev); a = torch.rand((B, 1), device=dev); r = torch.rand(B, device=dev)
    d8 = torch.zeros(B, dtype=torch.uint8, device=dev)
    for _ in range(20):
        fl.learn_batch(s, a, r, s, d8)
    torch.cuda.synchronize()
    for rep in range(3):
        fl.learn_batch(s, a, r, s, d8)
        torch.cuda.synchronize()
        report(read(lib), f"learn() alone on the chip, eager launches (repeat {rep})")
    # captured: the same five launches as one hipGraph
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=side):
        fl.learn_batch(s, a, r, s, d8)
    torch.cuda.current_stream().wait_stream(side)
    for rep in range(3):
        g.replay(); g.replay(); g.replay()
        torch.cuda.synchronize()
        report(read(lib), f"learn() alone, third of three back-to-back hipGraph replays (repeat {rep})")
        phases(lib)
    if os.environ.get("TT_LB_SHORT") == "1":
        return
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    env = TruckTrailerVecEnv(n); env.reset(seed=27)
    loop = DDPGRollout(env, batch_size=256, replay_slots=64, seed=27, graph_steps=20)
    loop.run(4 + 20 + 4 + 1)
    torch.cuda.synchronize()
    for rep in range(3):
        loop.run(20)
        torch.cuda.synchronize()
        report(read(lib), f"learn() of the LAST step of a 20-step graph of the N = {n} loop (policy grids beside it) (repeat {rep})")
        phases(lib)
    # several updates per vector step: the LAST update of the last step runs alone on the chip (the policy and env launches of
    # the step are long over), after 63 others back to back
    del loop
    U = int(os.environ.get("TT_UPDATES", "64"))
    env = TruckTrailerVecEnv(n); env.reset(seed=27)
    loop = DDPGRollout(env, batch_size=256, replay_slots=64, seed=27, graph_steps=4, updates_per_step=U)
    loop.run(4 + 4 + 4 + 1)
    torch.cuda.synchronize()
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); loop.run(8); e1.record()
        torch.cuda.synchronize()
        print(f"{U} updates per step, N = {n}: {e0.elapsed_time(e1) * 1000 / 8 / U:.2f} us per update")
        report(read(lib), f"the LAST of {U} updates of the last step of a 4-step graph (repeat {rep})")
        phases(lib)


if __name__ == "__main__":
    main()
