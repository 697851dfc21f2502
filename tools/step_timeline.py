"""GPU (TT_STAMPS build of the library, TT_LIB_PATH=tools/dbg/libttenv_stamps.so): a timeline of the loop THAT SHIPS.

rocprofv3 cannot show it: under the tool the loop falls back to graph edges (DDPGRollout.policy_edge), and every intercepted
dispatch moves.  Here every workgroup of every launch of a vector step leaves wall-clock stamps (100 MHz, one clock for the chip):
the policy launch (begin / end of every tile round of every workgroup), the env step (begin / end of every workgroup) and
learn()'s five launches (begin / end of every workgroup) -- of the SAME 20-step hipGraph replay of the config-3 loop at
N = 65536, device-memory hand-overs as in bench.py's default run.  The stamps of a launch are overwritten by its next launch, so
what is read after a replay are the LAST vector step's launches.  Also counted over the replay: how the two hand-over waits
ended in every workgroup that made them (first poll succeeded / had to wait).

    TT_LIB_PATH=$PWD/tools/dbg/libttenv_stamps.so python3 tools/step_timeline.py [n_envs] > profiles/r04_step_timeline_stamps.txt"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import learn_blocks
import torch

from ddpg_trucktrailer_amd import _lib as L
from ddpg_trucktrailer_amd.rollout import DDPGRollout
from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv

LEARN = ["k_fwd_multi", "k_bwd_rows_pair", "k_bwd_weights<critic>", "k_fwd_small<critic>", "k_bwd_weights<actor>"]
LGRID = [64, 32 + 1 + 5, 133 + 25 + 52, 16, 133 + 25 + 42]


def read(lib):
    for name in ("tt_debug_kblocks", "tt_debug_policy_tiles", "tt_debug_step_blocks", "tt_debug_policy_poll", "tt_debug_learn_poll"):
        if not hasattr(lib, name):
            raise SystemExit(f"{name} missing: run with TT_LIB_PATH=tools/dbg/libttenv_stamps.so (python ddpg-trucktrailer_amd/build.py --variant stamps)")
    kb = (C.c_ulonglong * (6 * 1024))(); lib.tt_debug_kblocks(kb)
    tl = (C.c_ulonglong * 4096)(); lib.tt_debug_policy_tiles(tl)
    sb = (C.c_ulonglong * 2048)(); lib.tt_debug_step_blocks(sb)
    return (np.array(kb, dtype=np.int64).reshape(6, 512, 2), np.array(tl, dtype=np.int64).reshape(512, 4, 2),
            np.array(sb, dtype=np.int64).reshape(1024, 2))


LOG_CAP = 16384


def read_log(fn, *pre):
    """-> [[begin, end], ...] of every workgroup logged since the last reset (ticks of the 100 MHz clock), then reset."""
    buf = (C.c_ulonglong * (1 + 2 * LOG_CAP))()
    assert fn(*pre, buf, 1) == 0
    k = min(int(buf[0]), LOG_CAP)
    return np.array(buf[1:1 + 2 * k], dtype=np.int64).reshape(k, 2)


def launches(e, gap_us=20.0):
    """Cluster workgroup stamps of ONE kernel into its launches: sorted by begin, a new launch starts where a begin lies more than
    gap_us after the one before it (the workgroups of a learn() / env launch begin within a few us of each other and launches of one
    kernel are a whole vector step apart; the policy's tile rounds begin ~20 us apart, its launches > 40 us after the last round)."""
    if len(e) == 0:
        return []
    e = e[np.argsort(e[:, 0])]
    cuts = np.nonzero(np.diff(e[:, 0]) / 100.0 > gap_us)[0] + 1
    return np.split(e, cuts)


def whole_replay(lib, loop):
    """Every launch of one 20-step replay, from the logs: one line per vector step."""
    for fn, pre in ((lib.tt_debug_log_policy, ()), (lib.tt_debug_log_step, ())) + tuple((lib.tt_debug_log_learn, (k,)) for k in range(5)):
        fn(*pre, None, 1)
    torch.cuda.synchronize()
    loop.run(20)
    torch.cuda.synchronize()
    pol = launches(read_log(lib.tt_debug_log_policy), gap_us=32.0)
    stp = launches(read_log(lib.tt_debug_log_step))
    lrn = [launches(read_log(lib.tt_debug_log_learn, k)) for k in range(5)]
    print(f"\nall {len(pol)} policy launches, {len(stp)} env steps and {[len(x) for x in lrn]} learn() launches of ONE 20-step replay (us; time 0 = the first "
          "policy workgroup of the replay).  P = policy launch first begin -> last end, E = env step, then learn()'s five launches "
          "first begin -> last end; 'idle' = from the previous env step's last end to this policy launch's first begin:")
    t0 = pol[0][:, 0].min()
    us = lambda x: (x - t0) / 100.0
    prev_env_end = None
    for i in range(len(pol)):
        P, E = pol[i], stp[i] if i < len(stp) else None
        idle = "" if prev_env_end is None else f" idle {us(P[:, 0].min()) - prev_env_end:5.2f}"
        line = f"  step {i:2d}: P {us(P[:, 0].min()):8.2f} -> {us(P[:, 1].max()):8.2f} ({(P[:, 1].max() - P[:, 0].min()) / 100.0:5.2f})"
        if E is not None:
            line += f"  E {us(E[:, 0].min()):8.2f} -> {us(E[:, 1].max()):8.2f} ({(E[:, 1].max() - E[:, 0].min()) / 100.0:5.2f})"
            prev_env_end = us(E[:, 1].max())
        line += idle
        print(line)
    n_l = min(len(x) for x in lrn)
    for i in range(n_l):
        parts = []
        for k in range(5):
            Lk = lrn[k][i]
            parts.append(f"{us(Lk[:, 0].min()):8.2f}->{us(Lk[:, 1].max()):8.2f}")
        first, last = lrn[0][i][:, 0].min(), lrn[4][i][:, 1].max()
        med0 = np.median((lrn[0][i][:, 1] - lrn[0][i][:, 0]) / 100.0)
        print(f"  learn {i:2d}: " + " | ".join(parts) + f"   chain {(last - first) / 100.0:6.2f}  (k_fwd_multi median workgroup {med0:5.2f})")
    if len(pol) > 2:
        per = np.diff([us(p[:, 0].min()) for p in pol])
        print(f"  policy launch to policy launch: median {np.median(per):.2f} us (min {per.min():.2f}, max {per.max():.2f})")
    if n_l > 2:
        per = np.diff([us(l[:, 0].min()) for l in lrn[0][:n_l]])
        print(f"  learn() start to learn() start: median {np.median(per):.2f} us (min {per.min():.2f}, max {per.max():.2f})")


def polls(lib):
    a, b = (C.c_ulonglong * 4)(), (C.c_ulonglong * 4)()
    lib.tt_debug_policy_poll(a); lib.tt_debug_learn_poll(b)
    return np.array([a[0], a[1], b[2], b[3]], dtype=np.int64)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    lib = L.load()
    env = TruckTrailerVecEnv(n); env.reset(seed=27)
    loop = DDPGRollout(env, batch_size=256, replay_slots=64, seed=27, graph_steps=20)
    loop.run(4 + 20 + 4 + 1)
    for _ in range(30):                       # steady clocks
        loop.run(20)
    torch.cuda.synchronize()
    print(f"config 3 at N = {n}: 20-step hipGraph replays of the loop as bench.py runs it; image hand-over: {loop.policy_edge()}; "
          f"policy grid cap {loop.policy_workgroups}")
    lib.tt_debug_bstamps.argtypes = [C.c_void_p, C.c_int]
    for rep in range(3):
        p0 = polls(lib)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); loop.run(20); e1.record()
        torch.cuda.synchronize()
        dp = polls(lib) - p0
        kb, tl, sb = read(lib)
        print(f"\nreplay {rep}: {e0.elapsed_time(e1) * 1000 / 20:.2f} us per vector step (events around the replay)")
        print(f"  hand-overs over the 20 steps: policy image -- first poll succeeded in {dp[0]} workgroups, waited in {dp[1]}; "
              f"learn()'s wait for the step chain -- first poll in {dp[2]} workgroups, waited in {dp[3]}")
        wg = int((tl[:, 0, 0] > 0).sum())
        rounds = [r for r in range(4) if (tl[:wg, r, 1] > 0).any()]
        ns = (n + 255) // 256
        # time zero: the policy launch's first workgroup of the LAST step
        t0 = tl[:wg, 0, 0].min()
        us = lambda x: (x - t0) / 100.0
        print(f"  step chain of the last step (time 0 = first workgroup of its policy launch; {wg} workgroups):")
        print(f"    policy launch      start {us(tl[:wg, 0, 0].min()):7.2f}..{us(tl[:wg, 0, 0].max()):7.2f}   end {us(max(tl[:wg, r, 1].max() for r in rounds)):7.2f}")
        for r in rounds:
            m = tl[:wg, r, 1] > 0
            d = (tl[:wg, r, 1][m] - tl[:wg, r, 0][m]) / 100.0
            print(f"      tile round {r}: {int(m.sum()):3d} workgroups  start {us(tl[:wg, r, 0][m].min()):7.2f}..{us(tl[:wg, r, 0][m].max()):7.2f}  "
                  f"end {us(tl[:wg, r, 1][m].min()):7.2f}..{us(tl[:wg, r, 1][m].max()):7.2f}  median tile {np.median(d):5.2f}  max {d.max():5.2f}")
        # phases of every workgroup's FIRST tile (kernel entry, operands landed, layer 1, LayerNorm 1, layer 2, epilogue)
        bs = np.zeros((512, 8), np.uint64)
        lib.tt_debug_bstamps(bs.ctypes.data_as(C.c_void_p), 512)
        bs = bs[:wg, :6].astype(np.int64)
        ph = np.diff(bs, axis=1) / 100.0
        print(f"      kernel entry (first instruction) {us(bs[:, 0].min()):7.2f}..{us(bs[:, 0].max()):7.2f}; first tile, median per workgroup: entry -> image "
              f"awaited, operands landed {np.median(ph[:, 0]):.2f}, layer 1 {np.median(ph[:, 1]):.2f}, LayerNorm 1 + split {np.median(ph[:, 2]):.2f}, "
              f"layer 2 {np.median(ph[:, 3]):.2f}, epilogue {np.median(ph[:, 4]):.2f}")
        pol_end = max(tl[:wg, r, 1].max() for r in rounds)
        print(f"    k_step ({ns} wgs)   start {us(sb[:ns, 0].min()):7.2f}..{us(sb[:ns, 0].max()):7.2f}   end {us(sb[:ns, 1].min()):7.2f}..{us(sb[:ns, 1].max()):7.2f}   "
              f"median wg {np.median((sb[:ns, 1] - sb[:ns, 0]) / 100.0):5.2f}   gap after the policy launch {(sb[:ns, 0].min() - pol_end) / 100.0:5.2f}")
        print(f"    step chain, first policy workgroup -> last k_step workgroup: {us(sb[:ns, 1].max()):7.2f} us")
        print("  learn chain (the launches whose stamps are the newest: learn() of the last step), same time axis:")
        prev = None
        for k, (name, g) in enumerate(zip(LEARN, LGRID)):
            g = min(g, 512)
            m = kb[k, :g, 0] > 0
            st, en = kb[k, :g, 0][m], kb[k, :g, 1][m]
            gap = "" if prev is None else f"  gap after previous launch {(st.min() - prev) / 100.0:5.2f}"
            print(f"    {name:22s} {int(m.sum()):3d} wgs  start {us(st.min()):7.2f}..{us(st.max()):7.2f}  end {us(en.min()):7.2f}..{us(en.max()):7.2f}  "
                  f"median wg {np.median((en - st) / 100.0):5.2f}{gap}")
            prev = en.max()
        first = kb[0, :64, 0][kb[0, :64, 0] > 0].min()
        print(f"    learn chain, first start -> last end: {(prev - first) / 100.0:7.2f} us; it began {us(first):+.2f} us relative to the policy launch")
        print("  the same launches' workgroup 0, phase by phase, as they ran IN the loop (tools/learn_blocks.py prints them for learn() alone):")
        learn_blocks.phases(lib)
    lib.tt_debug_bstamps.argtypes = [C.c_void_p, C.c_int]
    for name in ("tt_debug_log_policy", "tt_debug_log_step", "tt_debug_log_learn"):
        getattr(lib, name).restype = C.c_int
    lib.tt_debug_log_policy.argtypes = [C.c_void_p, C.c_int]
    lib.tt_debug_log_step.argtypes = [C.c_void_p, C.c_int]
    lib.tt_debug_log_learn.argtypes = [C.c_int, C.c_void_p, C.c_int]
    whole_replay(lib, loop)
    assert loop.ring.policy_gave_up() == 0
    env.close()


if __name__ == "__main__":
    main()
