"""GPU (TT_STAMPS build at TT_LIB_PATH): phases of the FIRST tile of the policy forward's workgroups, and per-tile times.

    TT_LIB_PATH=$PWD/tools/dbg/libttenv_stamps.so python3 tools/policy_phases.py [n_envs]

Phases (csrc/ttnet_split.hip NSTAMP 0..5): entry -> operands landed -> layer 1 -> LayerNorm 1 (statistics) -> layer 2 -> epilogue."""
import ctypes as C
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpg_trucktrailer_amd import _lib as L
from ddpg_trucktrailer_amd import fused
from ddpg_trucktrailer_amd.networks import ActorNetwork

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
a = ActorNetwork(1e-4, (23,), 400, 300, 1, name="actor", device=dev)
obs = torch.rand((n, 23), device=dev)
out = torch.empty(n, device=dev)
lib = L.load()
if not hasattr(lib, "tt_debug_bstamps"):
    raise SystemExit("needs the TT_STAMPS build: TT_LIB_PATH=tools/dbg/libttenv_stamps.so")
lib.tt_debug_bstamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.tt_debug_policy_tiles.argtypes = [C.POINTER(C.c_ulonglong)]
for _ in range(10):
    fused.actor_forward(a, obs, out)
torch.cuda.synchronize()
names = ["operands landed", "layer 1", "LayerNorm 1 before layer 2", "layer 2", "epilogue"]
for rep in range(3):
    fused.actor_forward(a, obs, out)
    torch.cuda.synchronize()
    nb = 512
    buf = (C.c_ulonglong * (8 * nb))()
    lib.tt_debug_bstamps(buf, nb)
    rows = [[buf[8 * b + i] for i in range(6)] for b in range(nb) if buf[8 * b + 5] > buf[8 * b]]
    t0 = min(r[0] for r in rows)
    med = [statistics.median((r[i + 1] - r[i]) / 100 for r in rows) for i in range(5)]
    tiles = (C.c_ulonglong * 4096)()
    lib.tt_debug_policy_tiles(tiles)
    per_round = []
    for q in range(4):
        d = [(tiles[b * 8 + q * 2 + 1] - tiles[b * 8 + q * 2]) / 100 for b in range(512) if tiles[b * 8 + q * 2 + 1] > tiles[b * 8 + q * 2] > 0]
        if d:
            per_round.append((len(d), statistics.median(d), max(d)))
    print(f"N={n} rep {rep}: {len(rows)} workgroups; first tile (median us): " + ", ".join(f"{k} {v:.2f}" for k, v in zip(names, med))
          + f" | sum {sum(med):.2f}")
    ns = (C.c_ulonglong * 16)()
    lib.tt_debug_nstamps.argtypes = [C.POINTER(C.c_ulonglong)]
    lib.tt_debug_nstamps(ns)
    print("   workgroup 0: shader clock during layer 2 %.2f GHz, during the epilogue %.2f GHz" % (
        (ns[8 + 4] - ns[8 + 3]) / ((ns[4] - ns[3]) * 10.0), (ns[8 + 5] - ns[8 + 4]) / ((ns[5] - ns[4]) * 10.0)))
    print("   tiles per round of the loop (count, median us, max us): " + "; ".join(f"{c} {m:.2f} {x:.2f}" for c, m, x in per_round)
          + f" | launch first start -> last end {(max(r[5] for r in rows) - t0) / 100:.2f} (first tiles only)")
