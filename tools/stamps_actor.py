"""GPU (TT_STAMPS build, TT_LIB_PATH): phase times of the split-f16 forward -- workgroup 0 in shader clocks, and every
workgroup's wall-clock phase boundaries (start order, duration by dispatch round, XCC)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ddpg_trucktrailer_amd import _lib as L, fused
from ddpg_trucktrailer_amd.networks import ActorNetwork
dev = torch.device("cuda:0")
a = ActorNetwork(1e-4, (23,), 400, 300, 1, name="actor", device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
obs = torch.rand((n, 23), device=dev); out = torch.empty(n, device=dev)
lib = L.load(); buf = (C.c_ulonglong * 16)()
lib.tt_debug_nstamps.argtypes = [C.c_void_p]
nb = min(4096, (n + 127) // 128)
bbuf = np.zeros((nb, 8), np.uint64)
lib.tt_debug_bstamps.argtypes = [C.c_void_p, C.c_int]
for rep in range(3):
    fused.actor_forward(a, obs, out); torch.cuda.synchronize()
    lib.tt_debug_nstamps(buf); f = list(buf)
    ghz = (f[8 + 4] - f[8 + 3]) / ((f[4] - f[3]) * 10.0)
    print("n=%d  shader clock in layer 2: %.2f GHz -> %.1f cycles per f16 MFMA (750 per wave)" % (n, ghz, (f[8 + 4] - f[8 + 3]) / 750.0))
    print("  block 0, us: stage %.2f  layer1 %.2f  LN1 %.2f  layer2 %.2f  epilogue %.2f | total %.2f" %
          (tuple((f[i + 1] - f[i]) / 100 for i in range(5)) + ((f[5] - f[0]) / 100,)))
lib.tt_debug_bstamps(bbuf.ctypes.data, nb)
t = bbuf[:, :6].astype(np.int64)
t0 = t[:, 0].min()
start = (t[:, 0] - t0) / 100.0
dur = (t[:, 5] - t[:, 0]) / 100.0
ph = np.diff(t, axis=1) / 100.0
first = start < np.median(start)
print(f"kernel span {(t[:,5].max() - t0) / 100.0:.2f} us over {nb} workgroups")
for name, m in (("first half of starts", first), ("second half", ~first)):
    print(f"  {name}: start {start[m].min():.2f}..{start[m].max():.2f} us, duration mean {dur[m].mean():.2f} (min {dur[m].min():.2f} max {dur[m].max():.2f}); "
          "phases mean: stage %.2f layer1 %.2f LN1 %.2f layer2 %.2f epilogue %.2f" % tuple(ph[m].mean(0)))
xcc = bbuf[:, 7].astype(np.int64) & 7
print("  workgroups per XCC:", np.bincount(xcc, minlength=8).tolist(), " mean duration per XCC:", [round(float(dur[xcc == x].mean()), 1) for x in range(8)])
