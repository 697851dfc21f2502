"""GPU (TT_STAMPS build, TT_LIB_PATH): phase times of workgroup 0 of the split-bf16 forward."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpg_trucktrailer_amd import _lib as L, fused
from ddpg_trucktrailer_amd.networks import ActorNetwork
dev = torch.device("cuda:0")
a = ActorNetwork(1e-4, (23,), 400, 300, 1, name="actor", device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
obs = torch.rand((n, 23), device=dev); out = torch.empty(n, device=dev)
lib = L.load(); buf = (C.c_ulonglong * 16)()
lib.tt_debug_nstamps.argtypes = [C.c_void_p]
for rep in range(3):
    fused.actor_forward(a, obs, out); torch.cuda.synchronize()
    lib.tt_debug_nstamps(buf); f = list(buf)
    ghz = (f[8 + 4] - f[8 + 3]) / ((f[4] - f[3]) * 10.0)
    print("n=%d  shader clock in layer 2: %.2f GHz -> %.1f cycles per bf16 MFMA (1500 per wave)" % (n, ghz, (f[8 + 4] - f[8 + 3]) / 1500.0))
    print("  block 0, us: stage %.2f  layer1 %.2f  LN1 %.2f  layer2 %.2f  epilogue %.2f | total %.2f" %
          (tuple((f[i + 1] - f[i]) / 100 for i in range(5)) + ((f[5] - f[0]) / 100,)))
