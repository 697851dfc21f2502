"""GPU: the persistent policy kernel alone, by workgroup cap (image packed once; no pack in the timed loop)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpg_trucktrailer_amd import _lib as L, fused
from ddpg_trucktrailer_amd.networks import ActorNetwork
dev = torch.device("cuda:0")
a = ActorNetwork(1e-4, (23,), 400, 300, 1, name="actor", device=dev)
n = 65536
obs = torch.rand((n, 23), device=dev); out = torch.empty(n, device=dev)
fused.pack(a, 0)
lib = L.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for cap in (0, 256, 224, 192, 171, 128, 64):
    w = fused.packed_weights_of(a, 0, cap)
    f = lambda: L.check(lib.tt_actor_forward(n, C.c_void_p(obs.data_ptr()), C.byref(w), C.c_void_p(out.data_ptr()), st))
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print(f"cap {cap:4d}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
