import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpg_trucktrailer_amd import _lib as L
from ddpg_trucktrailer_amd.agent import Agent
from ddpg_trucktrailer_amd.fused_learn import FusedLearner
dev = torch.device("cuda:0"); B = 256
ag = Agent(1e-4, 1e-3, (23,), 1e-3, 1, batch_size=B, device=dev, replay=False)
fl = FusedLearner(ag, B)
s = torch.rand((B, 23), device=dev); a = torch.rand((B, 1), device=dev)
lib = L.load()
buf = (C.c_ulonglong * 32)()
for rep in range(3):
    
    fl._fwd(ag.critic, s, a, fl.q, fl.critic.saved); torch.cuda.synchronize()
    lib.tt_debug_stamps(buf); f = list(buf)
    fl._bwd(fl.critic, 1, 2.0 / B, s, a, fl.q, y=fl.y); torch.cuda.synchronize()
    lib.tt_debug_stamps(buf); b = list(buf)
    print("fwd_small us: layer1 %.2f  LN1 %.2f  layer2 %.2f  epilogue %.2f | total %.2f" % (tuple((f[i+1]-f[i])/100 for i in range(4)) + ((f[4]-f[0])/100,)))
    print("  shader clock during fwd layer 2: %.2f GHz; during bwd phase B: %.2f GHz" % ((f[16+3]-f[16+2])/((f[3]-f[2])*10.0), (b[16+10]-b[16+9])/((b[10]-b[9])*10.0)))
    blk = (C.c_ulonglong * 1024)(); lib.tt_debug_blocks(blk)
    st = [blk[2*i] for i in range(205)]; en = [blk[2*i+1] for i in range(205)]
    t0 = min(st)
    late = sorted(range(205), key=lambda i: -en[i])[:5]
    print("bwd_weights grid: starts span %.2f us, ends span up to %.2f us; latest blocks:" % ((max(st)-t0)/100, (max(en)-t0)/100), [(i, round((st[i]-t0)/100,1), round((en[i]-t0)/100,1)) for i in late])
    print("bwd_weights us: dW2 block %.2f  dW1 block %.2f  colsum block %.2f ; starts rel. to dW2 start: dW1 %.2f colsum %.2f" % ((b[13]-b[12])/100, (b[15]-b[14])/100, (b[6]-b[5])/100, (b[14]-b[12])/100, (b[5]-b[12])/100))
    print("bwd_rows  us: phaseA %.2f  phaseB %.2f  phaseC %.2f | total %.2f" % ((b[9]-b[8])/100, (b[10]-b[9])/100, (b[11]-b[10])/100, (b[11]-b[8])/100))
