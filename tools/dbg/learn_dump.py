"""GPU: three learn() calls on fixture F5's batch with the library at TT_LIB_PATH; everything they leave -> an .npz (A/B of two builds)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ddpg_trucktrailer_amd.fused_learn import FusedLearner
from test_gpu_fused_learn import _agent
from test_learner import _batch
z = np.load(os.path.join(ROOT, "tests", "golden", "f5_learner.npz"), allow_pickle=False)
dev = torch.device("cuda:0")
ag = _agent(dev, z)
s, a, r, s2, d = _batch(z, dev)
fl = FusedLearner(ag, 256, fc2_images=(sys.argv[2] == "1"))
out = {}
for step in range(3):
    fl.learn_batch(s, a, r, s2, d.to(torch.uint8))
    torch.cuda.synchronize()
    for name, st in (("critic", fl.critic), ("actor", fl.actor)):
        out[f"{step}/{name}/grad"] = st.flat_grad.cpu().numpy().copy()
        out[f"{step}/{name}/m"] = st.m.cpu().numpy().copy()
        out[f"{step}/{name}/w"] = torch.cat([p.detach().reshape(-1) for p in st.params]).cpu().numpy()
        out[f"{step}/{name}/t"] = torch.cat([p.detach().reshape(-1) for p in st.targets]).cpu().numpy()
    for k in ("q", "y", "mu", "dq_da", "q_pi", "mu_t", "q_t"):
        out[f"{step}/{k}"] = getattr(fl, k).cpu().numpy().copy()
    for k, v in fl.ws_t.items():
        out[f"{step}/ws/{k}"] = v.cpu().numpy().copy()
    for k, v in fl.critic.saved_t.items():
        out[f"{step}/saved_c/{k}"] = v.cpu().numpy().copy()
np.savez(sys.argv[1], **out)
