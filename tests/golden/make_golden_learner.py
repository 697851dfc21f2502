#!/usr/bin/env python3
"""F5: learner vectors from the REAL reference agent (DDPG/DDPG_agent.py, networks.py, noise.py,
replay_buffer.py), run on the CPU of the build container.  Writes tests/golden/f5_learner.npz.

The reference picks `cuda:1` when no GPU is visible (networks.py:51,134) and then fails in `.to`; here
`nn.Module.to` is wrapped to ignore CUDA devices that do not exist, and `.device` of the four nets is
set to cpu afterwards.  `memory.sample_buffer` is replaced by a function returning ONE fixed batch so
that learn() (the reference's own code, unmodified) is a deterministic function of weights + batch."""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("TT_REFERENCE", "/root/reference")


def import_agent():
    orig_to = torch.nn.Module.to

    def to_cpu_if_no_gpu(self, *args, **kw):
        if args and isinstance(args[0], torch.device) and args[0].type == "cuda" and not torch.cuda.is_available():
            return orig_to(self, torch.device("cpu"))
        return orig_to(self, *args, **kw)

    torch.nn.Module.to = to_cpu_if_no_gpu
    sys.path.insert(0, os.path.join(REF, "DDPG"))
    import DDPG_agent
    return DDPG_agent


SAMPLE_STRIDE = 16


def sd(net, prefix, out, full=False):
    """state_dict -> arrays.  Snapshots after learn() keep every 16th element of fc2.weight (120k floats) so
    that the fixture stays small; everything else is stored whole."""
    for k, v in net.state_dict().items():
        a = v.detach().cpu().numpy().copy()
        if not full and k == "fc2.weight":
            a = a.reshape(-1)[::SAMPLE_STRIDE].copy()
        out[f"{prefix}/{k}"] = a


def main():
    mod = import_agent()
    torch.manual_seed(1234)
    np.random.seed(1234)
    B = 256
    agent = mod.Agent(alpha=1e-4, beta=1e-3, input_dims=(23,), tau=1e-3, n_actions=1, batch_size=B,
                      fc1_dims=400, fc2_dims=300)
    for net in (agent.actor, agent.critic, agent.target_actor, agent.target_critic):
        net.device = torch.device("cpu")
    out = {}
    for name in ("actor", "critic"):   # the targets start as exact copies (update_network_parameters(tau=1))
        sd(getattr(agent, name), f"init/{name}", out, full=True)
    for a, b in ((agent.actor, agent.target_actor), (agent.critic, agent.target_critic)):
        assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))

    # a realistic batch: consecutive observations / rewards / dones of the F2 and F3 fixtures
    rows = []
    for f in ("f2_seeded.npz", "f3_branches.npz"):
        z = np.load(os.path.join(HERE, f), allow_pickle=False)
        for n in z["names"]:
            obs = np.concatenate([z[f"{n}/obs0"][None], z[f"{n}/obs"]], 0)
            for k in range(len(z[f"{n}/actions"])):
                rows.append((obs[k], z[f"{n}/actions"][k] / np.float32(np.pi / 4), z[f"{n}/reward"][k], obs[k + 1],
                             z[f"{n}/done"][k]))
    rng = np.random.RandomState(7)
    pick = rng.choice(len(rows), B, replace=False)
    states = np.stack([rows[i][0] for i in pick]).astype(np.float64)
    actions = (np.array([rows[i][1] for i in pick]) + rng.normal(0, 0.1, B)).reshape(B, 1).astype(np.float64)
    rewards = np.array([rows[i][2] for i in pick], dtype=np.float64)
    states_ = np.stack([rows[i][3] for i in pick]).astype(np.float64)
    dones = np.array([rows[i][4] for i in pick], dtype=np.bool_)
    assert dones.any() and not dones.all()
    out.update(batch_states=states, batch_actions=actions, batch_rewards=rewards, batch_states_=states_, batch_dones=dones)

    # forward passes of the reference modules on the batch (before any update)
    with torch.no_grad():
        s = torch.tensor(states, dtype=torch.float)
        a = torch.tensor(actions, dtype=torch.float)
        out["fwd_actor"] = agent.actor.forward(s).numpy()
        out["fwd_critic"] = agent.critic.forward(s, a).numpy()
        ta = agent.target_actor.forward(torch.tensor(states_, dtype=torch.float))
        q_ = agent.target_critic.forward(torch.tensor(states_, dtype=torch.float), ta)
        q_[torch.tensor(dones)] = 0.0
        out["target_y"] = (torch.tensor(rewards, dtype=torch.float) + agent.gamma * q_.view(-1)).numpy()

    # choose_action(evaluate=True) on a few observations
    out["act_obs"] = states[:8].astype(np.float32)
    out["act_mu"] = np.stack([agent.choose_action(o, evaluate=True) for o in states[:8].astype(np.float32)])

    agent.memory.mem_cntr = B
    agent.memory.sample_buffer = lambda batch_size: (states, actions, rewards, states_, dones)

    # Gradients and losses of the reference's OWN learn() (DDPG_agent.py:95-104), observed without touching its code:
    # each optimizer's step() is wrapped to copy every parameter's .grad first (the two sites of the data-parallel
    # all-reduce), F.mse_loss / T.mean as seen from DDPG_agent are wrapped to record the two losses.
    tap = {"n": 0}

    def grab(net, name, orig):
        def step(*a, **kw):
            for k, p_ in net.named_parameters():
                g = p_.grad.detach().cpu().numpy().copy()
                if k == "fc2.weight":
                    g = g.reshape(-1)[::SAMPLE_STRIDE].copy()
                out[f"grad{tap['n']}/{name}/{k}"] = g
            return orig(*a, **kw)
        return step
    agent.critic.optimizer.step = grab(agent.critic, "critic", agent.critic.optimizer.step)
    agent.actor.optimizer.step = grab(agent.actor, "actor", agent.actor.optimizer.step)
    orig_mse, orig_mean = mod.F.mse_loss, mod.T.mean

    def mse(*a, **kw):
        r = orig_mse(*a, **kw)
        out[f"loss{tap['n']}/critic"] = np.float32(r.item())
        return r

    def mean(*a, **kw):
        r = orig_mean(*a, **kw)
        out[f"loss{tap['n']}/actor"] = np.float32(r.item())
        return r

    def learn_tapped():
        tap["n"] += 1
        mod.F.mse_loss, mod.T.mean = mse, mean
        try:
            agent.learn()
        finally:
            mod.F.mse_loss, mod.T.mean = orig_mse, orig_mean

    learn_tapped()
    for name in ("actor", "critic", "target_actor", "target_critic"):
        sd(getattr(agent, name), f"after1/{name}", out)
    learn_tapped()
    learn_tapped()
    for name in ("actor", "critic", "target_actor", "target_critic"):
        sd(getattr(agent, name), f"after3/{name}", out)

    # OU noise recurrence under a seeded numpy stream (noise.py:13-17)
    np.random.seed(5)
    agent.noise.reset()
    out["ou_seed5"] = np.stack([agent.noise() for _ in range(16)])
    # exploration action = mu + noise (DDPG_agent.py:41-45), same stream continued
    out["act_noisy"] = agent.choose_action(states[0].astype(np.float32))
    # replay buffer behaviour (replay_buffer.py:13-34)
    buf = mod.ReplayBuffer(5, (23,), 1)
    for i in range(7):
        buf.store_transition(states[i], actions[i], rewards[i], states_[i], dones[i])
    np.random.seed(11)
    smp = buf.sample_buffer(4)
    out["rb_rewards_after_wrap"] = buf.reward_memory.copy()
    out["rb_sample_seed11_rewards"] = smp[2]
    out["sample_stride"] = np.int64(SAMPLE_STRIDE)
    out["provenance"] = np.array(f"reference DDPG/DDPG_agent.py learn() on CPU, torch {torch.__version__}, numpy {np.__version__}, "
                                 "torch.manual_seed(1234), fixed batch of 256 built from fixtures F2/F3")
    np.savez_compressed(os.path.join(HERE, "f5_learner.npz"), **out)
    npar = lambda net: sum(p.numel() for p in net.parameters())
    print("wrote f5_learner.npz; params actor", npar(agent.actor), "critic", npar(agent.critic))


if __name__ == "__main__":
    main()
