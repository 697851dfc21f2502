"""Checkpoint / resume and best-model policy.

The reference saves the four networks' state_dicts (networks.py:70-95; Agent.save_models keeps doing exactly
that, same file names), a training-state pickle (trainv2.py:210-229) and the last transitions
(trainv2.py:333-351) -- but not the optimizer state, the env or the replay contents, so a resumed run restarts
Adam cold.  `save_training_checkpoint` writes ONE torch file with all of it; `BestModelTracker` is the
reference's "save when best by success rate, then by average score" rule (trainv2.py:538-572)."""
import numpy as np
import torch


def _rng_state():
    kind, keys, pos, has_gauss, gauss = np.random.get_state()
    rng = {"torch": torch.get_rng_state(),
           "numpy": {"kind": str(kind), "keys": torch.from_numpy(keys.astype(np.int64)), "pos": int(pos),
                     "has_gauss": int(has_gauss), "gauss": float(gauss)}}
    if torch.cuda.is_available():
        rng["cuda"] = torch.cuda.get_rng_state()
    return rng


def _restore_rng(rng):
    torch.set_rng_state(rng["torch"])
    nps = rng.get("numpy")
    if isinstance(nps, dict):       # the full legacy-MT state: keys, position and the cached Gaussian (noise.py / replay draws)
        np.random.set_state((nps["kind"], nps["keys"].numpy().astype(np.uint32), nps["pos"], nps["has_gauss"], nps["gauss"]))
    if "cuda" in rng and torch.cuda.is_available():
        torch.cuda.set_rng_state(rng["cuda"])


def save_training_checkpoint(path, agent, env=None, ring=None, noise=None, training_state=None, with_replay=True):
    if getattr(agent, "fused_learner", None) is not None:
        agent.fused_learner.export_to_optimizers()          # Adam moments live in the fused learner's flat buffers
    ck = {"format": 2,
          "nets": {n: getattr(agent, n).state_dict() for n in ("actor", "critic", "target_actor", "target_critic")},
          "optim": {"actor": agent.actor.optimizer.state_dict(), "critic": agent.critic.optimizer.state_dict()},
          "hyper": dict(alpha=agent.alpha, beta=agent.beta, tau=agent.tau, gamma=agent.gamma, batch_size=agent.batch_size),
          "training_state": training_state or {},
          "rng": _rng_state()}
    if env is not None:
        ck["env"] = env.state_dict()
    if noise is not None:
        ck["ou"] = noise.x.detach().cpu()
    if ring is not None:
        ck["ring"] = ring.state_dict(with_replay=with_replay)
    torch.save(ck, path)
    return path


def load_training_checkpoint(path, agent, env=None, ring=None, noise=None):
    """Restores everything save_training_checkpoint wrote.  Load BEFORE DDPGRollout.prepare()/run() capture their
    graphs, or through DDPGRollout.load_state_dict (which re-captures): a captured step launch bakes the env's reset
    seed and modes by value."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    for n, sd in ck["nets"].items():
        getattr(agent, n).load_state_dict(sd)
    agent.actor.optimizer.load_state_dict(ck["optim"]["actor"])
    agent.critic.optimizer.load_state_dict(ck["optim"]["critic"])
    if getattr(agent, "fused_learner", None) is not None:
        agent.fused_learner.import_from_optimizers()
    if env is not None and "env" in ck:
        env.load_state_dict(ck["env"])
    if noise is not None and "ou" in ck:
        noise.x.copy_(ck["ou"].to(noise.x.device))
    if ring is not None and "ring" in ck:
        ring.load_state_dict(ck["ring"])       # counters always; contents when they were saved
    _restore_rng(ck["rng"])
    return ck.get("training_state", {})


def save_loop_checkpoint(path, loop, training_state=None, force=False):
    """The whole N-env loop (DDPGRollout.state_dict: networks, Adam state, env batch, replay ring + side buffer, OU state,
    counters) in one file; a loop restored from it continues bit for bit (tests/test_gpu_rollout.py).
    Refuses (RuntimeError, nothing written) a loop in which a launch gave up waiting for the other chain of its step: that
    step acted on a stale policy image or drew from rows still being written (DDPGRollout._check_handover); force=True
    saves it all the same, with the steps listed under "handover_gave_up"."""
    sd = loop.state_dict()                 # (synchronises and looks at the give-up word)
    if sd.get("handover_gave_up") and not force:
        steps = ", ".join(str(x - 1) for x in sd["handover_gave_up"])
        raise RuntimeError(f"not saving {path}: a launch of vector step(s) {steps} gave up waiting for the other chain of its step and "
                           "went on with stale inputs, so this loop's state is not what the reference's order of operations "
                           "(trainv2.py:511-531) produces; pass force=True to save it anyway")
    torch.save({"format": 2, "loop": sd, "training_state": training_state or {}}, path)
    return path


def load_loop_checkpoint(path, loop):
    ck = torch.load(path, map_location="cpu", weights_only=True)
    loop.load_state_dict(ck["loop"])
    return ck.get("training_state", {})


class BestModelTracker:
    """trainv2.py:538-572: success_rate over the last 100 episodes; best = higher success rate, or equal success
    rate and higher 100-episode average score, and only after 100 episodes of this run."""

    def __init__(self, start_episode=0, best_score=-float("inf"), best_success_rate=0.0):
        self.start_episode = start_episode
        self.best_score, self.best_success_rate = best_score, best_success_rate
        self.score_history, self.success_history, self.step_history = [], [], []
        self.total_steps = 0

    def update(self, episode_index, score, success, steps):
        """Returns (is_best, avg_score, success_rate); the caller saves models when is_best."""
        self.total_steps += int(steps)
        self.score_history.append(float(score))
        self.step_history.append(self.total_steps)
        self.success_history.append(1 if success else 0)
        avg_score = float(np.mean(self.score_history[-100:]))
        success_rate = float(np.mean(self.success_history[-100:]))
        better = success_rate > self.best_success_rate
        equal_better_score = success_rate == self.best_success_rate and avg_score > self.best_score
        is_best = (better or equal_better_score) and episode_index > (self.start_episode + 100)
        if is_best:
            self.best_success_rate, self.best_score = success_rate, avg_score
        return is_best, avg_score, success_rate

    def training_state(self, episode_num):
        """The dict trainv2.py:210-229 pickles (same keys)."""
        return {"episode_num": episode_num, "score_history": list(self.score_history), "best_score": self.best_score,
                "best_success_rate": self.best_success_rate, "success_history": list(self.success_history),
                "total_steps": self.total_steps, "step_history": list(self.step_history)}
