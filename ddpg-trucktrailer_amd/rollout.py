"""The N-env DDPG loop: the vector form of the reference's training loop (DDPG/trainv2.py:511-531).

Per vector step, for all N envs of this rank at once:
    mu = actor(obs);  a = mu + OU noise            (DDPG_agent.choose_action, trainv2.py:512)
    scaled = clip(a, -1, 1) * f32(pi/4)            (trainv2.py:516)
    obs', r, done = env.step(scaled)               (HIP kernel; finished envs restart in-kernel)
    remember(obs, a, r, obs', done)                (a = the UNCLIPPED noisy action, trainv2.py:525):
                                                   obs', r, done are written by the kernel straight into the ring
    learn()                                        (one gradient step, batch from the ring)
Everything stays on the device.  run(k) replays hipGraphs of `graph_steps` WHOLE vector steps (policy, env step and
learn(): about ten launches per step, back to back with no host in between); step() is the same vector step launched
eagerly, with learn() alone captured (sampling included).  Both give the same bits."""
import math
import os

import numpy as np
import torch

from ddpg_trucktrailer_amd import fused
from ddpg_trucktrailer_amd.agent import Agent
from ddpg_trucktrailer_amd.noise import VecOUNoise
from ddpg_trucktrailer_amd.replay_buffer import TrajectoryRing


# every launch of a capture comes from the capturing thread; "thread_local" keeps another thread's runtime calls (the
# collective library's watchdog, a data loader) from invalidating it
_CAPTURE_MODE = "thread_local"


class DDPGRollout:
    def __init__(self, env, batch_size=256, replay_slots=64, seed=27, alpha=1e-4, beta=1e-3, tau=1e-3, gamma=0.99,
                 fc1_dims=400, fc2_dims=300, world_size=1, use_graph=True, agent=None, fused_learn=True, graph_steps=4):
        self.env, self.n, self.device = env, env.n_envs, env.device
        self.batch_size = batch_size
        torch.manual_seed(seed)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        self.agent = agent if agent is not None else Agent(
            alpha=alpha, beta=beta, input_dims=(env.observation_dim,), tau=tau, n_actions=1, gamma=gamma,
            fc1_dims=fc1_dims, fc2_dims=fc2_dims, batch_size=batch_size, device=self.device,
            capturable=use_graph, replay=False)
        if world_size > 1:
            self.agent.enable_data_parallel()
        self.ring = TrajectoryRing(self.n, replay_slots, env.observation_dim, self.device)
        if self.device.type == "cuda":
            self.ring.attach(env)                          # the step kernel advances the ring's device counter
        self.noise = VecOUNoise(self.n, self.device)
        self.high = float(np.float32(math.pi / 4))       # env.action_space.high (f32 pi/4, simv2.py:86-91)
        self.scaled = torch.zeros(self.n, dtype=torch.float32, device=self.device)
        # the first observation of every env goes into slot 0
        env.observe(out=self.ring.obs[0])
        self.seed = seed
        self.fused_act = fused.supported(self.agent.actor)      # csrc/ttnet.hip: reference-shaped 23-400-300-1 actor
        self.agent.fused_targets = self.fused_act and fused.supported(self.agent.target_critic)
        # hand-fused learn() (csrc/ttlearn.hip) when the networks have the reference's shapes; else torch autograd
        self.learner = None
        if fused_learn and self.fused_act and fused.supported(self.agent.critic):
            from ddpg_trucktrailer_amd.fused_learn import FusedLearner
            self.learner = FusedLearner(self.agent, batch_size)
            self.agent.fused_learner = self.learner
            if world_size > 1:
                self.learner.enable_data_parallel()
        self.use_graph = use_graph and self.device.type == "cuda"
        self.dp = world_size > 1
        if os.environ.get("TT_FORCE_DP") == "1" and self.learner is not None and not self.dp:
            # measurement aid: the data-parallel launch structure (three graph segments, separate Adam launches) on ONE
            # rank with no-op collectives -- what a rank's step costs before any time on the wire
            self.dp = True
            self.learner.grad_sync_critic = self.learner.grad_sync_actor = lambda: None
        self.graph = None
        self.vector_steps = 0
        # whole-step graphs: the ring slots a step touches depend on k mod slots only, so a graph of G steps captured at
        # ring position c*G is valid whenever k = c*G (mod slots): slots/G graphs cover the cycle
        self.graph_steps = int(graph_steps) if (self.use_graph and self.learner is not None and graph_steps
                                                and replay_slots % int(graph_steps) == 0) else 0
        if self.dp and self.graph_steps:
            self.graph_steps = 1        # data-parallel: a step is three graphs with the two gradient all-reduces between
        self.step_graphs = None

    # -------------------------------------------------------------- acting
    @torch.no_grad()
    def act(self, obs, act_out, done_prev=None):
        if self.fused_act:     # actor forward + OU noise + clip*high in ONE launch (tt_actor_act)
            if self.ring._env_counts:      # noise keyed by the DEVICE step counter: the launch is graph-replayable
                return fused.actor_act(self.agent.actor, obs, self.noise.x, act_out, self.scaled, seed=self.seed,
                                       step=0, step_dev=self.ring.k_dev, done_prev=done_prev, high=self.high)
            return fused.actor_act(self.agent.actor, obs, self.noise.x, act_out, self.scaled, seed=self.seed,
                                   step=self.vector_steps, done_prev=done_prev, high=self.high)
        if done_prev is not None:
            self.noise.reset(done_prev)
        mu = self.agent.actor(obs).view(-1)
        torch.add(mu, self.noise.sample(), out=act_out)                   # stored action: unclipped mu + noise
        torch.clamp(act_out, -1.0, 1.0, out=self.scaled).mul_(self.high)  # what the env is driven with
        return self.scaled

    # -------------------------------------------------------------- learning
    def _learn_once(self):
        if self.device.type == "cuda":
            s, a, r, s2, d = self.ring.sample_fused(self.batch_size, seed=self.seed, done_as_bool=self.learner is None)
        else:
            s, a, r, s2, d = self.ring.sample(self.batch_size)
        if self.learner is not None:
            self.learner.learn_batch(s, a, r, s2, d)                      # raw uint8 done flags of the sample
        else:
            self.agent.learn_batch(s, a, r, s2, d)

    def learn(self):
        if self.ring.k < 2:
            return
        if not self.use_graph or self.dp:      # (collectives are not captured)
            return self._learn_once()
        if self.graph is None:
            # warm up (allocator, Adam state, autograd's AccumulateGrad nodes) on the SAME side stream the
            # capture then uses: a backward captured on another stream than the one those nodes were created
            # on needs cross-stream syncs that break the capture
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    self._learn_once()
            side.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode=_CAPTURE_MODE):
                self._learn_once()
            torch.cuda.current_stream().wait_stream(side)
        self.graph.replay()

    # -------------------------------------------------------------- one vector step
    def _act_and_step(self, k):
        """The policy + env launches of vector step number k (k selects the ring slots; everything else is on the device)."""
        ring = self.ring
        t, t1 = ring.slot(k), ring.slot(k + 1)
        # the noise of an env whose episode ended at the previous step restarts at 0 (trainv2.py:492)
        done_prev = ring.done[ring.slot(k - 1)] if k > 0 else None
        scaled = self.act(ring.obs[t], ring.act[t], done_prev)
        self.env.step(scaled, auto_reset=True, obs_out=ring.obs[t1], reward_out=ring.rew[t], done_out=ring.done[t])

    def step(self):
        self._act_and_step(self.ring.k)
        self.ring.advance()
        self.learn()
        self.vector_steps += 1

    # -------------------------------------------------------------- many vector steps
    def _capture_step_graphs(self):
        ring, G = self.ring, self.graph_steps
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        self.step_graphs = []
        for c in range(ring.slots // G):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode=_CAPTURE_MODE):
                for i in range(G):
                    self._act_and_step(ring.slots + c * G + i)      # + slots: any k > 0 with this ring position
                    if self.dp:                                     # up to the critic's gradient; see _dp_step
                        s, a, r, s2, d = ring.sample_fused(self.batch_size, seed=self.seed, done_as_bool=False)
                        self.learner.phase_a(s, a, r, s2, d, fuse_adam=False)
                    else:
                        self._learn_once()
            self.step_graphs.append(g)
        if self.dp:
            s = ring._bufs[0]
            self.dp_graphs = []
            for fn in (lambda: self.learner.phase_b(s, separate_adam=True), self.learner.phase_c):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side, capture_error_mode=_CAPTURE_MODE):
                    fn()
                self.dp_graphs.append(g)
        torch.cuda.current_stream().wait_stream(side)

    def _dp_step(self):
        """One data-parallel vector step: three graph replays with the reference's two optimizer-site all-reduces
        (DDPG_agent.py:95-104) between them, launched eagerly on the flat gradient buffers."""
        self.step_graphs[self.ring.k % self.ring.slots].replay()
        self.learner.grad_sync_critic()
        self.dp_graphs[0].replay()
        self.learner.grad_sync_actor()
        self.dp_graphs[1].replay()

    def prepare(self):
        """Everything one-off that run() would otherwise do lazily inside the first calls (a few eager vector steps that
        warm up allocators / kernel attributes / Adam state, then the graph captures), so that a timed region holds
        steady-state steps only.  Advances the loop by 4 vector steps."""
        while self.ring.k < 4 or (self.graph_steps and self.ring.k % self.graph_steps):
            self.step()
        if self.graph_steps and self.step_graphs is None and self.ring._env_counts:
            try:
                self._capture_step_graphs()
            except Exception as exc:
                import warnings
                warnings.warn(f"whole-step hipGraph capture failed ({exc!r}); continuing with eager steps")
                self.step_graphs, self.graph_steps = None, 0

    def run(self, k):
        """k vector steps.  Whole-step hipGraphs whenever the ring position is a multiple of graph_steps and at least
        graph_steps steps remain (after a few eager steps that warm everything up); eager step() otherwise."""
        ring, G = self.ring, self.graph_steps
        while k > 0:
            if G and k >= G and ring.k >= 4 and ring.k % G == 0 and ring._env_counts:
                if self.step_graphs is None:
                    try:
                        self._capture_step_graphs()
                    except Exception as exc:        # capture refused (driver / library state): the eager path is the same bits
                        import warnings
                        warnings.warn(f"whole-step hipGraph capture failed ({exc!r}); continuing with eager steps")
                        self.step_graphs, self.graph_steps = None, 0
                        G = 0
                        continue
                if self.dp:
                    self._dp_step()
                else:
                    self.step_graphs[(ring.k % ring.slots) // G].replay()
                ring.k += G                     # host mirror; the step kernels advanced k_dev
                self.vector_steps += G
                k -= G
            else:
                self.step()
                k -= 1
