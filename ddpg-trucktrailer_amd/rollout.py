"""The N-env DDPG loop: the vector form of the reference's training loop (DDPG/trainv2.py:511-531).

Per vector step, for all N envs of this rank at once:
    mu = actor(obs);  a = mu + OU noise            (DDPG_agent.choose_action, trainv2.py:512)
    scaled = clip(a, -1, 1) * f32(pi/4)            (trainv2.py:516)
    obs', r, done = env.step(scaled)               (HIP kernel; finished envs restart in-kernel)
    remember(obs, a, r, obs', done)                (a = the UNCLIPPED noisy action, trainv2.py:525):
                                                   obs', r, done are written by the kernel straight into the ring
    learn() x updates_per_step                     (gradient steps, each on a fresh batch from the ring)
Everything stays on the device.  run(k) replays hipGraphs of whole vector steps (policy, env step and learn(): seven to
nine launches per step with no host in between): graphs of `graph_steps`, 4 and 1 steps serve every
ring position (the launches find the step's ring slots through a device cursor), so that no step of a run() is launched
eagerly whatever k and the position are.  step() is the same vector step launched eagerly.  All of them give the same bits.

Data-parallel ranks (one process per GPU): the two gradient all-reduces of DDPG_agent.py:95-104 are nodes of the step's
graph when dp_probe found captured RCCL collectives working on this node; else a step is three graph segments with
eager collectives between them.

pipeline=True (the default on a GPU with the fused learner): learn() of vector step t runs BESIDE the policy and env
launches of steps t-1 and t, on a second stream.  What this needs:
  * learn() of step t draws from the steps up to t-2 (_PIPE_LAG / _PIPE_RESERVE below; the reference's, and
    pipeline=False's, window also holds steps t-1 and t: a two-slot difference, stated like the other choices of the
    vector loop);
  * the policy reads a packed IMAGE of the actor (csrc/ttnet_split.hip) made by the opening launch of step t from the
    weights learn() of step t-1 left -- never the live weights learn() of step t is updating -- so the policy of step t
    acts with the weights after learn() of step t-1, exactly as in the serial order; image and ring cursor exist twice
    (even / odd steps), because the opening launch of step t may run beside the policy of step t-1;
  * the policy kernel's grid is capped (policy_workgroups) so that learn()'s launches always find free CUs.
Captured graphs hold the two chains with one edge each way per step (_capture_lagged); eager steps join the two streams at
the end of every step, which is stricter; no launch reads what a concurrent one writes, so graphs, eager launches and a
resumed run still agree bit for bit."""
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
_LEARN_EDGE_EVERY = 4                    # pipelined graphs: steps between graph edges from the step chain into the learn chain
_SEED_STRIDE = 0x9E3779B97F4A7C15     # sampling key of update u of a vector step = seed + u * stride (mod 2^64)
# Pipelined order: learn() of vector step t draws from the steps up to t-2 (lag 1: step t-1 may still be under way beside the
# draw) and keeps off the two observation rows the env steps t-1 and t write meanwhile (reserve 2)
_PIPE_LAG, _PIPE_RESERVE = 1, 2


class DDPGRollout:
    def __init__(self, env, batch_size=256, replay_slots=64, seed=27, alpha=1e-4, beta=1e-3, tau=1e-3, gamma=0.99,
                 fc1_dims=400, fc2_dims=300, world_size=1, use_graph=True, agent=None, fused_learn=True, graph_steps=4,
                 updates_per_step=1, data_parallel=None, pipeline=None, policy_workgroups=192, graph_collectives=None,
                 policy_capped_grids=4, dp_exchange=None):
        """updates_per_step: learn() calls per vector step (the reference does one per ENV step, trainv2.py:520-528; one
        per vector step is 1/N of that -- the knob moves the data/update ratio back towards the reference's).
        data_parallel: None = (world_size > 1); True forces the data-parallel launch structure with the process group's
        real collectives even at world size 1 (tests of the RCCL path on one GPU).
        graph_collectives (data-parallel, backend nccl only): the two gradient all-reduces are captured INSIDE the step's
        hipGraph (RCCL launches are capturable), so a data-parallel step is one graph replay like a single-rank one instead
        of three segments with eager collectives between them.  None = the TT_DP_GRAPH_COLLECTIVES environment variable
        ("1" after dp_probe.graph_collectives_ok() saw a captured all-reduce replay correctly on this node).
        dp_exchange (data-parallel): "collective" (default; env TT_DP_EXCHANGE) = one all-reduce per optimizer site on the
        process group; "p2p" = no collective at all on learn()'s chain: every rank's Adam launch reads the peers' flat gradient
        buffers itself through IPC-opened device memory behind a flag barrier (include/ttenv.h: tt_p2p_*; fused learner only).
        A step is then one hipGraph of plain kernel launches on any backend, and data_parallel=True works without a process
        group at world size 1."""
        self.env, self.n, self.device = env, env.n_envs, env.device
        self.batch_size = batch_size
        self.updates_per_step = int(updates_per_step)
        assert self.updates_per_step >= 1
        torch.manual_seed(seed)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        self.dp = (world_size > 1) if data_parallel is None else bool(data_parallel)
        self.agent = agent if agent is not None else Agent(
            alpha=alpha, beta=beta, input_dims=(env.observation_dim,), tau=tau, n_actions=1, gamma=gamma,
            fc1_dims=fc1_dims, fc2_dims=fc2_dims, batch_size=batch_size, device=self.device,
            capturable=use_graph, replay=False)
        self.dp_exchange = dp_exchange or os.environ.get("TT_DP_EXCHANGE", "collective")
        assert self.dp_exchange in ("collective", "p2p"), self.dp_exchange
        if self.dp and self._have_group():
            self.agent.enable_data_parallel()          # rank 0's weights to everyone (+ the torch path's gradient all-reduce)
        elif self.dp and self.dp_exchange != "p2p":
            raise RuntimeError("data_parallel=True needs an initialised process group (or dp_exchange='p2p' at world size 1)")
        if graph_collectives is None:
            graph_collectives = os.environ.get("TT_DP_GRAPH_COLLECTIVES") == "1"
        self.dp_single_graph = False
        if self.dp and graph_collectives:
            import torch.distributed as dist
            self.dp_single_graph = dist.is_initialized() and dist.get_backend() == "nccl"
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
            if self.dp and self.dp_exchange == "p2p":
                self.learner.enable_p2p()
                self.dp_single_graph = True              # nothing but kernel launches in a step: one graph, any backend
            elif self.dp:
                self.learner.enable_data_parallel()
        if self.learner is not None and not self.dp and self.device.type == "cuda":
            # learn()'s last two launches as one grid where learn() bounds the step (include/ttenv.h: tt_mlp_actor_tail): a policy
            # launch of at most one round of 128 tiles, or several updates per step (at N = 65536 with one update per step the grid's 200
            # waiting workgroups meet a policy launch that owns 171 CUs: 0.084 -> 0.094 ms per step).  TT_ACTOR_TAIL=0/1 overrides.
            tail = os.environ.get("TT_ACTOR_TAIL")
            self.learner.fuse_tail = (tail == "1") if tail in ("0", "1") else (self.n <= 16384 or self.updates_per_step > 1)
        if self.dp and self.dp_exchange == "p2p" and self.learner is None:
            raise RuntimeError("dp_exchange='p2p' needs the fused learner (reference-shaped networks on a GPU)")
        self.use_graph = use_graph and self.device.type == "cuda"
        if os.environ.get("TT_FORCE_DP") == "1" and self.learner is not None and not self.dp:
            # measurement aid: the data-parallel launch structure (three graph segments, separate Adam launches) on ONE
            # rank with no-op collectives -- what a rank's step costs before any time on the wire
            self.dp = True
            self.dp_single_graph = os.environ.get("TT_DP_GRAPH_COLLECTIVES") == "1"
            self.learner.grad_sync_critic = self.learner.grad_sync_actor = lambda: None
        self.graph = None
        self._learn_side, self._learn_warm = None, 0
        # pipelined order (module docstring): needs the fused learner and the fused policy kernel
        can_pipe = self.learner is not None and self.fused_act and self.device.type == "cuda" and replay_slots >= 3 + _PIPE_RESERVE \
            and self.ring._env_counts
        self.pipeline = can_pipe if pipeline is None else (bool(pipeline) and can_pipe)
        self.policy_workgroups = int(os.environ.get("TT_POLICY_WG", policy_workgroups))     # (env: A/B measurements)
        # learn() is over after about four of the policy's capped grids (~100 us): the tiles left then (N > 98304 envs) go out
        # in one grid over all CUs
        self.policy_capped_grids = int(os.environ.get("TT_POLICY_CAPPED_GRIDS", policy_capped_grids))
        self.k_pipe_dev = torch.zeros((), dtype=torch.int64, device=self.device)   # steps completed before the running one
        self._k_snap_dev = torch.zeros((), dtype=torch.int64, device=self.device)  # its value as a learn()'s first launch saw it
        self._pipe_side = None
        if self.pipeline:
            self._pipe_side = torch.cuda.Stream(device=self.device)
        self.vector_steps = 0
        # ring addressing: the policy and env launches find the step's ring slots through a device cursor that the step's
        # opening pack launch writes (include/ttenv.h: tt_ring_view), not through per-slot pointers -- so ONE captured
        # graph serves every ring position: a single-step graph and a graph of `graph_steps` steps are all there is
        self.ring_mode = self.fused_act and self.device.type == "cuda" and self.ring._env_counts
        self._view = self.ring.view() if self.ring_mode else None
        if self.ring_mode and self.pipeline:     # both policy images exist before the first opening launch writes one of them
            fused.packed_weights_of(self.agent.actor, 0, self.policy_workgroups, self.policy_capped_grids, two_images=True)
        ok = self.use_graph and self.learner is not None and graph_steps and self.ring_mode
        self.graph_steps = int(graph_steps) if ok else 0
        if self.graph_steps and self.dp and not self.dp_single_graph:
            self.graph_steps = 1        # data-parallel: a step is three graphs with the two gradient all-reduces between
        self.graphM = None              # four vector steps (only when graph_steps > 4)
        self.graph1 = None              # one vector step (data-parallel: up to the critic's gradient)
        self.graphG = None              # graph_steps vector steps (one rank)
        self.dp_graphs = None
        self._graph_epoch = None        # env.graph_epoch the captures were made under
        # hand-overs through device memory (policy image, step progress) are bounded waits: a launch that gives up goes on with
        # stale inputs.  The loop notices (_check_handover), falls back to graph edges once, and raises the second time
        self.handover_gave_up = []      # steps (+1) at which a launch gave up, in the order they were noticed
        self._edge_forced = False

    @staticmethod
    def _have_group():
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()

    # -------------------------------------------------------------- acting
    @torch.no_grad()
    def act(self, obs, act_out, done_prev=None, k=None):
        if self.fused_act:     # actor forward + OU noise + clip*high in ONE launch (tt_actor_act)
            w = None
            if self.pipeline:  # the image packed at the start of this step, never the live weights learn() is updating
                w = fused.packed_weights_of(self.agent.actor, 0, self.policy_workgroups, self.policy_capped_grids, two_images=True)
            if self.ring._env_counts:      # noise keyed by the DEVICE step counter: the launch is graph-replayable
                return fused.actor_act(self.agent.actor, obs, self.noise.x, act_out, self.scaled, seed=self.seed,
                                       step=0, step_dev=self.ring.k_dev, done_prev=done_prev, high=self.high, weights=w)
            return fused.actor_act(self.agent.actor, obs, self.noise.x, act_out, self.scaled, seed=self.seed,
                                   step=self.vector_steps, done_prev=done_prev, high=self.high, weights=w)
        if done_prev is not None:
            self.noise.reset(done_prev)
        mu = self.agent.actor(obs).view(-1)
        torch.add(mu, self.noise.sample(), out=act_out)                   # stored action: unclipped mu + noise
        torch.clamp(act_out, -1.0, 1.0, out=self.scaled).mul_(self.high)  # what the env is driven with
        return self.scaled

    # -------------------------------------------------------------- learning
    def _sample(self, u):
        key = self._sample_key(u)
        if self.device.type == "cuda":
            if self.pipeline:      # beside the env step of the same vector step: its slot is not part of the window
                return self.ring.sample_fused(self.batch_size, seed=key, done_as_bool=False, k_dev=self.k_pipe_dev,
                                              reserve=_PIPE_RESERVE, lag=_PIPE_LAG)
            return self.ring.sample_fused(self.batch_size, seed=key, done_as_bool=self.learner is None)
        return self.ring.sample(self.batch_size)

    def _sample_key(self, u):
        return (self.seed + u * _SEED_STRIDE) & (2 ** 64 - 1)

    def _image_job(self):
        """What lets learn()'s second launch carry the pack of the step's policy image (fused_learn.phase_a: image)."""
        from ddpg_trucktrailer_amd import _lib as L
        w = fused.packed_weights_of(self.agent.actor, 0, self.policy_workgroups, self.policy_capped_grids, two_images=True)
        cur = L.TTRingCursor(self._k_snap_dev.data_ptr(), self.ring.slots, 0, self.ring.cursor_dev.data_ptr())
        return (w, cur, self._k_snap_dev)

    def _learn_once(self, u=0, presampled=False, with_image=False, wait_for_steps=False):
        sample = None
        if presampled:         # the step's opening launch already drew this batch into the ring's buffers
            B, draws = self.batch_size, self._draws_per_opening()
            s, a, r, s2, d = (x[u * B:(u + 1) * B] for x in self.ring._batch_bufs(B * draws)[:5])
        elif self.learner is not None and self.device.type == "cuda":
            # the fused learner's first launch makes the draw itself (tt_mlp_forward_multi_sampled): the same draw as
            # _sample(u), one launch less per update
            if self.pipeline:
                sample = self.ring.sample_args(self.batch_size, seed=self._sample_key(u), k_dev=self.k_pipe_dev,
                                               reserve=_PIPE_RESERVE, lag=_PIPE_LAG, wait_for_steps=wait_for_steps)
            else:
                sample = self.ring.sample_args(self.batch_size, seed=self._sample_key(u))
            s, a, r, s2, d = self.ring._batch_bufs(self.batch_size)[:5]
        else:
            s, a, r, s2, d = self._sample(u)
        if self.learner is not None:
            # (pipelined order) the last update of a vector step moves the sampling window on
            last = self.pipeline and u == self.updates_per_step - 1
            self.learner.learn_batch(s, a, r, s2, d, window_dev=self.k_pipe_dev if last else None, sample=sample,
                                     image=self._image_job() if with_image else None)   # raw uint8 done flags
        else:
            self.agent.learn_batch(s, a, r, s2, d)

    def _draws_per_opening(self):
        """Batches the opening launch of a pipelined step draws: all of the step's updates' (tt_sample_args.draws) -- each the
        draw its own update would make (same window, seed of update u), but made once per step, so that no update waits for
        the ring's rows on the learn chain (2.9 us per update).  Data-parallel ranks and the
        torch learner keep one draw per opening launch."""
        multi = (self.updates_per_step > 1 and self.pipeline and self.learner is not None and not self.dp
                 and self.device.type == "cuda" and os.environ.get("TT_MULTI_DRAW", "1") == "1")
        return self.updates_per_step if multi else 1

    def _learn_all(self, presampled=False, with_image=False, wait_for_steps=False):
        draws = self._draws_per_opening() if presampled else 1
        for u in range(self.updates_per_step):
            self._learn_once(u, presampled and u < draws, with_image and u == 0, wait_for_steps and u == 0)

    def learn(self):
        if self.ring.k < 2:
            return
        # Only the fused learner's launches are captured.  The torch-autograd learner stays eager: its backward runs on
        # the autograd engine's thread and synchronises with whatever stream each parameter's gradient accumulator was
        # first used on -- a graph node kept alive elsewhere (a clone that carries its grad_fn, made on another stream) pulls
        # that stream into the capture, and HIP's EndCapture then takes the PROCESS down (no exception to fall back from).
        if not self.use_graph or (self.dp and not self.dp_single_graph) or self.learner is None:   # (eager collectives)
            return self._learn_all()
        self.learner.refresh_images()
        self._check_epoch()
        if self.graph is None:
            # The first calls run eagerly -- they ARE the updates of their vector steps, so a captured loop makes exactly
            # the updates an eager one makes -- on the side stream the capture then uses; they create everything a
            # capture must not (allocator blocks, the torch optimizers' state, kernel attributes).  The next call is
            # captured (a capture executes nothing) and replayed.
            if self._learn_side is None:
                self._learn_side = torch.cuda.Stream(device=self.device)
            side = self._learn_side
            side.wait_stream(torch.cuda.current_stream())
            if self._learn_warm < 3:
                self._learn_warm += 1
                with torch.cuda.stream(side):
                    self._learn_all()
                torch.cuda.current_stream().wait_stream(side)
                return
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode=_CAPTURE_MODE):
                self._learn_all()
            torch.cuda.current_stream().wait_stream(side)
        self.graph.replay()

    # -------------------------------------------------------------- one vector step
    def _open_step(self, learn):
        """The launch that opens a vector step: the policy's image from the actor's current weights and the ring cursor of
        the step -- in the pipelined order also the first batch of the step's learn()."""
        if self.pipeline and learn:
            fused.pack_and_sample(self.agent.actor, 0, self.ring.sample_args(
                self.batch_size, seed=self._sample_key(0), k_dev=self.k_pipe_dev, reserve=_PIPE_RESERVE, lag=_PIPE_LAG,
                draws=self._draws_per_opening(), seed_stride=_SEED_STRIDE), cursor=self.ring.cursor(self.k_pipe_dev))
        elif self.pipeline:
            fused.pack(self.agent.actor, 0, cursor=self.ring.cursor(self.k_pipe_dev))
        else:
            fused.pack(self.agent.actor, 0, cursor=self.ring.cursor())

    def policy_edge(self):
        """How the policy launch of a captured step learns that its image is complete: "flag" (device memory, the default) or
        "graph" (a graph edge from the opening launch) -- with several updates per step, for data-parallel ranks, on request
        (TT_POLICY_EDGE=graph) and under rocprofv3: a tool that intercepts every dispatch keeps the learn chain ~45 us per
        launch behind, the policy launch then spends its life waiting (189 us per launch in a kernel trace) and the trace says
        nothing about the loop; with --pmc (kernels serialised) a waiting launch would only leave by its time limit."""
        if os.environ.get("TT_POLICY_EDGE", "flag") == "graph" or self.updates_per_step > 1 or self.dp or self._edge_forced:
            return "graph"
        # a policy workgroup takes a whole CU (155 KB of LDS): a grid that is not capped BELOW the number of CUs fills the chip,
        # and a policy launch that spins there for its image keeps the learn chain -- which makes the image -- off the GPU
        if self.policy_workgroups <= 0 or self.policy_workgroups >= self._cu_count():
            return "graph"
        if os.environ.get("TT_POLICY_EDGE") != "flag" and any(k.startswith("ROCPROF") or k == "ROCP_TOOL_LIBRARIES" for k in os.environ):
            return "graph"
        return "flag"

    def _cu_count(self):
        if not hasattr(self, "_cus"):
            self._cus = int(torch.cuda.get_device_properties(self.device).multi_processor_count) if self.device.type == "cuda" else 0
        return self._cus

    def _check_handover(self, exact=False):
        """Has a launch given up waiting for the other chain of its step (include/ttenv.h: TT_CURSOR_GAVE_UP)?  exact=False reads
        the host-visible mirror of the word -- no GPU call, done after every graph replay --, exact=True synchronises and reads
        the device word (state_dict, prepare, the end of a checkpoint).  The first time: warn, drop the graphs and capture them
        again with graph edges between the chains (what bench.py does in its setup: ~4 us per step slower, never waits).  A
        second time -- with edges no launch ever has to wait -- is an error."""
        ring = self.ring
        if self.learner is not None and self.learner.tail_gave_up():
            raise RuntimeError(f"learn step {self.learner.tail_gave_up()}: a weight-gradient workgroup of learn()'s last launch gave up waiting "
                               "(0.25 s) for dQ/da from the row workgroups of the same launch (include/ttenv.h: tt_mlp_actor_tail) and "
                               "went on: that update is garbage.  TT_ACTOR_TAIL=0 runs the two launches apart")
        if self.learner is not None and self.learner.p2p_gave_up():
            raise RuntimeError(f"learn step {self.learner.p2p_gave_up()}: this rank's optimizer launch gave up waiting for a peer's gradients "
                               "(peer-to-peer exchange, include/ttenv.h: tt_p2p_*) and used whatever the buffers held: the ranks have "
                               "diverged.  Keep the ranks within the exchange's time limit of each other (tt_p2p_set_timeout)")
        if not self.ring_mode or ring.gave_up_host is None:
            return 0
        mark = ring.policy_gave_up() if exact else ring.gave_up_seen()
        if not mark:
            return 0
        import warnings
        self.handover_gave_up.append(int(mark))
        torch.cuda.synchronize(self.device)
        ring.clear_gave_up()
        if self.policy_edge() == "graph":
            raise RuntimeError(f"a launch of vector step {mark - 1} gave up waiting for the other chain of its step although the "
                               "chains are ordered by graph edges: the loop's state is not trustworthy (DESIGN.md section 11)")
        warnings.warn(f"a launch of vector step {mark - 1} gave up waiting (0.25 s) for the other chain of its step -- the policy for its "
                      "image, or learn() for the env step -- and went on with stale inputs; capturing the steps again with graph edges "
                      "between the chains (as TT_POLICY_EDGE=graph does)")
        self._edge_forced = True
        self.invalidate_graphs()
        return int(mark)

    def policy_launch(self):
        """The policy launch of the running step alone (ring mode; after _open_step): bench.py times it."""
        w = fused.packed_weights_of(self.agent.actor, 0, self.policy_workgroups if self.pipeline else 0,
                                    self.policy_capped_grids, two_images=self.pipeline)
        return fused.actor_act_ring(self.agent.actor, self._view, w, self.noise.x, self.scaled, seed=self.seed, step=0,
                                    step_dev=self.ring.k_dev, high=self.high)

    def _act_and_step(self, k=None):
        """The policy + env launches of the running vector step.  Ring mode: everything that selects the slots is on the
        device (after _open_step).  Otherwise (CPU, torch actor) k selects them."""
        if self.ring_mode:
            self.policy_launch()
            self.env.step_ring(self.scaled, self._view, auto_reset=True)
            return
        ring = self.ring
        t, t1 = ring.slot(k), ring.slot(k + 1)
        # the noise of an env whose episode ended at the previous step restarts at 0 (trainv2.py:492)
        done_prev = ring.done[ring.slot(k - 1)] if k > 0 else None
        scaled = self.act(ring.obs[t], ring.act[t], done_prev, k=k)
        self.env.step(scaled, auto_reset=True, obs_out=ring.obs[t1], reward_out=ring.rew[t], done_out=ring.done[t])

    def _pipelined(self, k, learn, dp_capture=False):
        """The running vector step (number k) in the pipelined order with a JOIN at its end -- the eager form of a step, and
        the first of the three graph segments of the data-parallel fallback (captured graphs of whole steps use
        _capture_lagged: the same launches without the per-step join):
            current:  opening launch (image + cursor + batch); then, beside each other,
            side:     learn(), whose last update moves the sampling window on      | current:  policy, env step
        dp_capture: only learn()'s first segment (up to the critic's gradient) goes beside the policy; _dp_step does the rest."""
        cur, side = torch.cuda.current_stream(self.device), self._pipe_side
        self._open_step(learn)     # image + cursor + (one kernel and one gap less) the first batch of this step's learn()
        side.wait_stream(cur)
        # learn()'s branch is recorded first: with the policy's launches first the step takes 0.131 ms instead of 0.120
        with torch.cuda.stream(side):
            if dp_capture:
                self.learner.phase_a(*self.ring._batch_bufs(self.batch_size)[:5], fuse_adam=False,
                                     window_dev=self.k_pipe_dev if self.updates_per_step == 1 else None)
            elif learn:
                self._learn_all(presampled=True)
            else:
                self.k_pipe_dev.add_(1)                     # no learn() yet: the window still moves with the steps
        self._act_and_step()
        cur.wait_stream(side)

    def step(self):
        k = self.ring.k
        if self.pipeline:
            self._check_epoch()
            self._pipelined(k, k >= 2)
            self.ring.advance()
        else:
            if self.ring_mode:
                self._open_step(False)
            self._act_and_step(k)
            self.ring.advance()
            self.learn()
        self.vector_steps += 1

    # -------------------------------------------------------------- many vector steps
    def invalidate_graphs(self):
        """Drop every captured graph (they bake kernel arguments by value: the env's reset seed, per-env-goal mode and
        pose pool, the ring's side-buffer count; ring and network addresses).  Called automatically when the env or the
        ring reports a change of those (env.graph_epoch, ring.side_epoch)."""
        self.graph = self.graph1 = self.graphG = self.graphM = self.dp_graphs = None

    def _check_epoch(self):
        # (the actor's parameter storages are part of it: the policy's packed-image struct is keyed on them, fused.py)
        epoch = (getattr(self.env, "graph_epoch", 0), self.ring.side_epoch,
                 fused.packed_key_of(self.agent.actor) if self.fused_act else None)
        if self._graph_epoch != epoch:
            if self._graph_epoch is not None:
                self.invalidate_graphs()
            self._graph_epoch = epoch

    def _graphs_current(self):
        self._check_epoch()
        return self.graph1 is not None

    def _capture_body(self):
        """One whole vector step (data-parallel: up to the critic's gradient; the rest in _dp_step)."""
        segments = self.dp and not self.dp_single_graph
        if self.pipeline:
            self._pipelined(None, True, dp_capture=segments)
            return
        self._open_step(False)
        self._act_and_step()
        if segments:
            s, a, r, s2, d = self._sample(0)
            self.learner.phase_a(s, a, r, s2, d, fuse_adam=False)
        else:
            self._learn_all()

    def _capture_lagged(self, steps):
        """`steps` vector steps in the pipelined order as TWO chains that never join inside the graph:
            B (side stream):  [opening launch: image + cursor + batch] learn()   [opening launch] learn()   ...
            A (this stream):            policy, env step                                policy, env step     ...
        with one hand-over each way per step: A(t) waits for the opening launch of step t (its image and cursor), and the opening
        launch of step t waits for the env step of step t-2 -- not t-1: the batch of step t holds transitions up to step t-2
        (_PIPE_LAG) and the image / cursor of step t go to the buffers of t's parity, so the opening launch and learn() of
        step t may run beside the policy and env launches of step t-1.  A common join per step put two cross-queue
        hand-overs and the opening launch on every step's critical path: 0.104 -> 0.096 ms per step at N = 65536."""
        cur, side = torch.cuda.current_stream(self.device), self._pipe_side
        side.wait_stream(cur)
        stepped = []
        # A(t) after the opening launch of step t: through DEVICE MEMORY (the pack launch publishes the step's image epoch,
        # the policy launch waits for it: include/ttenv.h, "image hand-over"), not through a graph edge -- the queue stopped
        # 9 us per step at that edge's wait packet although the image had always been ready for ~60 us.
        # TT_POLICY_EDGE=graph puts the edge back (runs under a tool that serialises kernels, e.g. rocprofv3 --pmc: a policy
        # launch waiting for a pack launch queued BEHIND it would only leave by its 0.25 s limit).
        # With several updates per step the policy waits for milliseconds: a waiting policy launch would sit on its 171 CUs all
        # that time (learn() 5 us per update slower beside it), so there the launch is held back by the edge.
        # Data-parallel ranks keep the edge as well: that path has never run on more than one GPU, and a collective that takes
        # long inside the learn chain must never meet a policy launch with a time limit.
        # ... and so do the short graphs (the 4-step and the single-step one): with the device-memory hand-over a graph keeps an
        # edge between its chains only every _LEARN_EDGE_EVERY-th step from step 2 on, so a graph that short would hold two chains
        # with NO edge at all -- the shape that starts ~0.2 ms late into an idle GPU (below) and that lets the policy wait for
        # learn packets which are not queued yet.  They serve the remainders of a run(k), not its bulk.
        edge = self.policy_edge() == "graph" or steps <= _LEARN_EDGE_EVERY
        for t in range(steps):
            with torch.cuda.stream(side):
                # (without the hand-over through device memory: every step; with it: every _LEARN_EDGE_EVERY-th step, see below)
                if t >= 2 and (edge or t % _LEARN_EDGE_EVERY == 0):
                    side.wait_event(stepped[t - 2])
                if edge:
                    self._open_step(True)
                    opened = torch.cuda.Event()
                    opened.record(side)
                    self._learn_all(presampled=True)
                else:
                    # no opening launch: learn()'s first launch makes the step's draw, its second one carries the pack of the
                    # step's image (weights as learn() of step t-1 left them: the actor is next written by the LAST launch of
                    # this learn()) -- one launch and one dependent boundary less on the learn chain
                    # (a pack launch of its own for the FIRST step of a graph, so that its policy launch starts ~15 us earlier,
                    # made no measurable difference: 0.0925 ms either way on one box)
                    # ... and B(t) after the env step of t-2, through device memory as well: its first launch waits for the step
                    # chain's progress word (the policy launch of step t-1 has begun: include/ttenv.h, TT_CURSOR_PROGRESS) -- the
                    # wait packet of that graph edge, always satisfied, cost 3.4 us of every step (0.0923 -> 0.0889 ms).
                    # One step in _LEARN_EDGE_EVERY keeps the edge all the same (redundant as an ordering): a graph of two
                    # chains with NO edge between them starts 0.2 ms late when it is launched into an idle GPU (one 20-step
                    # replay after a synchronize -- the driver's timed region: 0.1005 ms per step against 0.0894 back to
                    # back; presumably the runtime submits such a graph chain after chain, and the policy launch waits for a
                    # learn chain whose packets are still on their way).  With an edge every 3rd..6th step: 0.088-0.092 for
                    # the single replay, 0.089-0.090 back to back (tools/driver_form.py).
                    self._learn_all(presampled=False, with_image=True, wait_for_steps=True)
            if edge:
                cur.wait_event(opened)
            self._act_and_step()
            ev = torch.cuda.Event()
            ev.record(cur)
            stepped.append(ev)
        cur.wait_stream(side)

    def _capture(self, fn, side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode=_CAPTURE_MODE):
            fn()
        return g

    def _capture_step_graphs(self):
        ring, G = self.ring, self.graph_steps
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        ring._batch_bufs(self.batch_size)       # (allocated before the capture; kept per batch size for the ring's lifetime)
        ring._batch_bufs(self.batch_size * self._draws_per_opening())
        lagged = self.pipeline and not (self.dp and not self.dp_single_graph)
        many = (lambda k: (lambda: self._capture_lagged(k))) if lagged else (lambda k: (lambda: [self._capture_body() for _ in range(k)]))
        self.graph1 = self._capture(many(1), side)
        self.graphG = self._capture(many(G), side) if G > 1 else None
        # a launch of a graph costs ~12 us whatever it holds: long graphs for the bulk, a 4-step one for what is left over
        self.graphM = self._capture(many(4), side) if G > 4 else None
        if self.dp and not self.dp_single_graph:
            s = ring._batch_bufs(self.batch_size)[0]
            self.dp_graphs = {}
            pieces = {"b": lambda: self.learner.phase_b(s, separate_adam=True), "c": self.learner.phase_c}
            for u in range(1, self.updates_per_step):      # the further updates of a step: sample + up to the critic's gradient
                last = self.pipeline and u == self.updates_per_step - 1
                pieces[("a", u)] = (lambda u=u, last=last: self.learner.phase_a(
                    *self._sample(u), fuse_adam=False, window_dev=self.k_pipe_dev if last else None))
            for name, fn in pieces.items():
                self.dp_graphs[name] = self._capture(fn, side)
        torch.cuda.current_stream().wait_stream(side)

    def _dp_step(self):
        """One data-parallel vector step: three graph replays per update with the reference's two optimizer-site
        all-reduces (DDPG_agent.py:95-104) between them, launched eagerly on the flat gradient buffers."""
        self.graph1.replay()
        for u in range(self.updates_per_step):
            if u:
                self.dp_graphs[("a", u)].replay()
            self.learner.grad_sync_critic()
            self.dp_graphs["b"].replay()
            self.learner.grad_sync_actor()
            self.dp_graphs["c"].replay()

    def _try_capture(self):
        import gc
        # No garbage collection while a capture is open: a collected hipGraph, stream, event or env handle of some EARLIER
        # object runs HIP calls in its destructor that are illegal during capture (torch aborts the process).  Collect
        # first, then keep the collector off until the last graph is captured.
        gc.collect()
        was_on = gc.isenabled()
        gc.disable()
        try:
            self._capture_step_graphs()
            return True
        except Exception as exc:        # capture refused (driver / library state): the eager path is the same bits.
            import warnings             # (covers capture ERRORS only: a crash inside the runtime is not an exception)
            warnings.warn(f"whole-step hipGraph capture failed ({exc!r}); continuing with eager steps")
            self.invalidate_graphs()
            self.graph_steps = 0
            return False
        finally:
            if was_on:
                gc.enable()

    def prepare(self):
        """Everything one-off that run() would otherwise do lazily inside its first calls (a few eager vector steps that
        warm up allocators / kernel attributes / Adam state, then the graph captures), so that a timed region holds
        steady-state steps only.  Advances the loop by 4 vector steps."""
        while self.ring.k < 4:
            self.step()
        self._check_handover(exact=True)
        if self.graph_steps and not self._graphs_current():
            self._try_capture()

    def first_launches(self):
        """Replay every captured graph once (its first launch uploads it: ~0.2 ms more than any later one), as part of the
        setup: advances the loop by graph_steps (+ 4 when graph_steps > 4) + 1 vector steps.  Returns that number."""
        if not (self.graph_steps and self._graphs_current()):
            return 0
        k = 1 if (self.dp and not self.dp_single_graph) else self.graph_steps + (4 if self.graphM is not None else 0) + 1
        self.run(k)
        return k

    def run(self, k):
        """k vector steps, every one a graph replay once the loop is warm (4 eager steps) when whole-step graphs are on:
        the graph of graph_steps steps while that many remain, the single-step graph for the rest; eager step() otherwise."""
        ring = self.ring
        if self.learner is not None:
            self.learner.refresh_images()      # fc2 written by anyone but the learner's own launches since the last look?
        while k > 0:
            self._check_handover()             # (a host-memory read; a give-up drops the graphs: captured again just below)
            G = self.graph_steps
            if G and ring.k >= 4 and (self._graphs_current() or self._try_capture()):
                if self.dp and not self.dp_single_graph:
                    self._dp_step()
                    done = 1
                elif G > 1 and k >= G:
                    self.graphG.replay()
                    done = G
                elif self.graphM is not None and k >= 4:
                    self.graphM.replay()
                    done = 4
                else:
                    self.graph1.replay()
                    done = 1
                ring.k += done                  # host mirror; the step kernels advanced k_dev
                self.vector_steps += done
                k -= done
            else:
                self.step()
                k -= 1
        self._check_handover()

    # -------------------------------------------------------------- checkpoint / resume of the whole loop
    def state_dict(self):
        """Everything the next vector step depends on: the four networks, both optimizers' state (the
        fused learner's flat Adam moments and step count, or the torch optimizers'), the env batch, the replay ring with
        its counters, the OU state.  The RNG streams of the loop are counter-based (Philox keyed by seed and the
        counters saved here), so there is no generator state to save."""
        torch.cuda.synchronize(self.device) if self.device.type == "cuda" else None
        self._check_handover(exact=True)
        ag = self.agent
        sd = {"format": 2, "seed": int(self.seed), "vector_steps": int(self.vector_steps),
              "handover_gave_up": [int(x) for x in self.handover_gave_up],
              "batch_size": int(self.batch_size), "updates_per_step": self.updates_per_step,
              "nets": {n: {k: v.detach().cpu().clone() for k, v in getattr(ag, n).state_dict().items()}
                       for n in ("actor", "critic", "target_actor", "target_critic")},
              "ring": self.ring.state_dict(), "ou": self.noise.x.detach().cpu().clone(),
              "env": self.env.state_dict() if hasattr(self.env, "state_dict") else None}
        if self.learner is not None:
            sd["fused_adam"] = self.learner.state_dict()
        else:
            sd["optim"] = {"actor": ag.actor.optimizer.state_dict(), "critic": ag.critic.optimizer.state_dict()}
        return sd

    def load_state_dict(self, sd):
        ag = self.agent
        assert int(sd["batch_size"]) == int(self.batch_size), "batch size differs"
        with torch.no_grad():
            for n, net_sd in sd["nets"].items():
                for k, v in getattr(ag, n).state_dict().items():      # in place: captured graphs keep the addresses
                    v.copy_(net_sd[k].to(v.device))
        if self.learner is not None and "fused_adam" in sd:
            self.learner.load_state_dict(sd["fused_adam"])
        elif "optim" in sd:
            ag.actor.optimizer.load_state_dict(sd["optim"]["actor"])
            ag.critic.optimizer.load_state_dict(sd["optim"]["critic"])
            self.graph = None                                         # optimizer state tensors were replaced
        self.ring.load_state_dict(sd["ring"])
        self.noise.x.copy_(sd["ou"].to(self.noise.x.device))
        if sd.get("env") is not None:
            self.env.load_state_dict(sd["env"])                       # bumps env.graph_epoch: graphs are re-captured
        self.k_pipe_dev.fill_(self.ring.k)
        if int(sd["seed"]) != int(self.seed):
            self.invalidate_graphs()                                  # the Philox keys are kernel arguments
        self.seed = int(sd["seed"])
        self.handover_gave_up = [int(x) for x in sd.get("handover_gave_up", [])]
        self.vector_steps = int(sd["vector_steps"])
        self.updates_per_step = int(sd.get("updates_per_step", self.updates_per_step))
