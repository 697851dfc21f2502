"""GPU (-m gpu): the DDPG loop on top of the HIP env -- the reference's single-env trainv2 loop through the
gym facade (BASELINE.json config 1 plumbing) and the N-env vector loop with the device replay ring and the
hipGraph-captured learn() (config 3)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_trainv2_shaped_single_env_loop(gpu_device):
    """DDPG/trainv2.py:488-531 verbatim in shape: reset(seed+i), noise.reset, choose_action, clip*high, step,
    remember(unclipped action), learn."""
    import torch
    from ddpg_trucktrailer_amd.agent import Agent
    from ddpg_trucktrailer_amd.env import Truck_trailer_Env_2
    env = Truck_trailer_Env_2()
    torch.manual_seed(27); np.random.seed(27)
    agent = Agent(alpha=0.0001, beta=0.001, input_dims=env.observation_space.shape, tau=0.001, batch_size=64,
                  fc1_dims=400, fc2_dims=300, n_actions=env.action_space.shape[0], max_size=5000)
    assert env.reward_range[0] == -float("inf")
    total = 0
    before = [p.detach().clone() for p in agent.actor.parameters()]
    for i in range(2):
        observation, info = env.reset(seed=27 + i)
        assert observation.shape == (23,) and observation.dtype == np.float32 and info == {}
        if i == 0:   # seed 27 -> the pose fixture F2 recorded from the reference
            assert abs(env.startx - -4.011043831979627) < 1e-12 and env.max_episode_steps == 205
        done, score = False, 0
        agent.noise.reset()
        while not done:
            action = agent.choose_action(observation)
            scaled_action = np.clip(action, -1, 1) * env.action_space.high
            observation_, reward, done, info = env.step(scaled_action)
            agent.remember(observation, action, reward, observation_, done)
            agent.learn()
            score += reward
            observation = observation_
            total += 1
        assert np.isfinite(score) and info["violation_type"] in ("jackknife", "minor_boundary", "major_boundary",
                                                                  "past_the_goal", "max_step", "excessive_backward", "none")
    assert total >= 64 and agent.memory.mem_cntr == total
    assert any(not torch.equal(a, b) for a, b in zip(before, agent.actor.parameters())), "learn() never ran"
    env.close()


@pytest.mark.parametrize("use_graph", [False, True])
def test_vector_loop_ring_and_learn(gpu_device, use_graph):
    import torch
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    n = 4096
    env = TruckTrailerVecEnv(n)
    env.reset(seed=27)
    loop = DDPGRollout(env, batch_size=256, replay_slots=8, seed=27, use_graph=use_graph, pipeline=False)
    first_obs = loop.ring.obs[0].clone()
    assert torch.equal(first_obs, env.observe(out=torch.empty_like(first_obs)))
    # detach(): a clone that keeps its grad_fn pins the parameter's AccumulateGrad node to THIS stream, and a
    # backward captured on another stream then forks the capture (HIP's EndCapture crashes on that)
    w0 = [p.detach().clone() for p in loop.agent.critic.parameters()]
    t0 = [p.detach().clone() for p in loop.agent.target_actor.parameters()]
    dones = 0
    for k in range(20):
        t, t1 = loop.ring.slot(), loop.ring.slot(loop.ring.k + 1)
        loop.step()
        ring = loop.ring
        a, d = ring.act[t], ring.done[t].bool()
        # the env was driven with clip(a, -1, 1) * f32(pi/4) and the UNCLIPPED action was stored (trainv2.py:516,525)
        assert torch.equal(loop.scaled, torch.clamp(a, -1, 1) * np.float32(np.pi / 4))
        assert (a.abs() > 0).all() and torch.isfinite(ring.rew[t]).all()
        if k > 0:                                                                # noise restarts with the episode:
            dp = ring.done[loop.ring.slot(loop.ring.k - 2)].bool()              # envs done at the PREVIOUS step drew
            assert (loop.noise.x[dp].abs() <= 0.015 * 6).all()                  # their first sample from x = 0
        cur = env.observe(steering=loop.scaled, out=torch.empty_like(first_obs))
        # s' of running envs = env's observation.  k_step forms sin/cos of the hitch angle from the sin/cos of the two
        # headings, k_observe from the angle itself: the same f64 value to ~1e-16, which can round to neighbouring f32
        # when the hitch angle is ~1e-8 rad -- hence not torch.equal (the parity bound on observations is 1e-5)
        assert (cur[~d] - ring.obs[t1][~d]).abs().max().item() <= 1e-7
        if d.any():
            assert torch.equal(env.observe(out=torch.empty_like(first_obs))[d], ring.obs[t1][d])   # fresh episode, steering 0
        dones += int(d.sum())
    assert loop.ring.k == 20 and len(loop.ring) == 7 * n
    torch.cuda.synchronize()
    assert any(not torch.equal(a, b) for a, b in zip(w0, loop.agent.critic.parameters()))
    assert any(not torch.equal(a, b) for a, b in zip(t0, loop.agent.target_actor.parameters()))
    for p in list(loop.agent.actor.parameters()) + list(loop.agent.critic.parameters()):
        assert torch.isfinite(p).all()
    assert (loop.graph is not None) == use_graph
    env.close()


def test_graph_learn_equals_eager_learn(gpu_device):
    """One captured learn() replay == one eager learn() from the same weights, batch and Adam state."""
    import torch
    from ddpg_trucktrailer_amd.agent import Agent
    z = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "f5_learner.npz"), allow_pickle=False)
    from test_learner import _load_init, _batch
    agents = []
    for _ in range(2):
        a = Agent(alpha=1e-4, beta=1e-3, input_dims=(23,), tau=1e-3, n_actions=1, batch_size=256, device=gpu_device,
                  capturable=True, replay=False)
        _load_init(a, z)
        agents.append(a)
    batch = _batch(z, gpu_device)
    eager, graphed = agents
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graphed.learn_batch(*batch)        # warm-up step 1 (eager) on both
    side.synchronize()
    eager.learn_batch(*batch)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        graphed.learn_batch(*batch)
    torch.cuda.current_stream().wait_stream(side)
    g.replay()                            # capture does not execute: this replay is step 2
    eager.learn_batch(*batch)
    torch.cuda.synchronize()
    for name in ("actor", "critic", "target_actor", "target_critic"):
        for (k, x), y in zip(getattr(eager, name).state_dict().items(), getattr(graphed, name).state_dict().values()):
            assert torch.allclose(x, y, rtol=1e-5, atol=1e-7), (name, k)


def test_fused_ring_sampling(gpu_device):
    """tt_ring_sample: every drawn transition is intact ring content, indices are uniform over the valid window,
    and consecutive replays (k_dev advanced) draw different indices."""
    import torch
    from ddpg_trucktrailer_amd.replay_buffer import TrajectoryRing
    n, slots, B = 1000, 6, 4096
    ring = TrajectoryRing(n, slots, 23, gpu_device)
    for k in range(9):
        t, t1 = ring.slot(), ring.slot(ring.k + 1)
        if k == 0:
            ring.obs[t].fill_(0.0)
        lane = torch.arange(n, device=gpu_device, dtype=torch.float32)
        ring.act[t] = k * 10000 + lane
        ring.rew[t] = -(k * 10000 + lane)
        ring.done[t] = ((torch.arange(n, device=gpu_device) + k) % 7 == 0).to(torch.uint8)
        ring.obs[t1] = (k + 1) + lane.unsqueeze(1) / 4096 + torch.arange(23, device=gpu_device) / 100000
        ring.advance()
        s, a, r, s2, d, idx = ring.sample_fused(B, seed=5, return_index=True)
        step = torch.div(a[:, 0], 10000, rounding_mode="floor")
        e = a[:, 0] - step * 10000
        assert torch.equal(r, -a[:, 0])
        assert step.min() >= max(0, k - (slots - 2)) and step.max() <= k
        if k >= 1:
            ok = step >= 1
            assert torch.allclose(s[ok][:, 0], step[ok] + e[ok] / 4096, atol=1e-4)
        assert torch.allclose(s2[:, 0], step + 1 + e / 4096, atol=1e-4)
        assert torch.equal(d, ((e.long() + step.long()) % 7 == 0))
        assert torch.equal(idx[:, 1].long(), e.long()) and torch.equal(idx[:, 0].long(), step.long() % slots)
    # uniformity over envs and over the 5 valid steps
    hist_e = torch.bincount(idx[:, 1].long() // 100, minlength=10).float()
    hist_t = torch.bincount((step - step.min()).long(), minlength=slots - 1).float()
    assert (hist_e / B - 0.1).abs().max() < 0.03 and (hist_t / B - 0.2).abs().max() < 0.04
    again = ring.sample_fused(B, seed=5, return_index=True)[5].clone()
    ring.advance()
    assert not torch.equal(again, ring.sample_fused(B, seed=5, return_index=True)[5])


def test_several_draws_in_the_opening_launch(gpu_device):
    """tt_mlp_split_pack_and_sample with tt_sample_args.draws = 3: rows [u B, (u + 1) B) of the buffers are exactly what
    tt_ring_sample draws with seed + u * seed_stride (same window, same side tuples), including the index pairs."""
    import torch
    from ddpg_trucktrailer_amd import fused
    from ddpg_trucktrailer_amd.agent import Agent
    from ddpg_trucktrailer_amd.replay_buffer import TrajectoryRing
    n, slots, B, U, stride = 500, 7, 96, 3, 0x9E3779B97F4A7C15
    g = torch.Generator(device="cpu").manual_seed(3)
    ring = TrajectoryRing(n, slots, 23, gpu_device)
    ring.obs.copy_(torch.rand(ring.obs.shape, generator=g)); ring.act.copy_(torch.rand(ring.act.shape, generator=g))
    ring.rew.copy_(torch.rand(ring.rew.shape, generator=g)); ring.done.copy_((torch.rand(ring.done.shape, generator=g) < 0.2).to(torch.uint8))
    ring.load_side(torch.rand((40, 23), generator=g), torch.rand(40, generator=g), torch.rand(40, generator=g),
                   torch.rand((40, 23), generator=g), torch.rand(40, generator=g) < 0.5)
    for _ in range(9):
        ring.advance()
    actor = Agent(1e-4, 1e-3, (23,), 1e-3, 1, batch_size=B, device=gpu_device, replay=False).actor
    fused.pack_and_sample(actor, 0, ring.sample_args(B, seed=11, reserve=2, lag=1, draws=U, seed_stride=stride))
    big = [x.clone() for x in ring._batch_bufs(U * B)]
    for u in range(U):
        one = ring.sample_fused(B, seed=(11 + u * stride) & (2 ** 64 - 1), return_index=True, done_as_bool=False, reserve=2, lag=1)
        for x, y in zip(big, one):
            assert torch.equal(x[u * B:(u + 1) * B].reshape(B, -1), y.reshape(B, -1))
    assert (big[5][:, 0] == -1).any() and (big[5][:, 0] >= 0).any()          # side tuples and ring rows both took part
    assert not torch.equal(big[0][:B], big[0][B:2 * B])


def test_whole_step_graphs_match_eager_steps(gpu_device):
    """DDPGRollout.run(k) replays hipGraphs of whole vector steps (policy + env step + learn); step() launches the same
    vector step eagerly.  Everything that varies per step lives on the device (ring counter, Philox counters), so the two
    must agree bit for bit: ring contents, env state, noise state, all four networks."""
    import torch
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    n, k = 2048, 26
    loops = []
    for graph_steps in (4, 0):
        env = TruckTrailerVecEnv(n)
        env.reset(seed=5)
        loop = DDPGRollout(env, batch_size=256, replay_slots=8, seed=5, use_graph=True, graph_steps=graph_steps)
        loops.append(loop)
    a, b = loops
    assert a.graph_steps == 4 and b.graph_steps == 0
    a.run(k)                       # 4 eager steps, 5 graphs of 4, 2 eager steps
    for _ in range(k):
        b.step()
    torch.cuda.synchronize()
    assert a.graph1 is not None and a.graphG is not None      # ONE single-step graph and ONE 4-step graph serve all positions
    assert a.ring.k == b.ring.k == k and int(a.ring.k_dev.item()) == int(b.ring.k_dev.item()) == k
    for name in ("obs", "act", "rew", "done"):
        assert torch.equal(getattr(a.ring, name), getattr(b.ring, name)), name
    assert torch.equal(a.noise.x, b.noise.x) and torch.equal(a.env.state, b.env.state)
    for net in ("actor", "critic", "target_actor", "target_critic"):
        for x, y in zip(getattr(a.agent, net).state_dict().values(), getattr(b.agent, net).state_dict().values()):
            assert torch.equal(x, y), net
    assert int(a.learner.step_dev.item()) == int(b.learner.step_dev.item()) >= k - 2    # learn() from the 2nd / 3rd step on
    for lp in loops:
        lp.env.close()


def _loop_flat(loop):
    import torch
    return torch.cat([p.detach().reshape(-1) for net in loop.agent._nets() for p in net.parameters()])


def test_run_is_all_graph_replays_for_any_warmup_and_steps(gpu_device, monkeypatch):
    """bench.py's driver form is --steps 20 --warmup 5: after prepare() no step of run() may be launched eagerly
    whatever the ring position (the launches find their ring slots through a device cursor, so one 4-step graph and one
    single-step graph serve every position and remainder), and the result is the eager loop's, bit for bit."""
    import torch
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    n = 1024
    loops = []
    for graph_steps in (4, 20, 0):
        env = TruckTrailerVecEnv(n)
        env.reset(seed=9)
        loops.append(DDPGRollout(env, batch_size=256, replay_slots=8, seed=9, use_graph=True, graph_steps=graph_steps))
    a, a20, b = loops

    def no_eager():
        raise AssertionError("run() launched a vector step eagerly")
    for lp in (a, a20):
        lp.prepare()
        assert lp.ring.k == 4 and lp.graph1 is not None and lp.graphG is not None
        assert (lp.graphM is not None) == (lp.graph_steps > 4)      # bench.py's default: graphs of 20, 4 and 1 steps
        monkeypatch.setattr(lp, "step", no_eager)
        lp.run(5)            # warm-up of 5: one 4-step graph + one single
        lp.run(20)           # five 4-step graphs / one 20-step graph -- from ring position 9 of 8 slots: position-independent
        lp.run(3)
    for _ in range(4 + 5 + 20 + 3):
        b.step()
    torch.cuda.synchronize()
    for lp in (a, a20):
        assert lp.ring.k == b.ring.k == 32 and int(lp.ring.k_dev.item()) == 32
        assert torch.equal(_loop_flat(lp), _loop_flat(b))
        for name in ("obs", "act", "rew", "done"):
            assert torch.equal(getattr(lp.ring, name), getattr(b.ring, name)), name
        assert torch.equal(lp.env.state, b.env.state)
    for lp in loops:
        lp.env.close()


def test_updates_per_step(gpu_device):
    """updates_per_step = 3: three learn() calls per vector step, each on its own batch; graphs == eager."""
    import torch
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    loops = []
    for graph_steps in (4, 0):
        env = TruckTrailerVecEnv(512)
        env.reset(seed=3)
        loops.append(DDPGRollout(env, batch_size=64, replay_slots=8, seed=3, graph_steps=graph_steps, updates_per_step=3))
    a, b = loops
    a.run(14)
    for _ in range(14):
        b.step()
    torch.cuda.synchronize()
    assert int(a.learner.step_dev.item()) == int(b.learner.step_dev.item()) == 3 * 12      # pipelined order: learn() from the 3rd step on
    assert torch.equal(_loop_flat(a), _loop_flat(b))
    s0 = a._sample(0)[0].clone(); s1 = a._sample(1)[0].clone()
    assert not torch.equal(s0, s1), "the updates of one vector step must draw different batches"
    for lp in loops:
        lp.env.close()


@pytest.mark.parametrize("side", [False, True])
def test_all_batches_of_a_step_from_its_opening_launch(gpu_device, monkeypatch, side):
    """Several updates per step: the opening launch of a step makes ALL of the step's draws (tt_sample_args.draws), each into its
    rows of one buffer -- bit for bit the batches the updates' own draws (learn()'s first launch, TT_MULTI_DRAW=0) would be, so
    the loop's state after 14 steps is the same either way, in graphs and eagerly; with expert tuples in the draw as well."""
    import torch
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    loops = []
    for multi, graph_steps in (("1", 4), ("0", 4), ("1", 0)):
        monkeypatch.setenv("TT_MULTI_DRAW", multi)
        env = TruckTrailerVecEnv(4096)
        env.reset(seed=5)
        lp = DDPGRollout(env, batch_size=256, replay_slots=8, seed=5, graph_steps=graph_steps, updates_per_step=4)
        assert lp._draws_per_opening() == (4 if multi == "1" else 1)
        if side:
            g = torch.Generator().manual_seed(1)
            k = 3000
            lp.ring.load_side(torch.rand((k, 23), generator=g), torch.rand(k, generator=g) * 2 - 1, -torch.rand(k, generator=g),
                              torch.rand((k, 23), generator=g), torch.rand(k, generator=g) < 0.1)
        if graph_steps:
            lp.run(14)
        else:
            for _ in range(14):
                lp.step()
        loops.append(lp)
    torch.cuda.synchronize()
    a, b, c = loops
    assert int(a.learner.step_dev.item()) == int(b.learner.step_dev.item()) == int(c.learner.step_dev.item()) == 4 * 12
    assert torch.equal(_loop_flat(a), _loop_flat(b))
    assert torch.equal(_loop_flat(a), _loop_flat(c))
    # the rows of the big buffer are the four batches: draw u of the last step == the draw an update makes on its own
    B = 256
    big = a.ring._batch_bufs(4 * B)[0].clone()
    assert not torch.equal(big[:B], big[B:2 * B])
    for lp in loops:
        lp.env.close()


def test_torch_learn_path_is_never_captured(gpu_device):
    """DDPGRollout(fused_learn=False, use_graph=True): learn() through torch autograd.  A clone of a parameter that still
    carries its grad_fn, made on ANOTHER stream, keeps that parameter's gradient accumulator pinned there; a captured
    backward then pulls that stream into the capture and HIP's EndCapture crashes the process (seen in round 1 and again
    with torch.autograd.grad).  The loop therefore never captures this learner: use_graph only covers the fused one, and
    the two settings give the same weights."""
    import torch
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    loops = []
    for use_graph in (True, False):
        env = TruckTrailerVecEnv(256)
        env.reset(seed=4)
        loop = DDPGRollout(env, batch_size=64, replay_slots=8, seed=4, use_graph=use_graph, fused_learn=False)
        assert loop.learner is None and not loop.pipeline
        loops.append(loop)
    a, b = loops
    keep = [p.clone() for p in a.agent.critic.parameters()] + [p.clone() for p in a.agent.actor.parameters()]
    assert all(k.grad_fn is not None for k in keep)
    for _ in range(9):
        a.step(); b.step()
    torch.cuda.synchronize()
    assert a.graph is None and b.graph is None
    assert torch.equal(_loop_flat(a), _loop_flat(b)) and torch.isfinite(_loop_flat(a)).all()
    del keep
    for lp in loops:
        lp.env.close()


def test_whole_loop_resume_is_bitwise(gpu_device, tmp_path):
    """save after 10 vector steps, continue 9 more; a FRESH loop that loads the file and runs the same 9
    steps ends bit-identical: env batch, ring + counters, OU state, all four networks, the fused Adam state."""
    import torch
    from ddpg_trucktrailer_amd import checkpoint
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv

    def make(seed):
        env = TruckTrailerVecEnv(1536)
        env.reset(seed=seed)
        return DDPGRollout(env, batch_size=128, replay_slots=8, seed=seed, graph_steps=4)
    a = make(21)
    a.run(10)
    path = checkpoint.save_loop_checkpoint(str(tmp_path / "loop.pt"), a, training_state={"episode_num": 7})
    a.run(9)
    b = make(99)                       # different seed, different poses, different weights: all must come from the file
    b.run(6)                           # ... and its graphs are already captured when the file is loaded
    ts = checkpoint.load_loop_checkpoint(path, b)
    assert ts == {"episode_num": 7} and b.ring.k == 10 and b.vector_steps == 10
    b.run(9)
    torch.cuda.synchronize()
    assert torch.equal(_loop_flat(a), _loop_flat(b))
    for name in ("obs", "act", "rew", "done"):
        assert torch.equal(getattr(a.ring, name), getattr(b.ring, name)), name
    assert torch.equal(a.noise.x, b.noise.x) and torch.equal(a.env.state, b.env.state)
    assert int(a.ring.k_dev.item()) == int(b.ring.k_dev.item()) == 19
    for st_a, st_b in ((a.learner.actor, b.learner.actor), (a.learner.critic, b.learner.critic)):
        assert torch.equal(st_a.m, st_b.m) and torch.equal(st_a.v, st_b.v)
    assert int(a.learner.step_dev.item()) == int(b.learner.step_dev.item())
    ea, eb = a.env.episode(), b.env.episode()
    assert torch.equal(ea["steps"], eb["steps"]) and torch.equal(ea["start"], eb["start"])
    a.env.close(); b.env.close()


@pytest.mark.parametrize("graph_steps,steps", [(4, 12), (20, 28)])
def test_whole_config3_loop_at_bench_size(gpu_device, graph_steps, steps):
    """BASELINE config 3 at its stated size (N = 65536, batch 256, 64-slot ring): the steps as graphs == the same number of
    eager steps bit for bit, finite weights, learn() really ran, both device counters in step.  graph_steps = 20 is THE graph
    bench.py's headline number replays -- 20 whole steps, three capped policy grids per step beside learn()'s launches on
    the second chain: 4 eager warm-up steps, then one 20-step graph and one 4-step graph."""
    import torch
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    n = 65536
    flats = []
    for g in (graph_steps, 0):
        env = TruckTrailerVecEnv(n)
        env.reset(seed=27)
        loop = DDPGRollout(env, batch_size=256, replay_slots=64, seed=27, graph_steps=g)
        w0 = _loop_flat(loop).clone()
        loop.run(steps)                # 4 eager warm-up steps + the graphs
        torch.cuda.synchronize()
        if g:
            assert loop.graphG is not None and (loop.graphM is not None) == (g > 4)
        flat = _loop_flat(loop)
        assert torch.isfinite(flat).all() and not torch.equal(flat, w0)
        assert torch.isfinite(loop.ring.rew[:steps]).all() and int(loop.learner.step_dev.item()) == steps - 2
        assert int(loop.ring.k_dev.item()) == steps and int(loop.k_pipe_dev.item()) == steps
        assert loop.ring.policy_gave_up() == 0          # no policy launch ran out of patience waiting for its image
        flats.append((flat.clone(), loop.ring.obs[:steps + 1].clone(), loop.ring.act[:steps].clone(), loop.env.state.clone(),
                      loop.noise.x.clone()))
        env.close()
        del loop
    for x, y in zip(*flats):
        assert torch.equal(x, y)


def test_image_handover_through_device_memory_over_many_steps(gpu_device):
    """The policy launch of a step takes its image and cursor from the step's opening pack launch with NO dependency between the
    two launches in the captured graphs (include/ttenv.h, "image hand-over": epoch words next to the cursor; the learn chain
    runs ~60 us ahead, under load, on other CUs and XCDs than the readers, which re-read the same two image buffers every other
    step -- warm in their L1 / L2).  1500 graph-replayed steps at the bench size == 1500 eager steps (where the two launches
    are ordered on one stream), bit for bit: one stale fragment anywhere in 1500 x 512 tile forwards would move an action, and
    with it the env state, the ring and every weight after it."""
    import torch
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    n, steps = 65536, 1500
    outs = []
    for g in (20, 0):
        env = TruckTrailerVecEnv(n)
        env.reset(seed=3)
        loop = DDPGRollout(env, batch_size=256, replay_slots=16, seed=3, graph_steps=g)
        loop.run(steps)
        torch.cuda.synchronize()
        assert loop.ring.policy_gave_up() == 0
        outs.append((_loop_flat(loop).clone(), loop.env.state.clone(), loop.noise.x.clone(), loop.ring.act.clone(), loop.ring.rew.clone()))
        env.close()
        del loop
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    assert torch.isfinite(outs[0][0]).all()


def test_run_notices_a_give_up_and_falls_back_to_graph_edges(gpu_device, monkeypatch, tmp_path):
    """A launch that gives up waiting for the other chain of its step (0.25 s) goes on with stale inputs and leaves a mark
    (include/ttenv.h: TT_CURSOR_GAVE_UP + its host-visible mirror).  DDPGRollout.run() -- not only bench.py -- must notice it
    without a synchronize in the loop: warn, capture the steps again with graph edges between the chains and go on; results
    then equal a loop that had graph edges from the start, bit for bit (the mark is set artificially: no launch stalled, so no
    step is actually stale); a second mark raises; a checkpoint of such a loop is refused.  Reference semantics at stake: the
    policy acts with the weights learn() of step t-1 left (DDPG/trainv2.py:511-531)."""
    import warnings
    import torch
    from ddpg_trucktrailer_amd import checkpoint
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    monkeypatch.delenv("TT_POLICY_EDGE", raising=False)

    def make(**kw):
        env = TruckTrailerVecEnv(4096)
        env.reset(seed=5)
        return DDPGRollout(env, batch_size=256, replay_slots=16, seed=5, graph_steps=20, **kw)
    a = make()
    if a.policy_edge() != "flag":
        pytest.skip("the loop already uses graph edges here (a tool is attached)")
    a.run(4 + 20 + 4 + 1)
    torch.cuda.synchronize()
    assert a.handover_gave_up == [] and a.ring.gave_up_seen() == 0
    a.ring.mark_gave_up_for_test(a.ring.k - 1)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        a.run(20 + 4 + 1)                                  # notices at its first look, re-captures, goes on
    assert any("gave up waiting" in str(w.message) for w in seen)
    assert a.handover_gave_up == [29] and a.policy_edge() == "graph" and a.graphG is not None
    assert a.ring.policy_gave_up() == 0                    # both words cleared
    monkeypatch.setenv("TT_POLICY_EDGE", "graph")
    b = make()
    b.run(4 + 20 + 4 + 1 + 20 + 4 + 1)
    torch.cuda.synchronize()
    assert a.ring.k == b.ring.k == 54
    assert torch.equal(_loop_flat(a), _loop_flat(b)) and torch.equal(a.env.state, b.env.state)
    for name in ("obs", "act", "rew", "done"):
        assert torch.equal(getattr(a.ring, name), getattr(b.ring, name)), name
    # a checkpoint of a loop with a give-up in its history is refused; force=True writes it, history included
    with pytest.raises(RuntimeError, match="gave up waiting"):
        checkpoint.save_loop_checkpoint(str(tmp_path / "no.pt"), a)
    assert not (tmp_path / "no.pt").exists()
    path = checkpoint.save_loop_checkpoint(str(tmp_path / "forced.pt"), a, force=True)
    assert torch.load(path, weights_only=True)["loop"]["handover_gave_up"] == [29]
    checkpoint.save_loop_checkpoint(str(tmp_path / "ok.pt"), b)
    # state_dict() looks at the DEVICE word (exact): a mark the mirror never got is still found
    b.ring.cursor_dev[15] = 7
    with pytest.raises(RuntimeError, match="ordered by graph edges"):
        b.state_dict()
    monkeypatch.delenv("TT_POLICY_EDGE")
    # ... and with edges already in place a second give-up is an error in run() as well
    a.ring.mark_gave_up_for_test(a.ring.k - 1)
    with pytest.raises(RuntimeError, match="ordered by graph edges"):
        a.run(1)
    a.env.close(); b.env.close()


def test_a_launch_that_really_gives_up_reaches_the_host(gpu_device, monkeypatch):
    """The device side of the above: a policy launch whose image epoch never arrives leaves after 0.25 s and sets the give-up
    word AND its mirror in pinned host memory, which the host reads without any GPU call."""
    import time
    import torch
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    monkeypatch.delenv("TT_POLICY_EDGE", raising=False)
    env = TruckTrailerVecEnv(1024)
    env.reset(seed=2)
    loop = DDPGRollout(env, batch_size=256, replay_slots=16, seed=2, graph_steps=4)
    loop.prepare()
    torch.cuda.synchronize()
    k = loop.ring.k
    assert loop.ring.gave_up_seen() == 0
    loop.ring.cursor_dev[12:14] = 0                        # the epochs of both parities: "no image has ever been published"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.policy_launch()                                   # waits for epoch k + 1, bounded
    torch.cuda.synchronize()
    waited = time.perf_counter() - t0
    assert 0.2 < waited < 5.0, waited
    assert loop.ring.gave_up_seen() == k + 1               # host memory, written by the kernel (system scope)
    assert int(loop.ring.cursor_dev[15].item()) == k + 1
    loop.ring.clear_gave_up()
    assert loop.ring.policy_gave_up() == 0
    env.close()


def test_loop_with_the_actor_tail_in_one_launch(gpu_device, monkeypatch):
    """Where learn() bounds the step (N <= 16384, or several updates per step) the loop runs learn()'s last two launches as one grid
    (tt_mlp_actor_tail: dQ/da handed over in device memory inside the grid): four launches instead of five.  Same loop,
    TT_ACTOR_TAIL=0 / 1: every weight, the ring, the env state, the OU state after 4 + 20 + 4 + 1 + 6
    steps (eager warm-up, graphs of 20 / 4 / 1, sampled draws, the image pack as a rider), bit for bit; also with 3 updates per step."""
    import torch
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    for updates in (1, 3):
        outs = []
        for tail in ("0", "1"):
            monkeypatch.setenv("TT_ACTOR_TAIL", tail)
            env = TruckTrailerVecEnv(2048)
            env.reset(seed=11)
            loop = DDPGRollout(env, batch_size=256, replay_slots=16, seed=11, graph_steps=20, updates_per_step=updates)
            assert loop.learner.fuse_tail == (tail == "1")
            loop.run(4 + 20 + 4 + 1 + 6)
            torch.cuda.synchronize()
            assert loop.handover_gave_up == [] and loop.learner.tail_gave_up() == 0
            outs.append((_loop_flat(loop).clone(), loop.ring.obs.clone(), loop.ring.act.clone(), loop.ring.rew.clone(), loop.env.state.clone(),
                         loop.noise.x.clone(), loop.learner.actor.m.clone(), loop.learner.critic.v.clone()))
            env.close()
            del loop
        for x, y in zip(*outs):
            assert torch.equal(x, y), updates
        assert torch.isfinite(outs[0][0]).all()
    monkeypatch.delenv("TT_ACTOR_TAIL")
    env = TruckTrailerVecEnv(1024); env.reset(seed=1)
    assert DDPGRollout(env, batch_size=256, replay_slots=16, seed=1, graph_steps=4).learner.fuse_tail is True           # N <= 16384
    env.close()


def test_policy_edge_follows_the_grid_cap(gpu_device, monkeypatch):
    """A policy grid that is not capped below the CU count fills the chip (one workgroup per CU: 155 KB of LDS), and a launch
    spinning there for its image would keep the learn chain that makes the image off the GPU: such loops use graph edges."""
    import torch
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    monkeypatch.delenv("TT_POLICY_EDGE", raising=False)
    monkeypatch.delenv("TT_POLICY_WG", raising=False)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    env = TruckTrailerVecEnv(1024)
    env.reset(seed=2)
    tool = any(k.startswith("ROCPROF") or k == "ROCP_TOOL_LIBRARIES" for k in __import__("os").environ)
    for wg, want in ((192, "graph" if tool or 192 >= cus else "flag"), (0, "graph"), (cus, "graph"), (cus + 64, "graph")):
        loop = DDPGRollout(env, batch_size=256, replay_slots=16, seed=2, graph_steps=4, policy_workgroups=wg)
        assert loop.policy_edge() == want, (wg, cus)
    assert DDPGRollout(env, batch_size=256, replay_slots=16, seed=2, graph_steps=4, updates_per_step=2).policy_edge() == "graph"
    env.close()


@pytest.mark.parametrize("pipeline", [True, False])
def test_pipelined_and_serial_orders(gpu_device, pipeline):
    """Both orders of a vector step: graphs == eager bit for bit, exactly one learn() per step from the third step on,
    and in the pipelined order the batch of step t holds transitions of steps < t only while the policy of step t acts
    with the weights learn() of step t-1 left (its packed image)."""
    import torch
    from ddpg_trucktrailer_amd import fused
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    n, k = 2048, 23
    loops = []
    for graph_steps in (4, 0):
        env = TruckTrailerVecEnv(n)
        env.reset(seed=5)
        loops.append(DDPGRollout(env, batch_size=256, replay_slots=8, seed=5, graph_steps=graph_steps, pipeline=pipeline))
    a, b = loops
    assert a.pipeline == b.pipeline == pipeline
    a.run(k)
    for _ in range(k):
        b.step()
    torch.cuda.synchronize()
    assert torch.equal(_loop_flat(a), _loop_flat(b))
    for name in ("obs", "act", "rew", "done"):
        assert torch.equal(getattr(a.ring, name), getattr(b.ring, name)), name
    assert torch.equal(a.noise.x, b.noise.x) and torch.equal(a.env.state, b.env.state)
    want = k - 2 if pipeline else k - 1               # pipelined: learn() in steps 2..k-1; serial: after steps 1..k-1
    assert int(a.learner.step_dev.item()) == int(b.learner.step_dev.item()) == want
    if pipeline:
        assert int(a.k_pipe_dev.item()) == int(b.k_pipe_dev.item()) == k
        # the window of the next batch (step k): steps k-6 .. k-2 (slots - 3 = 5 of them) -- not step k-1, which may still be
        # under way beside the draw, and never a slot the env steps k-1 and k are writing
        from ddpg_trucktrailer_amd.rollout import _PIPE_LAG, _PIPE_RESERVE
        assert (_PIPE_LAG, _PIPE_RESERVE) == (1, 2)
        _, _, _, _, _, idx = a.ring.sample_fused(4096, seed=1, return_index=True, k_dev=a.k_pipe_dev, reserve=_PIPE_RESERVE,
                                                 lag=_PIPE_LAG)
        t = idx[:, 0].long()
        assert set(t.unique().tolist()) == {(k - 2 - j) % 8 for j in range(5)}
        writing = {(k - 1 + 1) % 8, (k + 1) % 8}                   # obs rows of the env steps k-1 and k
        assert not (set(t.tolist()) | set(((t + 1) % 8).tolist())) & writing
        # the policy acted with an image of the actor as learn() of the previous step left it: run one more step and compare
        # the stored action means with the pre-step actor on the observations the policy saw
        obs = a.ring.obs[a.ring.slot()].clone()
        ou_before = a.noise.x.clone()
        done_prev = a.ring.done[a.ring.slot(a.ring.k - 1)].bool()
        mu_before = fused.actor_forward(a.agent.actor, obs).view(-1).clone()
        a.run(1)
        torch.cuda.synchronize()
        stored = a.ring.act[a.ring.slot(a.ring.k - 1)]
        mu_used = stored - a.noise.x                                   # stored action = mu + the step's OU state
        assert (mu_used - mu_before).abs().max().item() <= 1e-6
        assert not torch.equal(fused.actor_forward(a.agent.actor, obs).view(-1), mu_before)   # learn() did move the actor meanwhile
    for lp in loops:
        lp.env.close()


@pytest.mark.parametrize("n", [1000, 77])
def test_ragged_env_counts_graphs_match_eager(gpu_device, n):
    """N that is no multiple of the 64-env tiles of the step kernel nor of the 128-env tiles of the policy kernel (and, for 77,
    below one tile): the graphs of the pipelined loop (two chains, ring cursor, two policy images) == eager steps, bit for
    bit, over a run that wraps the ring twice; nothing outside the N envs is touched (guard rows stay as they were)."""
    import torch
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    loops = []
    for graph_steps in (20, 0):
        env = TruckTrailerVecEnv(n)
        env.reset(seed=13)
        loops.append(DDPGRollout(env, batch_size=64, replay_slots=8, seed=13, graph_steps=graph_steps))
    a, b = loops
    assert a.pipeline and b.pipeline
    a.run(4 + 20 + 4 + 1 + 1)
    for _ in range(30):
        b.step()
    torch.cuda.synchronize()
    assert a.graphG is not None and a.graphM is not None and b.graph1 is None
    assert torch.equal(_loop_flat(a), _loop_flat(b))
    for name in ("obs", "act", "rew", "done"):
        assert torch.equal(getattr(a.ring, name), getattr(b.ring, name)), name
    assert torch.equal(a.noise.x, b.noise.x) and torch.equal(a.env.state, b.env.state)
    assert torch.isfinite(_loop_flat(a)).all() and a.ring.obs.shape[1] == n
    for lp in loops:
        lp.env.close()


def test_smallest_rings(gpu_device):
    """The pipelined order needs a ring of 3 + reserve = 5 slots (window: 2 steps); with 4 the loop keeps the serial order."""
    import torch
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    for slots, piped in ((5, True), (4, False)):
        loops = []
        for graph_steps in (4, 0):
            env = TruckTrailerVecEnv(300)
            env.reset(seed=3)
            loops.append(DDPGRollout(env, batch_size=32, replay_slots=slots, seed=3, graph_steps=graph_steps))
        a, b = loops
        assert a.pipeline == b.pipeline == piped
        a.run(17)
        for _ in range(17):
            b.step()
        torch.cuda.synchronize()
        assert torch.equal(_loop_flat(a), _loop_flat(b)) and torch.isfinite(_loop_flat(a)).all()
        assert int(a.learner.step_dev.item()) == int(b.learner.step_dev.item()) == (15 if piped else 16)
        for lp in loops:
            lp.env.close()
