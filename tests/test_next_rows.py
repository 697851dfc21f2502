"""grid evaluation, episode record, checkpoint/resume + best-model rule, expert ingestion."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_group


def test_best_model_tracker_follows_trainv2_rule():
    from ddpg_trucktrailer_amd.checkpoint import BestModelTracker
    tr = BestModelTracker(start_episode=0)
    best_seen = []
    for i in range(130):
        ok, avg, sr = tr.update(i, score=float(i), success=(i % 4 == 0), steps=10)
        best_seen.append(ok)
    assert not any(best_seen[:101])                    # `i > start_episode + 100` (trainv2.py:561)
    assert best_seen[101]                              # first eligible episode beats best_success_rate = 0
    st = tr.training_state(130)
    assert set(st) == {"episode_num", "score_history", "best_score", "best_success_rate", "success_history",
                       "total_steps", "step_history"} and st["total_steps"] == 1300 and len(st["score_history"]) == 130
    # equal success rate, higher average score -> best again; lower score -> not
    tr2 = BestModelTracker(start_episode=-200, best_score=5.0, best_success_rate=0.0)
    assert tr2.update(0, 10.0, False, 1)[0] and not tr2.update(1, -100.0, False, 1)[0]


def test_training_checkpoint_roundtrip_cpu(tmp_path):
    from ddpg_trucktrailer_amd.agent import Agent
    from ddpg_trucktrailer_amd.checkpoint import load_training_checkpoint, save_training_checkpoint
    from test_learner import _batch
    z = np.load(os.path.join(GOLDEN, "f5_learner.npz"), allow_pickle=False)
    dev = torch.device("cpu")
    mk = lambda: Agent(alpha=1e-4, beta=1e-3, input_dims=(23,), tau=1e-3, n_actions=1, batch_size=256, device=dev, max_size=10)
    a, b = mk(), mk()
    batch = _batch(z, dev)
    a.learn_batch(*batch)
    path = save_training_checkpoint(tmp_path / "ck.pt", a, training_state={"episode_num": 7})
    assert load_training_checkpoint(path, b)["episode_num"] == 7
    a.learn_batch(*batch); b.learn_batch(*batch)       # optimizer state came along: the next step is identical
    for n in ("actor", "critic", "target_actor", "target_critic"):
        for x, y in zip(getattr(a, n).state_dict().values(), getattr(b, n).state_dict().values()):
            assert torch.equal(x, y), n
    # the reference's own per-network files (networks.py:70-95) still work
    a.actor.checkpoint_dir = a.critic.checkpoint_dir = a.target_actor.checkpoint_dir = a.target_critic.checkpoint_dir = str(tmp_path)
    for net in a._nets():
        net.checkpoint_file = os.path.join(str(tmp_path), net.name + "_ddpg")
    a.save_models()
    assert sorted(os.listdir(tmp_path))[:4] == ["actor_ddpg", "ck.pt", "critic_ddpg", "target_actor_ddpg"]
    c = mk()
    for net in c._nets():
        net.checkpoint_file = os.path.join(str(tmp_path), net.name + "_ddpg")
    c.load_models()
    assert all(torch.equal(x, y) for x, y in zip(a.actor.state_dict().values(), c.actor.state_dict().values()))


def test_expert_transitions_bulk_load():
    from ddpg_trucktrailer_amd.expert import load_into_replay
    from ddpg_trucktrailer_amd.replay_buffer import ReplayBuffer
    t = load_group("f2_seeded.npz")["seed27"]
    obs = np.concatenate([t["obs0"][None], t["obs"]], 0)
    episode = [(obs[k], np.array([t["actions"][k] / np.radians(45)], np.float32), t["reward"][k], obs[k + 1], bool(t["done"][k]))
               for k in range(len(t["actions"]))]            # exp_gen.py:96-104 stores action / rad45
    bulk, loop = ReplayBuffer(1000, (23,), 1), ReplayBuffer(1000, (23,), 1)
    assert load_into_replay(bulk, [episode, episode[:10]]) == len(episode) + 10
    for tr in episode + episode[:10]:
        loop.store_transition(*tr)                           # the reference's loop (trainv2.py:462-465)
    assert bulk.mem_cntr == loop.mem_cntr
    for x, y in ((bulk.state_memory, loop.state_memory), (bulk.action_memory, loop.action_memory),
                 (bulk.reward_memory, loop.reward_memory), (bulk.new_state_memory, loop.new_state_memory),
                 (bulk.terminal_memory, loop.terminal_memory)):
        assert torch.equal(x, y)


@pytest.mark.gpu
def test_episode_recorder_reproduces_golden_episode_schema(gpu_device, tmp_path):
    """Record the reference's golden episode from lane 3 of a vector env; the saved file has the collector's schema
    and the recorded numbers match the fixture."""
    from ddpg_trucktrailer_amd.episode_replay import EpisodeRecorder, load_episode
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    t = load_group("f1_golden_episode.npz")["golden"]
    env = TruckTrailerVecEnv(8)
    env.set_pose(np.tile(t["start"], (8, 1)))
    env.set_state(np.tile(t["state0"], (8, 1)))
    rec = EpisodeRecorder(env, lanes=[3], save_dir=str(tmp_path))
    rec.begin()
    done_eps = []
    for a in t["actions"]:
        act = torch.full((8,), float(a), device="cuda")
        obs, rew, done, info = env.step(act, auto_reset=False, info=True)
        done_eps += rec.record(act, done, info)
    assert len(done_eps) == 1
    ep = done_eps[0]
    assert set(ep) == {"states", "actions", "episode_num", "env_data", "info"}          # episode_replay_collector.py:15-21
    assert len(ep["states"]) == 194 and len(ep["actions"]) == 193 and len(ep["info"]) == 193
    assert np.abs(np.array(ep["states"][1:]) - t["recorded_states"][1:]).max() <= 1e-5
    assert abs(sum(r["total_reward"] for r in ep["info"]) - 4792.9998) < 1e-3 and ep["info"][-1]["success"]
    files = os.listdir(tmp_path)
    assert files == ["episode_0_reward_4792.npz"]                                      # the reference's file name
    back = load_episode(os.path.join(tmp_path, files[0]))
    assert back["env_data"]["goaly"] == -30 and back["info"][-1]["final_success_bonus"] == 200.0
    env.close()


@pytest.mark.gpu
def test_env_checkpoint_resume_is_bitwise(gpu_device, tmp_path):
    from ddpg_trucktrailer_amd.agent import Agent
    from ddpg_trucktrailer_amd.checkpoint import load_training_checkpoint, save_training_checkpoint
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    n = 1000
    a, b = TruckTrailerVecEnv(n), TruckTrailerVecEnv(n)
    a.reset(seed=4)
    for k in range(30):
        a.step_random(9, auto_reset=True)
    agent = Agent(1e-4, 1e-3, (23,), 1e-3, 1, batch_size=64, device=gpu_device, replay=False)
    save_training_checkpoint(tmp_path / "ck.pt", agent, env=a)
    load_training_checkpoint(tmp_path / "ck.pt", agent, env=b)
    assert torch.equal(a.state, b.state)
    for k in range(40):                                # same actions, same in-kernel resets from here on
        oa, ra, da, _ = a.step_random(9, auto_reset=True)
        ob, rb, db, _ = b.step_random(9, auto_reset=True)
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
    a.close(); b.close()


@pytest.mark.gpu
def test_grid_evaluation_matches_single_env_episodes(gpu_device):
    """A small grid through the vector path equals the same episodes run one by one through the gym facade
    the way heatmap.py does (deterministic policy, per-trial yaw and L2)."""
    from ddpg_trucktrailer_amd.env import Truck_trailer_Env_2
    from ddpg_trucktrailer_amd.grid_eval import generate_heatmap_data
    from ddpg_trucktrailer_amd.networks import ActorNetwork
    torch.manual_seed(0)
    actor = ActorNetwork(1e-4, (23,), 400, 300, 1, name="actor", device=gpu_device)
    out = generate_heatmap_data(actor, grid_resolution=20.0, trials_per_cell=2, map_x_range=(-20, 21), map_y_range=(-10, 31), seed=66)
    reward_grid, success_grid, xs, ys, orient, endpoints, trajs = out
    assert reward_grid.shape == success_grid.shape == (len(ys), len(xs)) == (3, 3)
    assert len(orient) == len(endpoints) == 18 and len(trajs) == 9
    assert all(45 <= o["yaw_deg"] <= 120 for o in orient)
    rng = np.random.RandomState(66)
    yaw = rng.uniform(45, 120, 18); l2 = rng.uniform(5, 7, 18)
    env = Truck_trailer_Env_2()
    k = 0
    for iy, y0 in enumerate(ys):
        for ix, x0 in enumerate(xs):
            scores = []
            for trial in range(2):
                env.reset()
                env.goalx, env.goaly, env.goalyaw = 0.0, -30.0, np.deg2rad(90.0)
                env.startx, env.starty, env.startyaw = float(x0), float(y0), float(np.deg2rad(yaw[k]))
                env.L2 = float(l2[k])
                env.max_episode_steps = env.compute_max_steps()
                x1 = env.startx + env.L2 * np.cos(env.startyaw); y1 = env.starty + env.L2 * np.sin(env.startyaw)
                env.state = np.array([env.startyaw, env.startyaw, x1, y1, env.startx, env.starty], dtype=np.float32)
                obs = env.compute_observation(env.state, steering_angle=0.0)
                done, score = False, 0.0
                while not done:
                    with torch.no_grad():
                        mu = actor(torch.tensor(obs[None], device=gpu_device)).cpu().numpy()[0]
                    obs, r, done, info = env.step(np.clip(mu, -1, 1) * env.action_space.high)
                    score += r
                scores.append(score)
                assert abs(endpoints[k]["score"] - score) <= 1e-4 * max(1.0, abs(score))
                assert abs(endpoints[k]["end_x"] - env.state[4]) <= 1e-5
                k += 1
            assert abs(reward_grid[iy, ix] - np.mean(scores)) <= 1e-4 * max(1.0, abs(np.mean(scores)))
    env.close()


@pytest.mark.gpu
def test_expert_transitions_into_the_vector_loop_ring(gpu_device):
    """SURVEY 8f-4 on the replay the N-env loop actually samples: the reference re-inserts stored expert transitions one
    `agent.remember` at a time (trainv2.py:457-466) into the buffer it then samples uniformly (replay_buffer.py:23-34).
    Here they go into the TrajectoryRing's side buffer in bulk, and tt_ring_sample draws uniformly over ring + side
    transitions.  Checked against that remember loop: same stored tuples, every side draw an intact tuple, the expected
    share of side draws, and learn() runs on them from step 0 on."""
    import torch
    from ddpg_trucktrailer_amd import expert
    from ddpg_trucktrailer_amd.agent import Agent
    from ddpg_trucktrailer_amd.replay_buffer import TrajectoryRing
    rng = np.random.RandomState(0)
    episodes = []
    for e in range(5):                                   # exp_gen.py:77-110: (obs, action / rad45, reward, obs', done)
        ep = []
        for t in range(40 + e):
            ep.append((rng.uniform(-1, 1, 23).astype(np.float32), np.array([rng.uniform(-1, 1)], np.float32),
                       float(rng.normal()), rng.uniform(-1, 1, 23).astype(np.float32), t == 39 + e))
        episodes.append(ep)
    m = sum(len(ep) for ep in episodes)
    # the reference's way: one remember() per tuple
    agent = Agent(alpha=1e-4, beta=1e-3, input_dims=(23,), tau=1e-3, n_actions=1, batch_size=64, device=gpu_device,
                  max_size=1000)
    for ep in episodes:
        for s, a, r, s2, d in ep:
            agent.remember(s, a, r, s2, d)
    n, slots, B = 256, 6, 4096
    ring = TrajectoryRing(n, slots, 23, gpu_device)
    assert expert.load_into_ring(ring, episodes) == m == agent.memory.mem_cntr
    mem = agent.memory
    assert torch.equal(ring.side["obs"][:m], mem.state_memory[:m]) and torch.equal(ring.side["obs2"][:m], mem.new_state_memory[:m])
    assert torch.equal(ring.side["act"][:m], mem.action_memory[:m, 0]) and torch.equal(ring.side["rew"][:m], mem.reward_memory[:m])
    assert torch.equal(ring.side["done"][:m].bool(), mem.terminal_memory[:m])
    # empty ring: every draw is a side transition, intact and uniform
    s, a, r, s2, d, idx = ring.sample_fused(B, seed=1, return_index=True)
    assert (idx[:, 0] == -1).all()
    j = idx[:, 1].long()
    assert torch.equal(s, mem.state_memory[j]) and torch.equal(s2, mem.new_state_memory[j]) and torch.equal(r, mem.reward_memory[j])
    assert torch.equal(a, mem.action_memory[j]) and torch.equal(d, mem.terminal_memory[j])
    hist = torch.bincount(j // 21, minlength=10).float()[:10] / B
    assert (hist - 21 / m).abs().max() < 0.03
    # ring with 3 intact steps: side share = m / (m + 3 n); ring draws untouched by the side buffer
    for k in range(3):
        ring.act[ring.slot()] = 1000.0 + k
        ring.advance()
    s, a, r, s2, d, idx = ring.sample_fused(B, seed=2, return_index=True)
    side = idx[:, 0] == -1
    share = side.float().mean().item()
    assert abs(share - m / (m + 3 * n)) < 0.03, share
    assert (a[~side, 0] >= 1000).all() and (a[side, 0].abs() <= 1).all()
    assert torch.equal(s[side], mem.state_memory[idx[side, 1].long()])


@pytest.mark.gpu
def test_vector_loop_learns_from_expert_side_buffer(gpu_device):
    """A DDPGRollout whose ring holds expert transitions: graphs captured before the load are re-captured (the side
    count is a kernel argument), graphs == eager afterwards."""
    import torch
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    rng = np.random.RandomState(1)
    k = 500
    tup = (rng.uniform(-1, 1, (k, 23)).astype(np.float32), rng.uniform(-1, 1, k).astype(np.float32),
           rng.normal(size=k).astype(np.float32), rng.uniform(-1, 1, (k, 23)).astype(np.float32), rng.rand(k) < 0.05)
    flats = []
    for graph_steps in (4, 0):
        env = TruckTrailerVecEnv(512)
        env.reset(seed=6)
        loop = DDPGRollout(env, batch_size=128, replay_slots=8, seed=6, graph_steps=graph_steps)
        loop.run(8)
        loop.ring.load_side(*tup)
        loop.run(9)
        torch.cuda.synchronize()
        _, _, _, _, _, idx = loop.ring.sample_fused(4096, seed=11, return_index=True)
        assert 0.05 < (idx[:, 0] == -1).float().mean().item() < 0.25          # 500 / (500 + 7*512) = 0.12
        flats.append(torch.cat([p.detach().reshape(-1) for net in loop.agent._nets() for p in net.parameters()]).clone())
        env.close()
    assert torch.equal(flats[0], flats[1]) and torch.isfinite(flats[0]).all()
