"""Data-parallel learn(): world_size-2 gloo processes on CPU (the N>1 path of bench.py uses the same code with
the RCCL backend).  Each rank holds HALF of fixture F5's batch; after the two gradient all-reduces the
weights must (a) be identical on both ranks and (b) equal the reference's single-process learn() on the whole
batch (mean of the two half-batch mean-gradients == the full-batch mean-gradient)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT

F5 = os.path.join(GOLDEN, "f5_learner.npz")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from test_learner import _agent, _load_init
    z = np.load(F5, allow_pickle=False)
    torch.manual_seed(100 + rank)                 # different random init per rank: broadcast must fix that
    agent = _agent(torch.device("cpu"))
    if rank == 0:
        _load_init(agent, z)
    agent.enable_data_parallel()
    agent.update_network_parameters(tau=1)
    half = slice(rank * 128, (rank + 1) * 128)
    f = lambda k: torch.tensor(z[k][half], dtype=torch.float)
    batch = (f("batch_states"), f("batch_actions"), f("batch_rewards"), f("batch_states_"),
             torch.tensor(z["batch_dones"][half]))
    for _ in range(3):
        agent.learn_batch(*batch)
    flat = torch.cat([p.detach().reshape(-1) for net in agent._nets() for p in net.parameters()])
    torch.save(flat, os.path.join(out_dir, f"rank{rank}.pt"))
    if rank == 0:
        torch.save({n: getattr(agent, n).state_dict() for n in ("actor", "critic", "target_actor", "target_critic")},
                   os.path.join(out_dir, "state.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_reference_full_batch(tmp_path):
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    a = torch.load(tmp_path / "rank0.pt", weights_only=True)
    b = torch.load(tmp_path / "rank1.pt", weights_only=True)
    assert torch.equal(a, b), "ranks diverged"
    z = np.load(F5, allow_pickle=False)
    state = torch.load(tmp_path / "state.pt", weights_only=True)
    stride = int(z["sample_stride"])
    for name, sd in state.items():
        for k, v in sd.items():
            got = v.numpy()
            if k == "fc2.weight":
                got = got.reshape(-1)[::stride]
            ref = z[f"after3/{name}/{k}"]
            assert np.abs(got - ref).max() <= 4e-5 * max(1e-1, np.abs(ref).max()) + 1e-6, (name, k)


def _gpu_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)     # two ranks on ONE GPU: RCCL would refuse that
    dev = torch.device("cuda:0")
    from ddpg_trucktrailer_amd.agent import Agent
    from ddpg_trucktrailer_amd.fused_learn import FusedLearner
    from test_learner import _load_init
    z = np.load(F5, allow_pickle=False)
    torch.manual_seed(100 + rank)
    agent = Agent(alpha=1e-4, beta=1e-3, input_dims=(23,), tau=1e-3, n_actions=1, batch_size=128, device=dev, replay=False)
    if rank == 0:
        _load_init(agent, z)
    agent.enable_data_parallel()                 # broadcast of rank 0's weights
    agent.update_network_parameters(tau=1)
    fl = FusedLearner(agent, 128)
    fl.enable_data_parallel()
    half = slice(rank * 128, (rank + 1) * 128)
    f = lambda k: torch.tensor(z[k][half], dtype=torch.float, device=dev)
    d8 = torch.tensor(z["batch_dones"][half].astype(np.uint8), device=dev)
    for _ in range(3):
        fl.learn_batch(f("batch_states"), f("batch_actions"), f("batch_rewards"), f("batch_states_"), d8)
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for net in agent._nets() for p in net.parameters()]).cpu()
    torch.save(flat, os.path.join(out_dir, f"rank{rank}.pt"))
    if rank == 0:
        torch.save({n: {k: v.cpu() for k, v in getattr(agent, n).state_dict().items()}
                    for n in ("actor", "critic", "target_actor", "target_critic")}, os.path.join(out_dir, "state.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_fused_learner_matches_reference_full_batch(tmp_path, gpu_device):
    """The hand-fused learn() with the flat-gradient all-reduce: two ranks x half batches == the reference's
    full-batch learn() (fixture F5); ranks bit-identical."""
    port = _free_port()
    mp.start_processes(_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    a = torch.load(tmp_path / "rank0.pt", weights_only=True)
    b = torch.load(tmp_path / "rank1.pt", weights_only=True)
    assert torch.equal(a, b), "ranks diverged"
    z = np.load(F5, allow_pickle=False)
    state = torch.load(tmp_path / "state.pt", weights_only=True)
    stride = int(z["sample_stride"])
    for name, sd in state.items():
        for k, v in sd.items():
            got = v.numpy()
            if k == "fc2.weight":
                got = got.reshape(-1)[::stride]
            ref = z[f"after3/{name}/{k}"]
            assert np.abs(got - ref).max() <= 4e-5 * max(1e-1, np.abs(ref).max()) + 1e-6, (name, k)


def _loop_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)     # two ranks on ONE GPU: RCCL would refuse that
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    flats = []
    for graph_steps in (4, 0):             # graphs (three per step, collectives between them) / everything eager
        env = TruckTrailerVecEnv(1024, device="cuda:0")
        env.reset(seed=27 + rank)
        loop = DDPGRollout(env, batch_size=128, replay_slots=8, seed=27 + rank, world_size=world, graph_steps=graph_steps)
        loop.run(13)
        torch.cuda.synchronize()
        assert (loop.graph1 is not None) == (graph_steps > 0)
        flats.append(torch.cat([p.detach().reshape(-1) for net in loop.agent._nets() for p in net.parameters()]).cpu())
        env.close()
    torch.save(flats, os.path.join(out_dir, f"loop{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_loop_graphs_match_eager(tmp_path, gpu_device):
    """The N>1 vector loop: env shards differ per rank, gradients are averaged at the two optimizer sites, so the ranks'
    networks stay bit-identical; the segmented-graph launch path (DDPGRollout.run) equals the eager one."""
    port = _free_port()
    mp.start_processes(_loop_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    a = torch.load(tmp_path / "loop0.pt", weights_only=True)
    b = torch.load(tmp_path / "loop1.pt", weights_only=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), "ranks diverged"
    assert torch.equal(a[0], a[1]), "graph path differs from the eager path"
    assert torch.isfinite(a[0]).all()


def _nccl_loop_worker(rank, world, port, out_dir):
    """One rank per GPU on RCCL (backend "nccl"): the data-parallel vector loop, graph segments captured while the
    communicator is alive, ReduceOp.AVG on the flat gradient buffers."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert dist.get_backend() == "nccl"
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    flats = []
    # graphs + RCCL / everything eager + RCCL / (one rank only) the single-rank launch structure without collectives
    # the all-reduces as nodes of ONE graph per step (only where this node replays a captured collective correctly: every
    # rank asks the probe's answer from the environment, set by the test before the ranks start)
    variants = [dict(graph_steps=4, data_parallel=True), dict(graph_steps=0, data_parallel=True)]
    if os.environ.get("TT_TEST_GRAPH_COLLECTIVES") == "1":
        variants.append(dict(graph_steps=4, data_parallel=True, graph_collectives=True))
    if world == 1:
        variants.append(dict(graph_steps=4, data_parallel=False))
    steps = 13
    if os.environ.get("TT_TEST_G20") == "1":
        # the shape bench.py replays: ONE graph of 20 vector steps with 40 captured all-reduces in it, against eager steps
        variants = [dict(graph_steps=20, data_parallel=True, graph_collectives=True), dict(graph_steps=0, data_parallel=True)]
        steps = 4 + 20 + 4 + 1 + 20          # eager warm-up, then graphs of 20, 4, 1 and 20 again
    for kw in variants:
        env = TruckTrailerVecEnv(1024, device=dev)
        env.reset(seed=27 + rank)
        loop = DDPGRollout(env, batch_size=128, replay_slots=8, seed=27 + rank, world_size=world, **kw)
        assert loop.dp == kw["data_parallel"]
        loop.run(steps)
        torch.cuda.synchronize()
        if kw["graph_steps"]:
            single = bool(kw.get("graph_collectives"))
            assert loop.dp_single_graph == single
            assert loop.graph1 is not None and (loop.dp_graphs is not None) == (loop.dp and not single)
            assert (loop.graphG is not None) == (single or not loop.dp)
            assert (loop.graphM is not None) == (kw["graph_steps"] > 4 and (single or not loop.dp))
        flats.append(torch.cat([p.detach().reshape(-1) for net in loop.agent._nets() for p in net.parameters()]).cpu())
        env.close()
    torch.save(flats, os.path.join(out_dir, f"nccl{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_world_size_1_loop_graphs_match_eager(tmp_path, gpu_device):
    """The RCCL code path on ONE GPU: init_process_group("nccl", device_id=...), ReduceOp.AVG on the flat gradient
    buffers, the three graph segments captured with a live communicator.  Graph path == eager path bit for bit, and
    (AVG over one rank being the identity) == the single-rank loop whose Adam runs inside the weight-gradient launch."""
    port = _free_port()
    os.environ["TT_TEST_GRAPH_COLLECTIVES"] = "1"
    try:
        mp.start_processes(_nccl_loop_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True, start_method="spawn")
    finally:
        os.environ.pop("TT_TEST_GRAPH_COLLECTIVES", None)
    a = torch.load(tmp_path / "nccl0.pt", weights_only=True)
    assert torch.isfinite(a[0]).all()
    assert torch.equal(a[0], a[1]), "graph path differs from the eager path under RCCL"
    assert torch.equal(a[0], a[2]), "one graph per step with the all-reduces captured in it differs from the eager path"
    assert torch.equal(a[0], a[3]), "data-parallel structure at world size 1 differs from the single-rank loop"


@pytest.mark.gpu
def test_rccl_world_size_1_single_graph_of_20_steps(tmp_path, gpu_device):
    """The data-parallel launch structure bench.py uses where the probe says yes -- one hipGraph of 20 whole vector steps,
    the two gradient all-reduces of every step captured in it (40 RCCL nodes) -- on RCCL at world size 1: 49 steps through the
    graphs of 20, 4 and 1 == 49 eager steps, bit for bit."""
    port = _free_port()
    os.environ["TT_TEST_G20"] = "1"
    try:
        mp.start_processes(_nccl_loop_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True, start_method="spawn")
    finally:
        os.environ.pop("TT_TEST_G20", None)
    a = torch.load(tmp_path / "nccl0.pt", weights_only=True)
    assert torch.isfinite(a[0]).all()
    assert torch.equal(a[0], a[1]), "20-step graphs with captured all-reduces differ from eager data-parallel steps"


@pytest.mark.gpu
def test_graph_collective_probe_world_size_1(gpu_device):
    """dp_probe: a child process captures an AVG all-reduce into a hipGraph, replays it and checks the numbers; the
    caller (a process that has not touched a GPU: a fresh interpreter here) gets True.  One rank on the test box."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               TT_DP_PROBE_FORCE="1")
    code = ("import sys; sys.path.insert(0, %r); from ddpg_trucktrailer_amd.dp_probe import graph_collectives_ok; "
            "print('answer', graph_collectives_ok(200.0))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "answer True" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2])
def test_rccl_n_rank_loop(tmp_path, gpu_device, world):
    """One rank per device over RCCL/xGMI; skipped below `world` devices (the 1-GPU test box)."""
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    port = _free_port()
    from ddpg_trucktrailer_amd.dp_probe import graph_collectives_ok
    probe_env = dict(WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1")
    # (the probe's children are one per rank: ask for all of them from here, rank by rank, in threads)
    import threading
    answers = [False] * world
    def ask(r):
        import subprocess
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), MASTER_PORT=str(port + 7), **probe_env)
        code = ("import sys; sys.path.insert(0, %r); from ddpg_trucktrailer_amd.dp_probe import graph_collectives_ok; "
                "print('answer', graph_collectives_ok(200.0))" % ROOT)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=400)
        answers[r] = "answer True" in out.stdout
    threads = [threading.Thread(target=ask, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    single = all(answers)
    if single:
        os.environ["TT_TEST_GRAPH_COLLECTIVES"] = "1"
    try:
        mp.start_processes(_nccl_loop_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    finally:
        os.environ.pop("TT_TEST_GRAPH_COLLECTIVES", None)
    flats = [torch.load(tmp_path / f"nccl{r}.pt", weights_only=True) for r in range(world)]
    for r in range(1, world):
        assert all(torch.equal(a, b) for a, b in zip(flats[0], flats[r])), "ranks diverged"
    assert torch.equal(flats[0][0], flats[0][1]), "graph path differs from the eager path"
    if single:
        assert torch.equal(flats[0][0], flats[0][2]), "in-graph all-reduces differ from the eager path"
    assert torch.isfinite(flats[0][0]).all()


# ------------------------------------------------------------------------------------------------------------------
# Peer-to-peer gradient exchange (include/ttenv.h: tt_p2p_*): no collective on learn()'s chain; each rank's Adam launch reads
# the peers' flat gradient buffers through IPC-opened device memory behind a flag barrier.  Two processes on ONE GPU prove the
# protocol (handles, barrier, epochs, graphs); only its speed over xGMI needs a second GPU.
def _p2p_learner_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    from ddpg_trucktrailer_amd.agent import Agent
    from ddpg_trucktrailer_amd.fused_learn import FusedLearner
    from test_learner import _load_init
    z = np.load(F5, allow_pickle=False)
    torch.manual_seed(100 + rank)
    agent = Agent(alpha=1e-4, beta=1e-3, input_dims=(23,), tau=1e-3, n_actions=1, batch_size=128, device=dev, replay=False)
    if rank == 0:
        _load_init(agent, z)
    agent.enable_data_parallel()                 # broadcast of rank 0's weights (the control plane: gloo)
    agent.update_network_parameters(tau=1)
    fl = FusedLearner(agent, 128)
    fl.enable_p2p()
    assert fl.p2p is not None and fl.critic.flat_grad.data_ptr() != 0
    half = slice(rank * 128, (rank + 1) * 128)
    f = lambda k: torch.tensor(z[k][half], dtype=torch.float, device=dev)
    d8 = torch.tensor(z["batch_dones"][half].astype(np.uint8), device=dev)
    grads = []
    for _ in range(3):
        fl.learn_batch(f("batch_states"), f("batch_actions"), f("batch_rewards"), f("batch_states_"), d8)
        torch.cuda.synchronize()
        grads.append(torch.cat([fl.critic.flat_grad, fl.actor.flat_grad]).cpu())      # this rank's OWN half-batch gradients
    assert fl.p2p_gave_up() == 0
    flat = torch.cat([p.detach().reshape(-1) for net in agent._nets() for p in net.parameters()]).cpu()
    torch.save({"flat": flat, "grads": grads}, os.path.join(out_dir, f"p2p_rank{rank}.pt"))
    if rank == 0:
        torch.save({n: {k: v.cpu() for k, v in getattr(agent, n).state_dict().items()}
                    for n in ("actor", "critic", "target_actor", "target_critic")}, os.path.join(out_dir, "p2p_state.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_p2p_two_rank_fused_learner_matches_reference_full_batch(tmp_path, gpu_device):
    """Two ranks x half batches of fixture F5 with the peer-to-peer exchange == the reference's own full-batch learn() after three
    steps (the mean of the two half-batch mean-gradients is the full-batch mean-gradient; sites of DDPG/DDPG_agent.py:95-104);
    ranks bit-identical although their own gradients differ."""
    port = _free_port()
    mp.start_processes(_p2p_learner_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    a = torch.load(tmp_path / "p2p_rank0.pt", weights_only=True)
    b = torch.load(tmp_path / "p2p_rank1.pt", weights_only=True)
    assert torch.equal(a["flat"], b["flat"]), "ranks diverged"
    assert not torch.equal(a["grads"][0], b["grads"][0])          # each rank kept its own half-batch gradient in its own block
    z = np.load(F5, allow_pickle=False)
    state = torch.load(tmp_path / "p2p_state.pt", weights_only=True)
    stride = int(z["sample_stride"])
    for name, sd in state.items():
        for k, v in sd.items():
            got = v.numpy()
            if k == "fc2.weight":
                got = got.reshape(-1)[::stride]
            ref = z[f"after3/{name}/{k}"]
            assert np.abs(got - ref).max() <= 4e-5 * max(1e-1, np.abs(ref).max()) + 1e-6, (name, k)


def _p2p_loop_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", TT_P2P_TIMEOUT_S="20")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ddpg_trucktrailer_amd.rollout import DDPGRollout
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    flats = []
    steps = 4 + 20 + 4 + 1 + 4
    # p2p with graphs of 20 / 4 / 1 steps, p2p eager, the collective structure (gloo SUM / world) as three graph segments per step
    variants = [dict(graph_steps=20, dp_exchange="p2p"), dict(graph_steps=0, dp_exchange="p2p"), dict(graph_steps=4, dp_exchange="collective")]
    if world == 1:
        variants.append(dict(graph_steps=20, data_parallel=False))
    for kw in variants:
        env = TruckTrailerVecEnv(1024, device="cuda:0")
        env.reset(seed=27 + rank)
        kw.setdefault("data_parallel", True)
        loop = DDPGRollout(env, batch_size=128, replay_slots=8, seed=27 + rank, world_size=world, **kw)
        dist.barrier()                     # (the ranks start a variant together: the exchange's wait is bounded)
        loop.run(steps)
        torch.cuda.synchronize()
        if kw.get("dp_exchange") == "p2p":
            assert loop.dp and loop.dp_single_graph and loop.learner.p2p is not None and loop.learner.p2p_gave_up() == 0
            assert (loop.graphG is not None) == (kw["graph_steps"] > 0) and loop.dp_graphs is None
        assert loop.handover_gave_up == []
        flats.append(torch.cat([p.detach().reshape(-1) for net in loop.agent._nets() for p in net.parameters()]).cpu())
        env.close()
    torch.save(flats, os.path.join(out_dir, f"p2p_loop{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_p2p_loop_matches_the_collective_loop(tmp_path, gpu_device, world):
    """The N-env loop with the peer-to-peer exchange, `world` gloo-launched ranks on ONE GPU: graphs (20 / 4 / 1 whole steps, the
    exchange inside them) == eager steps bit for bit; every rank ends with the same bits; and the result equals the loop whose
    gradients go through the process group's all-reduce -- bitwise at these world sizes (a sum of two is order-free, / 2 exact;
    world 1: the identity), to rounding beyond.  World size 1 also equals the single-rank loop."""
    port = _free_port()
    mp.start_processes(_p2p_loop_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    flats = [torch.load(tmp_path / f"p2p_loop{r}.pt", weights_only=True) for r in range(world)]
    for r in range(1, world):
        assert all(torch.equal(x, y) for x, y in zip(flats[0], flats[r])), "ranks diverged"
    a = flats[0]
    assert torch.isfinite(a[0]).all()
    assert torch.equal(a[0], a[1]), "p2p graphs differ from p2p eager steps"
    assert torch.equal(a[0], a[2]), "p2p exchange differs from the collective (all-reduce) structure"
    if world == 1:
        assert torch.equal(a[0], a[3]), "data-parallel p2p structure at world size 1 differs from the single-rank loop"


_P2P_STALL = r"""
import os, sys, time
sys.path.insert(0, %r)
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from ddpg_trucktrailer_amd.rollout import DDPGRollout
from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
env = TruckTrailerVecEnv(1024, device="cuda:0")
env.reset(seed=27 + rank)
loop = DDPGRollout(env, batch_size=128, replay_slots=8, seed=27 + rank, world_size=world, graph_steps=4, data_parallel=True, dp_exchange="p2p")
loop.run(9)
torch.cuda.synchronize()
dist.barrier()
print("in step", rank, flush=True)
if rank == 1:
    time.sleep(8)              # rank 1 falls behind by more than the exchange's time limit
    print("rank 1 done sleeping", flush=True)
    os._exit(0)
try:
    loop.run(8)
    loop.state_dict()
    print("NOT NOTICED", flush=True)
except RuntimeError as exc:
    print("noticed:", exc, flush=True)
os._exit(0)
"""


@pytest.mark.gpu
def test_p2p_wait_is_bounded_and_a_give_up_is_an_error(gpu_device):
    """A rank whose peer stops taking part: its Adam launch waits for the peer's arrival word for the exchange's time limit (1 s
    here), marks a host-visible word and ends -- the GPU is never hung -- and the loop raises at its next look (the ranks'
    weights have diverged; there is no fallback for that)."""
    import subprocess
    import time
    port = _free_port()
    procs = []
    t0 = time.monotonic()
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TT_P2P_TIMEOUT_S="1",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", _P2P_STALL % ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    took = time.monotonic() - t0
    assert "in step 0" in outs[0][0] and "in step 1" in outs[1][0], outs
    assert "noticed:" in outs[0][0] and "gave up waiting for a peer's gradients" in outs[0][0], outs[0]
    assert "NOT NOTICED" not in outs[0][0]
    assert took < 200


@pytest.mark.gpu
def test_bench_two_ranks_p2p_rehearsal_on_one_gpu(gpu_device):
    """bench.py's own N > 1 path end to end with the peer-to-peer exchange: `python bench.py --gpus 2 --dp-mode p2p` starts its two
    ranks itself (TT_DIST_BACKEND=gloo lets them share the ONE GPU of the test box: the numbers mean nothing, the path does), the
    ranks open each other's exchange blocks, every vector step is one hipGraph, rank 0 prints ONE line that says which structure
    ran, and no wait was abandoned."""
    import json
    import subprocess
    env = dict(os.environ, TT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", TT_P2P_TIMEOUT_S="30")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dp-mode", "p2p", "--n-envs", "4096", "--steps", "40",
                        "--warmup", "8", "--repeats", "1", "--watchdog-seconds", "240"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["n_envs_total"] == 8192 and d["scaling"] == "weak"
    assert d["config"]["dp_mode"]["agreed_by_all_ranks"] == "p2p" and "no collective" in d["config"]["launch"]
    assert d["config"]["dp_mode"]["ranks_hold_the_same_weights"] is True
    assert d["value"] == pytest.approx(8192 * 40 / (d["ms_per_step"] * 1e-3 * 40), rel=1e-9)
    assert "allreduce_us" in d


@pytest.mark.gpu
def test_p2p_exchange_probe_and_auto_mode_on_one_gpu(gpu_device):
    """dp_probe.p2p_exchange_ok(): one throw-away child per rank opens the peers' exchange blocks, runs the exchange's launch for both
    sites eagerly and from a replayed hipGraph, and checks mean + identical bits across the ranks -- two ranks on the ONE GPU here;
    and `bench.py --gpus 2` with --dp-mode auto takes the exchange when every rank's probe said yes (TT_DIST_BACKEND=gloo: the ranks
    share the GPU), and says in its line who decided."""
    import json
    import subprocess
    port = _free_port()
    code = ("import sys; sys.path.insert(0, %r); from ddpg_trucktrailer_amd.dp_probe import p2p_exchange_ok; "
            "print('answer', p2p_exchange_ok(150.0))" % ROOT)
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              env=dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", TT_DP_PROBE_PORT=str(port),
                                       MASTER_PORT=str(port), TT_DP_PROBE_VERBOSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all("answer True" in o for o in outs), outs
    env = dict(os.environ, TT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", TT_P2P_TIMEOUT_S="30")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n-envs", "4096", "--steps", "20", "--warmup", "5",
                        "--repeats", "0", "--watchdog-seconds", "280"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["dp_mode"] == {"asked": "auto", "decided_by": "probe (p2p exchange)", "this_rank_vote": True,
                                                           "agreed_by_all_ranks": "p2p", "ranks_hold_the_same_weights": True}


@pytest.mark.gpu
def test_bench_auto_mode_falls_back_to_collectives_when_the_exchange_fails_its_validation(gpu_device):
    """--dp-mode auto trusts the probed peer-to-peer exchange only after it has carried the REAL loop for two graph lengths of steps on
    every rank; if one rank reports a failure there (forced here on rank 1), every rank drops the exchange, the loop is built again
    with all-reduces in three graph segments per step, and the line says what happened -- a first run on a new node still gives a
    number.  (TT_DIST_BACKEND=gloo: two ranks on the ONE GPU of the test box.)"""
    import json
    import subprocess
    env = dict(os.environ, TT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", TT_P2P_TIMEOUT_S="30", TT_BENCH_TEST_P2P_FAIL="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n-envs", "4096", "--steps", "20", "--warmup", "5",
                        "--repeats", "0", "--watchdog-seconds", "280"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "going on with RCCL all-reduces in three graph segments" in r.stderr
    d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    m = d["config"]["dp_mode"]
    assert d["n_gpus"] == 2 and m["agreed_by_all_ranks"] == "segments" and m["fell_back_from"] == "p2p", m
    assert m["ranks_hold_the_same_weights"] is True and "three hipGraph segments" in d["config"]["launch"]


def test_bench_refuses_fewer_gpus_than_ranks():
    """`python bench.py --gpus N` starts N ranks itself; with fewer than N GPUs visible it prints no line and exits 2."""
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    have = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 3), "--steps", "4", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr and "{" not in r.stdout


def test_graph_collective_probe_says_no_without_a_working_child(monkeypatch):
    """dp_probe.graph_collectives_ok(): any failure of the child -- here: no GPU, or (on a GPU box) a rendezvous that cannot
    complete because only one of two ranks exists -- within the time limit means "no", and the caller is left alone."""
    from ddpg_trucktrailer_amd.dp_probe import graph_collectives_ok
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    assert graph_collectives_ok(timeout=20.0) is False
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.delenv("TT_DP_PROBE_FORCE", raising=False)
    assert graph_collectives_ok(timeout=20.0) is False          # one rank: nothing to ask


_WATCHDOG_RANK = r"""
import os, sys, time
sys.path.insert(0, %r)
import bench
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
wd = bench.Watchdog(rank, world, limit=4.0)
wd.enter("init")
dist.init_process_group("gloo", rank=rank, world_size=world)
wd.enter("warm-up")          # TT_BENCH_STALL="1:warm-up": rank 1 never leaves this phase
dist.barrier()               # ... and rank 0 waits here for it
wd.enter("timed region")
dist.barrier()
wd.stop()
print("finished", rank, flush=True)
"""


@pytest.mark.parametrize("stall", ["1:warm-up", ""])
def test_bench_watchdog_names_the_stalled_rank_and_phase(stall):
    """bench.py's watchdog for world > 1 (two gloo ranks on the CPU): a rank that stalls in a phase ends with exit code 3 and a
    line that names rank and phase; the rank that waits for it in a collective ends the same way instead of hanging; with no
    stall both ranks finish and the watchdog stays silent."""
    import subprocess
    import time
    port = _free_port()
    procs = []
    t0 = time.monotonic()
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TT_BENCH_STALL=stall)
        procs.append(subprocess.Popen([sys.executable, "-c", _WATCHDOG_RANK % ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    took = time.monotonic() - t0
    if not stall:
        assert [p.returncode for p in procs] == [0, 0], outs
        assert all("watchdog" not in e for _, e in outs)
        return
    assert [p.returncode for p in procs] == [3, 3], outs
    assert "watchdog: rank 1 of 2 has been in phase 'warm-up'" in outs[1][1]
    assert "watchdog: rank 0 of 2 has been in phase 'warm-up'" in outs[0][1]      # blocked in the barrier behind rank 1
    assert all("finished" not in o for o, _ in outs)
    assert took < 60


def test_probe_port_is_agreed_through_the_launcher_store():
    """dp_probe.probe_port(): under torch.distributed.run every rank reads the port rank 0 found free from the agent's store
    (no GPU, no process group needed); the ranks' votes are reduced with MIN by dp_probe.agree()."""
    import subprocess
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "from ddpg_trucktrailer_amd import dp_probe\n"
            "import torch.distributed as dist\n"
            "p = dp_probe.probe_port()\n"
            "dist.init_process_group('gloo')\n"
            "r = dist.get_rank()\n"
            "print('port', p, 'agree', dp_probe.agree(r == 0), dp_probe.agree(True), flush=True)\n"
            "dist.destroy_process_group()\n" % ROOT)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "w.py")
        with open(path, "w") as f:
            f.write(code)
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                            "127.0.0.1", "--master-port", str(_free_port()), path], env=env, capture_output=True, text=True,
                           timeout=240)
    lines = [l.split() for l in r.stdout.splitlines() if l.startswith("port")]
    assert r.returncode == 0 and len(lines) == 2, r.stdout + r.stderr
    assert lines[0][1] == lines[1][1] and int(lines[0][1]) > 0          # one port for both ranks
    assert all(l[3] == "False" and l[4] == "True" for l in lines)       # one "no" vote is "no" everywhere
