#!/usr/bin/env python3
"""bench.py -- env-steps/s of the MI355X truck-trailer hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Started by `python -m torch.distributed.run ... bench.py --gpus N` the ranks read
RANK / LOCAL_RANK / WORLD_SIZE from the environment; started as plain `python bench.py --gpus N` the parent -- before it
touches any GPU -- starts those N ranks itself (torch.distributed.run as a child process), forwards their output and
exits with their code, and refuses (exit 2, no JSON line) when fewer than N GPUs are visible: `n_gpus` in the line is
always the number of ranks that ran.

A "step" is one vector step over all envs of the rank.  Workloads:
  ddpg  (default, BASELINE.json config 3): actor forward + OU noise for N envs -> env step kernel
        -> transitions into the device replay ring -> DDPG learn() (batch 256) per vector step;
  env   (config 2): random policy (Philox) -> env step kernel, auto-reset.
--variant simv1 runs either on the simv1 constants / termination mask / stateless reward / pose pool (config 5;
parity unpinned).  Inputs are synthetic (reference reset distribution, random-init 400x300 networks) and
resident in HBM before the timed region.  One JSON line on rank 0; see DESIGN.md "Measurement" for the fields."""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B_ALG = 313            # algorithmic bytes per env-step
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
PMC_FILES = ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")      # newest first
# the two gradient all-reduce sites of a data-parallel learn() (DDPG_agent.py:95-104): flat f32 buffers of these sizes
GRAD_NUMEL = {"critic": 132201, "actor": 131601}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)      # >= 200 warm vector steps, then >= 2000 timed
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--n-envs", type=int, default=65536, help="envs PER GPU (weak scaling)")
    ap.add_argument("--workload", choices=("ddpg", "env"), default="ddpg")
    ap.add_argument("--variant", choices=("simv2", "simv1"), default="simv2")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--updates-per-step", type=int, default=1, help="learn() calls per vector step")
    ap.add_argument("--replay-slots", type=int, default=64, help="ring length in vector steps (capacity = slots*N)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--repeats", type=int, default=5, help="extra hipEvent-timed repeats of the K steps after the timed region")
    ap.add_argument("--no-graph", action="store_true", help="launch everything eagerly (no hipGraphs)")
    ap.add_argument("--step-graph", type=int, default=20, help="ddpg workload: whole vector steps per captured hipGraph "
                    "(0: only learn() is captured)")
    ap.add_argument("--torch-learn", action="store_true", help="learn() through torch autograd instead of the fused HIP kernels")
    ap.add_argument("--serial", action="store_true", help="ddpg workload: the reference's strict order (policy, env step, then "
                    "learn() on a window that includes the new step) instead of the pipelined one")
    ap.add_argument("--graph-steps", type=int, default=50, help="env workload: vector steps per captured hipGraph")
    ap.add_argument("--dp-mode", choices=("auto", "graph", "segments", "p2p"), default="auto",
                    help="N > 1, ddpg workload: launch structure of a data-parallel vector step.  graph: one hipGraph per step "
                    "with the two RCCL all-reduces as its nodes; segments: three hipGraph segments with eager all-reduces "
                    "between them; auto: graph where every rank's probe (dp_probe) saw a captured all-reduce replay "
                    "correctly, else segments; p2p: NO collective on learn()'s chain -- every rank's Adam launch reads the peers' "
                    "gradient buffers itself (IPC-opened device memory, flag barrier; include/ttenv.h: tt_p2p_*), one hipGraph per "
                    "step of plain kernel launches (also with --gpus 1: the same launches against the rank's own block).  The "
                    "structure that ran is in config.launch / config.dp_mode")
    ap.add_argument("--settle-ms", type=float, default=None,
                    help="setup ends with this many milliseconds of untimed vector steps (default: 100 for the ddpg workload, env "
                         "TT_BENCH_SETTLE_MS; 0: none): after the idle of graph capture the GPU needs ~30 ms under load to reach its "
                         "steady clocks (tools/driver_form.py), more than a short --warmup gives it; the steps are counted in "
                         "config.setup_vector_steps")
    ap.add_argument("--watchdog-seconds", type=float, default=float(os.environ.get("TT_BENCH_WATCHDOG_S", "300")),
                    help="N > 1: a phase of the run (set-up, warm-up, timed region, ...) that has not completed this many "
                    "seconds after it began ends the rank with exit code 3 and a line that names rank and phase")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ N > 1 watchdog
class Watchdog:
    """A data-parallel run must not hang silently (a captured collective that the library mishandles stalls, it does not
    raise).  A daemon thread per rank: when the running phase is older than its limit it writes ONE line -- rank, phase,
    seconds -- to stderr and ends the process with os._exit(3) (no re-exec, no clean-up that could block on the GPU).
    TT_BENCH_STALL="rank:phase" makes that rank sleep in that phase (the test of this class)."""

    def __init__(self, rank, world, limit, out=None, period=0.25):
        self.rank, self.world, self.limit = rank, world, float(limit)
        self.out = out if out is not None else sys.stderr
        self.phase, self.since, self.limits = "start", time.monotonic(), {}
        self._stop = threading.Event()
        self._period = period
        self._thread = threading.Thread(target=self._watch, name="bench-watchdog", daemon=True)
        self._thread.start()

    def enter(self, phase, limit=None):
        if limit is not None:
            self.limits[phase] = float(limit)
        self.phase, self.since = phase, time.monotonic()
        stall = os.environ.get("TT_BENCH_STALL", "")
        if stall == f"{self.rank}:{phase}":
            while True:
                time.sleep(3600)

    def stop(self):
        self._stop.set()

    def _watch(self):
        while not self._stop.wait(self._period):
            phase, since = self.phase, self.since
            age = time.monotonic() - since
            if age > self.limits.get(phase, self.limit):
                print(f"bench.py watchdog: rank {self.rank} of {self.world} has been in phase '{phase}' for {age:.0f} s "
                      f"(limit {self.limits.get(phase, self.limit):.0f} s); giving up with exit code 3", file=self.out, flush=True)
                os._exit(3)


class _NoWatchdog:
    def enter(self, phase, limit=None):
        pass

    def stop(self):
        pass


def weights_agree(loop, dev, world):
    """(every rank holds the same finite weights, the ranks' checksums): a collective over the default process group."""
    import torch
    import torch.distributed as dist
    flat = torch.cat([p.detach().reshape(-1) for net in loop.agent._nets() for p in net.parameters()])
    h = torch.stack([flat.view(torch.int32).to(torch.int64).sum(), (flat.view(torch.int32).to(torch.int64) * torch.arange(
        1, flat.numel() + 1, device=dev, dtype=torch.int64)).sum()]).cpu()
    if dist.get_backend() == "nccl":      # (RCCL gathers device tensors only: every tensor of the call on this rank's GPU)
        h = h.to(dev)
    hs = [torch.zeros_like(h) for _ in range(world)]
    dist.all_gather(hs, h)
    same = all(torch.equal(hs[0].cpu(), x.cpu()) for x in hs[1:]) and bool(torch.isfinite(flat).all())
    return same, hs


def allreduce_us(dev, world, reps=50):
    """Wire + launch time of the two gradient all-reduces alone: `reps` eager AVG all-reduces of each site's flat f32 buffer,
    event-timed back to back after 5 untimed ones; max over ranks.  Lets a scaling curve be split into time on the wire and
    time on the chain (efficiency ~ t_1 / (t_1 + structure + 2 t_allreduce))."""
    import torch
    import torch.distributed as dist
    out = {}
    avg = dist.get_backend() == "nccl"
    for site, numel in GRAD_NUMEL.items():
        buf = torch.zeros(numel, dtype=torch.float32, device=dev)
        one = (lambda: dist.all_reduce(buf, op=dist.ReduceOp.AVG)) if avg else (lambda: dist.all_reduce(buf))
        for _ in range(5):
            one()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            one()
        b.record()
        torch.cuda.synchronize()
        t = torch.tensor([a.elapsed_time(b) * 1e3 / reps], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out[f"{site}_{numel * 4 // 1000}KB"] = float(t.item())
    out["note"] = f"{reps} eager all-reduces per site back to back on backend {dist.get_backend()}, max over the {world} ranks"
    return out


# ------------------------------------------------------------------------------------------------ N > 1 launcher
def spawn_ranks(args):
    """Parent of a plain `python bench.py --gpus N`: starts the N ranks and relays them.  Touches no GPU itself
    (torch.cuda.device_count() does not initialise one on this image)."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("TT_DIST_BACKEND", "nccl") == "nccl":      # (gloo: a rehearsal, ranks share GPUs)
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible; refusing to print a line for fewer ranks",
              file=sys.stderr, flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------ CPU baselines
def pmc_traffic(n):
    """HBM bytes per k_step launch from the committed PMC passes (profiles/r0X_pmc_traffic.json: rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    bench.py cannot collect counters itself; (None, None) when no summary is present."""
    for name in PMC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return float(json.load(f)["traffic_B_per_env_step"]) * n, "profiles/" + name
        except Exception:
            continue
    return None, None


def _twin_worker(seconds, seed, q):
    """One process of the all-cores baseline: the structure-faithful twin, random policy, reset on done."""
    import numpy as np
    from oracle.simv2_twin import Simv2Twin
    env = Simv2Twin()
    rng = np.random.RandomState(seed)
    env.reset(seed=seed)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(200):
            if env.step(np.array([rng.uniform(-1, 1) * np.pi / 4], np.float32))[2]:
                env.reset()
        n += 200
    q.put((n, time.perf_counter() - t0))


def host_cpu():
    """(model name, physical cores this process may run on, logical CPUs it may run on)."""
    allowed = sorted(os.sched_getaffinity(0))
    model, cores, cur = "unknown", set(), {}
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if ":" not in line:
                    if cur.get("processor") in allowed:
                        cores.add((cur.get("physical id", 0), cur.get("core id", cur.get("processor"))))
                    cur = {}
                    continue
                k, v = (x.strip() for x in line.split(":", 1))
                if k == "model name":
                    model = v
                elif k in ("processor", "physical id", "core id"):
                    cur[k] = int(v)
        if cur.get("processor") in allowed:
            cores.add((cur.get("physical id", 0), cur.get("core id", cur.get("processor"))))
    except OSError:
        pass
    return model, (len(cores) or len(allowed)), len(allowed)


def cpu_twin_all_cores(seconds, procs):
    """SURVEY 8d CPU baseline (2): one twin process per physical core (multiprocessing, spawn: fresh interpreters)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_twin_worker, args=(seconds, 1000 + i, q)) for i in range(procs)]
    t0 = time.perf_counter()
    for p in ps:
        p.start()
    res = [q.get() for _ in ps]
    for p in ps:
        p.join()
    wall = time.perf_counter() - t0
    steps = sum(r[0] for r in res)
    rate = sum(r[0] / r[1] for r in res)
    return {"value": rate, "unit": "env-steps/s", "cores": procs,
            "sample": f"{steps} steps of oracle/simv2_twin.py in {procs} processes (one per physical core), "
                      f"{seconds:.0f} s each, {wall:.1f} s wall incl. start-up"}


def cpu_full_loop(seconds=6.0, batch=256):
    """The reference's whole training loop shape on the host: twin env step + choose_action + remember + learn()
    (torch CPU, batch 256 as in BASELINE config 3), one process."""
    import numpy as np
    import torch
    from ddpg_trucktrailer_amd.agent import Agent
    from oracle.simv2_twin import Simv2Twin
    env = Simv2Twin()
    threads_before = torch.get_num_threads()
    torch.set_num_threads(min(8, threads_before))     # more threads only slow these 256-row kernels down
    agent = Agent(alpha=1e-4, beta=1e-3, input_dims=(23,), tau=1e-3, n_actions=1, batch_size=batch, device="cpu",
                  max_size=100000)
    obs, _ = env.reset(seed=0)
    high = np.float32(np.pi / 4)
    n, t0 = 0, None
    while True:
        a = agent.choose_action(obs)
        obs2, r, done, _ = env.step(np.clip(a, -1, 1) * high)
        agent.remember(obs, a, r, obs2, done)
        agent.learn()
        obs = env.reset()[0] if done else obs2
        n += 1
        if t0 is None and n == batch + 20:      # learn() is live from step `batch` on
            t0, n0 = time.perf_counter(), n
        if t0 is not None and time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    used = torch.get_num_threads()
    torch.set_num_threads(threads_before)
    return {"value": (n - n0) / dt, "unit": "env-steps/s", "cores": used,
            "sample": f"{n - n0} iterations of twin env.step + choose_action + remember + learn(batch {batch}) on torch CPU, {dt:.1f} s"}


def cpu_baseline(seconds):
    """Oracle timed on the host cores (rank 0, N=1 only, BEFORE this process touches the GPU): the structure-faithful
    Python/scipy twin of the reference's step loop -- on ONE core (the reference is single-threaded; this is `value`)
    and in one process per physical core; beside them the whole loop on torch-CPU and the plain-C port on all cores."""
    import numpy as np
    from oracle import c_oracle
    from oracle.simv2_twin import Simv2Twin
    model, phys, logical = host_cpu()
    env = Simv2Twin()
    rng = np.random.RandomState(0)
    env.reset(seed=0)
    for _ in range(200):  # warm-up
        if env.step(np.array([rng.uniform(-1, 1) * np.pi / 4], np.float32))[2]:
            env.reset()
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(500):
            if env.step(np.array([rng.uniform(-1, 1) * np.pi / 4], np.float32))[2]:
                env.reset()
        n += 500
    dt = time.perf_counter() - t0
    c_oracle.rollout_random(256, 50, seed=1, nthreads=logical)
    t1 = time.perf_counter()
    cn, _ = c_oracle.rollout_random(4096, 400, seed=2, nthreads=logical)
    cdt = time.perf_counter() - t1
    out = {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port", "cpu_model": model,
           "physical_cores": phys, "logical_cpus": logical,
           "sample": f"{n} steps of oracle/simv2_twin.py (numpy + scipy solve_ivp RK45 + reward object per step), "
                     f"uniform-random steering, reset on done, {dt:.1f} s"}
    try:
        out["all_cores"] = cpu_twin_all_cores(max(4.0, seconds * 0.6), phys)
    except Exception as exc:      # a box that refuses child processes still gets the one-core line
        out["all_cores"] = {"error": repr(exc)}
    out["full_loop"] = cpu_full_loop()
    out["c_port"] = {"value": cn / cdt, "unit": "env-steps/s", "cores": logical,
                     "sample": f"{cn} steps of oracle/tt_oracle.c (fixed DP5, OpenMP), {cdt:.2f} s"}
    return out


# ------------------------------------------------------------------------------------------------ the bench proper
def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world                      # the line reports the ranks that actually run

    wd = Watchdog(rank, world, args.watchdog_seconds) if world > 1 else _NoWatchdog()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_seconds)   # before any GPU call: its worker processes start from a GPU-free parent

    nccl = os.environ.get("TT_DIST_BACKEND", "nccl") == "nccl"
    dp_vote, dp_asked = False, "n/a"
    p2p = args.dp_mode == "p2p" and args.workload == "ddpg"
    p2p_vote = p2p
    if p2p:
        dp_vote, dp_asked = True, "--dp-mode p2p"
    elif world > 1 and args.workload == "ddpg":
        # N > 1: which structure a data-parallel vector step gets is asked in throw-away child processes BEFORE this rank touches
        # its GPU (ddpg-trucktrailer_amd/dp_probe.py), and the answers here are this rank's VOTES; the ranks agree below, once
        # the process group exists.  auto: (1) the peer-to-peer exchange if every rank's probe ran it correctly across the real
        # ranks (no collective on learn()'s chain at all), else (2) one hipGraph per step with the two RCCL all-reduces as its
        # nodes if this node replays a captured all-reduce correctly, else (3) three graph segments with eager all-reduces.
        limit = min(240.0, max(30.0, args.watchdog_seconds - 60.0))
        if args.dp_mode == "auto" and os.environ.get("TT_DP_AUTO_P2P", "1") == "1":
            from ddpg_trucktrailer_amd.dp_probe import p2p_exchange_ok, probe_port
            wd.enter("dp-probe (p2p exchange)", limit + 60.0)
            p2p_vote = p2p_exchange_ok(timeout=min(limit, 180.0), port=probe_port(tag="p2p"))
            dp_asked = "probe (p2p exchange)"
        if nccl and not p2p_vote:         # (a rank whose exchange probe passed does not ask about captured collectives as well: if the
            #                               others' did not, the run falls back to segments -- the slowest structure, the safest)
            if args.dp_mode == "auto" and "TT_DP_GRAPH_COLLECTIVES" in os.environ:
                dp_vote, dp_asked = os.environ["TT_DP_GRAPH_COLLECTIVES"] == "1", "TT_DP_GRAPH_COLLECTIVES"
            elif args.dp_mode == "auto":
                from ddpg_trucktrailer_amd.dp_probe import graph_collectives_ok, probe_port
                wd.enter("dp-probe", limit + 60.0)
                dp_vote, dp_asked = graph_collectives_ok(timeout=limit, port=probe_port()), "probe"
            else:
                dp_vote, dp_asked = args.dp_mode == "graph", "--dp-mode " + args.dp_mode

    wd.enter("init")
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path is a HIP kernel)"
    if world > 1 and os.environ.get("TT_DIST_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < world:
        print(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s)", file=sys.stderr, flush=True)
        sys.exit(2)
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") over xGMI is the product path; TT_DIST_BACKEND=gloo lets the N>1 code path be rehearsed with
        # several ranks on ONE GPU (RCCL refuses two ranks per device)
        backend = os.environ.get("TT_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        from ddpg_trucktrailer_amd.dp_probe import agree
        if not p2p and agree(p2p_vote, dev):   # every rank's probe ran the exchange correctly: it is the structure of this run
            p2p = True
        dp_graph = agree(dp_vote, dev)         # every rank builds the same launch structure
    else:
        dp_graph = False

    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv

    n = args.n_envs
    variant = 1 if args.variant == "simv1" else 0
    env = TruckTrailerVecEnv(n, device=dev, variant=variant)
    if variant == 1:
        from ddpg_trucktrailer_amd import simv1_reset
        env.set_reset_pool(simv1_reset.generate_pose_pool(1024, seed=27 + rank))     # simv1.py:255-282 (host logic)
    env.reset(seed=27 + rank)
    vname = ("simv1 (L1 5.74, L2 10.192, 300-step cap, mask jackknife|out-of-map|max-steps|goal, stateless reward, "
             "Dubins-feasible pose pool; PARITY UNPINNED)") if variant else "simv2"

    graph_k = 1
    ddpg_loop = None
    if args.workload == "env":
        # one vector step = one launch of the fused kernel (action drawn in-kernel).  Launch-bound in eager
        # Python at this N, so G steps are captured into one hipGraph and a "step" replays 1/G of it.
        graph_k = max(1, min(args.graph_steps, args.steps))
        while args.steps % graph_k or args.warmup % graph_k:
            graph_k -= 1
        graph = None
        if graph_k > 1:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                env.step_random(123 + rank, auto_reset=True)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(graph_k):
                    env.step_random(123 + rank, auto_reset=True)

        def run(k_steps):
            for _ in range(k_steps // graph_k):
                if graph is not None:
                    graph.replay()
                else:
                    env.step_random(123 + rank, auto_reset=True)
        workload = (f"{vname} N={n}/GPU, random policy U(-1,1)*pi/4 drawn in-kernel, auto-reset "
                    f"(BASELINE config {5 if variant else 2} at bench size), hipGraph of {graph_k} steps")
        extra = {"graph_steps": graph_k}
    else:
        from ddpg_trucktrailer_amd.rollout import DDPGRollout

        def build_ddpg(p2p, dp_graph):
            """The loop with one gradient-exchange structure, prepared and launched once, + what the line says about it."""
            loop = DDPGRollout(env, batch_size=args.batch, replay_slots=args.replay_slots, seed=27 + rank,
                               world_size=world, use_graph=not args.no_graph, fused_learn=not args.torch_learn,
                               graph_steps=args.step_graph, updates_per_step=args.updates_per_step,
                               pipeline=False if args.serial else None, graph_collectives=dp_graph,
                               dp_exchange="p2p" if p2p else None, data_parallel=True if p2p else None)
            wd.enter("prepare (eager steps + graph capture)")
            loop.prepare()       # one-off work (4 untimed vector steps + graph capture) before warm-up and the timed region
            wd.enter("first graph launches")
            loop.first_launches()   # ... and the first launch of every captured graph (it uploads the graph)
            torch.cuda.synchronize()
            workload = (f"{vname} N={n}/GPU + full DDPG learn() x{args.updates_per_step} per vector step (actor/critic 400x300, "
                        f"batch {args.batch}, OU noise, replay ring {args.replay_slots}xN) (BASELINE config {5 if variant else 3})")
            if loop.graph_steps:
                launch = (f"every vector step a hipGraph replay: graphs of {loop.graph_steps}, 4 and 1 whole steps "
                          f"serve every ring position (device cursor)" + ((" -- no collective: each rank's two Adam launches read the peers' gradient "
                                                                    "buffers themselves (IPC-opened device memory, flag barrier)" if p2p else
                                                                    " -- the two RCCL gradient all-reduces of a step are nodes of its graph")
                                                                   if loop.dp else "") if not (loop.dp and not loop.dp_single_graph) else
                          "three hipGraph segments per step with the two RCCL gradient all-reduces between them")
            else:
                launch = "eager, learn() as a hipGraph" if loop.use_graph else "eager"
            order = ("pipelined: learn() of vector step t (batch from the steps up to t-2) runs beside the policy + env launches of "
                     "steps t-1 and t on a second chain of the graph; the policy acts with the weights learn() of step t-1 left"
                     if loop.pipeline else "serial: policy, env step, then learn() on a window that includes the new step")
            extra = {"batch": args.batch, "updates_per_step": args.updates_per_step,
                     "replay_capacity": args.replay_slots * n, "launch": launch, "order": order,
                     "dp_mode": (None if not loop.dp else
                                 {"asked": args.dp_mode, "decided_by": dp_asked, "this_rank_vote": bool(p2p_vote if p2p else dp_vote),
                                  "agreed_by_all_ranks": "p2p" if p2p else ("graph" if loop.dp_single_graph else "segments")}),
                     "env_steps_per_update": n / args.updates_per_step, "setup_vector_steps": loop.vector_steps,
                     "note": ("throughput of the configuration BASELINE.json names; how the same loop trains at this and at other "
                              "update ratios (--updates-per-step): profiles/r03_training_behaviour.md")}
            return loop, workload, extra

        loop, workload, extra = build_ddpg(p2p, dp_graph)
        if p2p and world > 1 and args.dp_mode == "auto":
            # The peer-to-peer exchange was CHOSEN by a probe (auto), never asked for: before anything is timed it has to carry the real
            # loop on this node -- two graph lengths of steps, then no wait abandoned and the same weights on every rank.  If not, the
            # run goes on with the RCCL all-reduces in three graph segments per step (the structure that needs nothing from the node
            # but RCCL) and the line says so.  Every rank executes the same collectives here whatever happened to it locally.
            wd.enter("p2p validation (steps through the exchange before it is trusted)")
            ok_here, why = True, ""
            try:
                loop.run(2 * max(1, loop.graph_steps))
                torch.cuda.synchronize()
                if loop.learner.p2p_gave_up():
                    ok_here, why = False, f"an exchange wait was abandoned at learn step {loop.learner.p2p_gave_up()}"
            except RuntimeError as exc:
                ok_here, why = False, str(exc)[:300]
            if os.environ.get("TT_BENCH_TEST_P2P_FAIL") == str(rank):      # (tests/test_distributed.py: the fallback below must work)
                ok_here, why = False, "TT_BENCH_TEST_P2P_FAIL"
            from ddpg_trucktrailer_amd.dp_probe import agree
            ok_all = agree(ok_here, dev)
            same = weights_agree(loop, dev, world)[0] if ok_all else False       # (a collective: only when every rank gets here)
            if not (ok_all and same):
                print(f"bench.py: rank {rank}: the peer-to-peer exchange did not carry the loop on this node ({why or 'another rank' if not ok_all else 'the ranks differ'}); "
                      "going on with RCCL all-reduces in three graph segments per step", file=sys.stderr, flush=True)
                wd.enter("p2p fallback (tear down, build the RCCL loop)")
                torch.cuda.synchronize()
                dist.barrier()              # every rank has stopped launching into the exchange
                del loop                    # (its exchange blocks stay allocated until the process ends -- 2 MB; nothing a peer has
                import gc                   # mapped is freed under it)
                gc.collect()
                p2p = False
                loop, workload, extra = build_ddpg(False, False)
                extra["dp_mode"]["fell_back_from"] = "p2p"
                extra["dp_mode"]["fell_back_why"] = why or ("another rank's exchange failed" if not ok_all else "the ranks' weights differed")
        ddpg_loop = loop
        run = lambda k_steps: ddpg_loop.run(k_steps)      # every step a hipGraph replay (G-step graphs where aligned, single-step graphs elsewhere)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    ar_us = None
    if world > 1:
        wd.enter("all-reduce timing")
        ar_us = allreduce_us(dev, world)
    # Setup ends under load.  The capture above leaves the GPU idle for hundreds of milliseconds and its clocks low; they need
    # ~30 ms of work to come back (tools/driver_form.py: the first 2 ms region after a 10 ms idle runs 5-9 % slow), which a
    # warm-up of a few steps does not provide.  Untimed vector steps of the same loop, counted in config.setup_vector_steps;
    # every rank runs the same number (the data-parallel loop holds collectives).
    # What the driver's form gives WITHOUT the settle phase below (rounds 1-2 measured exactly this): W warm-up steps right after
    # the idle of graph capture, then K steps between synchronizes -- reported beside the headline as timing.cold_ms_per_step.
    cold_ms = None
    if ddpg_loop is not None:
        wd.enter("cold region (no settle phase)")
        run(args.warmup)
        sync_all()
        t0 = time.perf_counter()
        run(args.steps)
        sync_all()
        cold = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(cold, op=dist.ReduceOp.MAX)
        cold_ms = float(cold.item()) / args.steps * 1e3
    settle_steps = 0
    if args.settle_ms is None:     # (the env workload's 0.2 ms regions gain nothing: 9.2 -> 8.9 us per step on the GPU's clock, but
        args.settle_ms = float(os.environ.get("TT_BENCH_SETTLE_MS", "100")) if ddpg_loop is not None else 0.0   # +15 us of host time)
    if args.settle_ms > 0:
        wd.enter("settle (steady clocks)")
        unit = graph_k if ddpg_loop is None else max(1, ddpg_loop.graph_steps)
        t0 = time.perf_counter()
        run(unit)
        sync_all()
        per_unit = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(per_unit, op=dist.ReduceOp.MIN)
        reps = min(2000, max(1, int(args.settle_ms * 1e-3 / max(float(per_unit.item()), 1e-6))))
        for _ in range(reps):       # a synchronize per unit: hundreds of queued replays leave the runtime resources to reclaim,
            run(unit)               # and it does that in the next synchronize -- the timed region's (0.3 ms; seen with the env workload)
            sync_all()
        settle_steps = unit * (reps + 1)
    extra["setup_vector_steps"] = ddpg_loop.vector_steps if ddpg_loop is not None else settle_steps
    extra["setup_settle"] = {"ms_asked": args.settle_ms, "vector_steps": settle_steps,
                             "why": "untimed steps of the same loop right before the warm-up, so that the timed region starts at the "
                                    "GPU's steady clocks (~30 ms under load after the idle of graph capture); --settle-ms 0 for none"}
    wd.enter("warm-up")
    run(args.warmup)
    sync_all()
    # The policy launch of a captured step waits for its image in device memory (include/ttenv.h: image hand-over), bounded:
    # a launch that left by its time limit -- it never has on this stack (tests at this graph length, 200 k-step soaks), but which
    # hardware queue the runtime gives a graph's chain is not a promise -- means the chains did not run beside each other.
    # DDPGRollout.run() notices that itself (a host-visible mirror of the give-up word, looked at after every replay), captures
    # the steps again with graph edges (0.095 instead of 0.089 ms per step) and goes on; a second time it raises.  Seen BEFORE the
    # timed region it costs the run nothing but the hand-over, and the line says so.  Inside the timed region it fails the run.
    handover_fallback = False
    if ddpg_loop is not None and ddpg_loop.ring_mode and ddpg_loop.graph_steps and ddpg_loop.policy_edge() == "flag":
        if os.environ.get("TT_BENCH_TEST_GAVE_UP") == "1":      # (tests/test_gpu_bench_line.py: the fallback below must work)
            ddpg_loop.ring.mark_gave_up_for_test(ddpg_loop.ring.k - 1)
        if ddpg_loop._check_handover(exact=True) or ddpg_loop.handover_gave_up:
            print("bench.py: a launch gave up waiting for the other chain of its step during setup; the steps are captured again with graph "
                  "edges between the chains (as TT_POLICY_EDGE=graph does)", file=sys.stderr, flush=True)
            wd.enter("prepare again (graph edge)")
            handover_fallback = True
            ddpg_loop.prepare()
            ddpg_loop.first_launches()
            run(max(args.warmup, ddpg_loop.graph_steps))
            sync_all()
            extra["setup_vector_steps"] = ddpg_loop.vector_steps
    gave_up_in_setup = len(ddpg_loop.handover_gave_up) if ddpg_loop is not None else 0
    wd.enter("timed region")
    captured = graph_k > 1 or (ddpg_loop is not None and ddpg_loop.graph_steps > 0)
    if not captured:
        env.profile(args.steps)      # per-dispatch HIP events on the step kernel, inside the timed region
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    run(args.steps)
    e1.record(stream)
    sync_all()
    elapsed = time.perf_counter() - t0
    event_ms = e0.elapsed_time(e1)
    wd.enter("after the timed region (repeats, per-dispatch kernel timing)")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # hipEvent-timed repeats of the same K steps, median and spread (the line's value stays the contract's
    # wall-clock region above)
    reps = []
    for _ in range(max(0, args.repeats)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        run(args.steps)
        b.record(stream)
        sync_all()
        reps.append(a.elapsed_time(b) / args.steps)
    if captured:
        # a captured launch cannot carry its own events: time the same kernel in the same loop on the same state right
        # after, eagerly, with the per-dispatch events (the timed region above is untouched by this)
        m = min(args.steps, 2000)
        env.profile(m)
        for _ in range(m):
            if ddpg_loop is not None:
                ddpg_loop.step()
            else:
                env.step_random(123 + rank, auto_reset=True)
    kern_total_ms, kern_launches = env.profile_read()
    env.profile(0)
    kern_ms = kern_total_ms / max(1, kern_launches)
    total_env_steps = n * world * args.steps
    value = total_env_steps / elapsed
    achieved = (B_ALG * n) / (kern_ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(n)
    out = {
        "metric": "env-steps/s", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": dict({"workload": workload, "n_envs_per_gpu": n, "n_envs_total": n * world}, **extra),
        "timing": {"wall_ms_per_step": elapsed / args.steps * 1e3, "event_ms_per_step": event_ms / args.steps,
                   "cold_ms_per_step": cold_ms,      # the same K steps after W warm-up steps BEFORE the settle phase (config.setup_settle)
                   "repeats": len(reps), "repeat_event_ms_per_step": reps,
                   "median_ms_per_step": (sorted(reps)[len(reps) // 2] if reps else None),
                   "spread_ms_per_step": ((max(reps) - min(reps)) if reps else None)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "k_step", "kernel_ms": kern_ms, "alg_bytes_per_env_step": B_ALG,
                     "kernel_env_steps_per_s": n / (kern_ms * 1e-3),
                     "timed": ("per-dispatch HIP events on eager launches of the same loop right after the timed region"
                               if captured else "per-dispatch HIP events inside the timed region")},
    }
    if ddpg_loop is not None and ddpg_loop.fused_act:
        # the loop's largest kernel is the N-env policy forward (csrc/ttnet_split.hip): MFMA-bound.
        # Events on the launch stream around 20 back-to-back choose_action launches, right after the timed region.
        ring = ddpg_loop.ring
        t = ring.slot()
        if ddpg_loop.ring_mode:      # image packed once (its ~4 us launch is not in this time), then the policy launch alone
            ddpg_loop._open_step(False)
            one = ddpg_loop.policy_launch
        else:
            one = lambda: ddpg_loop.act(ring.obs[t], ring.act[t], None)
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        one()
        a0.record()
        for _ in range(20):
            one()
        a1.record()
        torch.cuda.synchronize()
        act_ms = a0.elapsed_time(a1) / 20
        from ddpg_trucktrailer_amd import fused
        info = fused.policy_kernel_info(n)
        useful = 2.0 * n * (23 * 400 + 400 * 300 + 300)
        out["roofline_mfma"] = {
            "bound": "mfma", "kernel": info["kernel"], "kernel_ms": act_ms,
            "achieved": info["mfma_flop"] / (act_ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
            "frac": info["mfma_flop"] / (act_ms * 1e-3) / 1e12 / 2500.0, "traffic": None,
            "note": info["note"],
            "algorithmic_f32_tflops": useful / (act_ms * 1e-3) / 1e12,
            "algorithmic_frac_of_16bit_peak": useful / (act_ms * 1e-3) / 1e12 / 2500.0,
            "f32_mfma_peak_tflops": 157.3}
    if ddpg_loop is not None and ddpg_loop.ring_mode:
        gave_up = ddpg_loop.ring.policy_gave_up()      # (device-memory hand-over of the policy image: include/ttenv.h)
        late = ddpg_loop.handover_gave_up[gave_up_in_setup:]
        assert gave_up == 0 and not late, (f"a launch gave up waiting for the other chain of its step (step {(gave_up or late[0]) - 1}) in or "
                                           "after the timed region: policy for its image, or learn() for the env step")
        out["config"]["policy_image_handover"] = ("graph edge" if ddpg_loop.policy_edge() == "graph" else "device memory (epoch word)")
        if handover_fallback:
            out["config"]["policy_image_handover"] += " (fallback: a launch gave up waiting during setup)"
    if ar_us is not None:
        out["allreduce_us"] = ar_us
    if world > 1 and ddpg_loop is not None:
        # data-parallel ranks must hold the SAME weights after every update (gradients averaged at both optimizer sites, rank-ordered
        # sums in the peer-to-peer exchange): compared here, after the timed region -- a run whose ranks drifted apart is not a
        # data-parallel run, whatever its throughput
        same, hs = weights_agree(ddpg_loop, dev, world)
        out["config"]["dp_mode"]["ranks_hold_the_same_weights"] = same
        if not same:
            print(f"bench.py: rank {rank}: the ranks' weights differ after the run (checksums {[x.tolist() for x in hs]}): the gradient "
                  "exchange is broken on this node", file=sys.stderr, flush=True)
    if rank == 0:
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    wd.enter("shutdown")
    diverged = world > 1 and ddpg_loop is not None and not out["config"]["dp_mode"].get("ranks_hold_the_same_weights", True)
    if world > 1:
        dist.destroy_process_group()
    wd.stop()
    if diverged:
        sys.exit(4)


if __name__ == "__main__":
    main()
