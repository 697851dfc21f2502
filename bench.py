#!/usr/bin/env python3
"""bench.py -- env-steps/s of the MI355X truck-trailer hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one vector step over all envs of the rank.  Workloads:
  ddpg  (default, BASELINE.json config 3): actor forward + OU noise for N envs -> env step kernel
        -> transitions into the device replay ring -> one DDPG learn() (batch 256) per vector step;
  env   (config 2): random policy (Philox) -> env step kernel, auto-reset.
Inputs are synthetic (reference reset distribution, random-init 400x300 networks) and resident in
HBM before the timed region.  One JSON line on rank 0; see DESIGN.md "Measurement" for the fields."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B_ALG = 313            # algorithmic bytes per env-step
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n-envs", type=int, default=65536, help="envs PER GPU (weak scaling)")
    ap.add_argument("--workload", choices=("ddpg", "env"), default="ddpg")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--replay-slots", type=int, default=64, help="ring length in vector steps (capacity = slots*N)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-graph", action="store_true", help="launch everything eagerly (no hipGraphs)")
    ap.add_argument("--step-graph", type=int, default=4, help="ddpg workload: whole vector steps per captured hipGraph "
                    "(0: only learn() is captured)")
    ap.add_argument("--torch-learn", action="store_true", help="learn() through torch autograd instead of the fused HIP kernels")
    ap.add_argument("--graph-steps", type=int, default=50, help="env workload: vector steps per captured hipGraph")
    return ap.parse_args()


def pmc_traffic(n):
    """HBM bytes per k_step launch from the committed PMC passes (profiles/r01_pmc_traffic.json: rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    bench.py cannot collect counters itself; null when the summary is absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return float(json.load(f)["traffic_B_per_env_step"]) * n
    except Exception:
        return None


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
    """Oracle timed on the host cores (rank 0, N=1 only): the structure-faithful Python/scipy twin of the
    reference's step loop on ONE core (the reference is single-threaded), random policy, reset on done.
    The plain-C port on all cores is reported next to it as the optimised-CPU line."""
    import numpy as np
    from oracle import c_oracle
    from oracle.simv2_twin import Simv2Twin
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
    cores = len(os.sched_getaffinity(0))
    c_oracle.rollout_random(256, 50, seed=1, nthreads=cores)
    t1 = time.perf_counter()
    cn, _ = c_oracle.rollout_random(4096, 400, seed=2, nthreads=cores)
    cdt = time.perf_counter() - t1
    return {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} steps of oracle/simv2_twin.py (numpy + scipy solve_ivp RK45 + reward object per step), "
                      f"uniform-random steering, reset on done, {dt:.1f} s",
            "full_loop": cpu_full_loop(),
            "c_port": {"value": cn / cdt, "unit": "env-steps/s", "cores": cores,
                       "sample": f"{cn} steps of oracle/tt_oracle.c (fixed DP5, OpenMP), {cdt:.2f} s"}}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path is a HIP kernel)"
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

    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv

    n = args.n_envs
    env = TruckTrailerVecEnv(n, device=dev)
    env.reset(seed=27 + rank)

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

        def one_step(timed):
            if graph is not None:
                graph.replay()
            else:
                env.step_random(123 + rank, auto_reset=True)
        workload = (f"simv2 N={n}/GPU, random policy U(-1,1)*pi/4 drawn in-kernel, auto-reset "
                    f"(BASELINE config 2 at bench size), hipGraph of {graph_k} steps")
        extra = {"graph_steps": graph_k}
    else:
        from ddpg_trucktrailer_amd.rollout import DDPGRollout
        loop = DDPGRollout(env, batch_size=args.batch, replay_slots=args.replay_slots, seed=27 + rank,
                           world_size=world, use_graph=not args.no_graph, fused_learn=not args.torch_learn,
                           graph_steps=args.step_graph)

        loop.prepare()       # one-off work (4 untimed vector steps + graph capture) before warm-up and the timed region

        def one_step(timed):
            loop.step()
        ddpg_loop = loop
        workload = (f"simv2 N={n}/GPU + full DDPG learn() per vector step (actor/critic 400x300, batch {args.batch}, "
                    f"OU noise, replay ring {args.replay_slots}xN) (BASELINE config 3)")
        extra = {"batch": args.batch, "replay_capacity": args.replay_slots * n,
                 "launch": (f"hipGraphs of {loop.graph_steps} whole vector steps" if loop.graph_steps else
                            ("eager, learn() as a hipGraph" if loop.use_graph else "eager"))}

    def run(k_steps, timed):
        if ddpg_loop is not None:
            ddpg_loop.run(k_steps)       # hipGraphs of whole vector steps (policy + env step + learn), eager remainder
            return
        for _ in range(k_steps // graph_k):
            one_step(timed)

    run(args.warmup, False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    captured = graph_k > 1 or (ddpg_loop is not None and ddpg_loop.graph_steps > 0)
    if not captured:
        env.profile(args.steps)      # per-dispatch HIP events on the step kernel, inside the timed region
    t0 = time.perf_counter()
    run(args.steps, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if captured:
        # a captured launch cannot carry its own events: time the same kernel in the same loop on the same state right
        # after, eagerly, with the per-dispatch events (the timed region above is untouched by this)
        env.profile(min(args.steps, 2000))
        for _ in range(min(args.steps, 2000)):
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
    out = {
        "metric": "env-steps/s", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": dict({"workload": workload, "n_envs_per_gpu": n, "n_envs_total": n * world}, **extra),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(n), "kernel": "k_step",
                     "kernel_ms": kern_ms, "alg_bytes_per_env_step": B_ALG,
                     "kernel_env_steps_per_s": n / (kern_ms * 1e-3)},
    }
    if ddpg_loop is not None and ddpg_loop.fused_act:
        # the loop's largest kernel is the N-env policy forward (k_split_pack + k_mlp_split, csrc/ttnet_split.hip): MFMA-bound.
        # Events on the launch stream around 20 back-to-back choose_action launches, right after the timed region.
        ring = ddpg_loop.ring
        t = ring.slot()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ddpg_loop.act(ring.obs[t], ring.act[t], None)
        e0.record()
        for _ in range(20):
            ddpg_loop.act(ring.obs[t], ring.act[t], None)
        e1.record()
        torch.cuda.synchronize()
        act_ms = e0.elapsed_time(e1) / 20
        waves = 4 * ((n + 127) // 128)
        bf16_flop = waves * 1500 * 32768.0          # v_mfma_f32_32x32x16_bf16: 25 k16 steps x 10 tiles x 6 products per wave
        out["roofline_mfma"] = {
            "bound": "mfma", "kernel": "k_split_pack + k_mlp_split (choose_action for N envs)", "kernel_ms": act_ms,
            "achieved": bf16_flop / (act_ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
            "frac": bf16_flop / (act_ms * 1e-3) / 1e12 / 2500.0, "traffic": None,
            "note": "bf16 MFMA FLOP executed (six bf16 products per f32 product block) over dense bf16 peak; layer 1 "
                    "(f32 MFMA, 2 % of the FLOP) not counted",
            "algorithmic_f32_tflops": 2.0 * n * (23 * 400 + 400 * 300 + 300) / (act_ms * 1e-3) / 1e12,
            "f32_mfma_peak_tflops": 157.3}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
