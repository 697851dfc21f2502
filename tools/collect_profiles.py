#!/usr/bin/env python3
"""Copy what a tools/profile_round4.sh run left under gpurun_out/<tag> into profiles/<prefix>_* (the files the documents cite).
usage: collect_profiles.py gpurun_out/<tag> profiles/r04"""
import glob, os, shutil, subprocess, sys
src, dst = sys.argv[1], sys.argv[2]
BENCH = {"default": "default", "driver_form": "driver_form_20_5", "env": "env_workload", "graph_edge": "graph_edge", "n4096": "n4096",
         "n4096_env": "n4096_env_workload", "serial": "serial_order", "simv1": "simv1", "simv1_env": "simv1_env_workload",
         "u64": "updates_per_step_64", "p2p_world1": "p2p_world_size_1",
         "dp_structure_noop_collectives": "dp_structure_noop_collectives"}
for a, b in BENCH.items():
    f = os.path.join(src, f"bench_{a}.json")
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, f"{dst}_bench_{b}.json")
STATS = {"ddpg": "ddpg_workload", "ddpg_serial": "ddpg_serial_order", "env": "env_workload", "env4m": "env_workload_4m"}
for a, b in STATS.items():
    fs = sorted(glob.glob(os.path.join(src, a, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getsize)
    if fs:
        shutil.copy(fs[-1], f"{dst}_{b}_kernel_stats.csv")       # (the largest file: the process that ran the kernels)
for a, b in (("ddpg_step_timeline.txt", "ddpg_step_timeline.txt"), ("ddpg_serial_step_timeline.txt", "ddpg_serial_order_step_timeline.txt"),
             ("learn_blocks.txt", "learn_workgroup_stamps.txt"), ("pmc4m_summary.json", "pmc_k_step_4m_envs.json"),
             ("step_timeline.txt", "step_timeline_stamps.txt"), ("time_p2p.txt", "p2p_learn_chain_world_size_1.txt"),
             ("ipc_probe.json", "ipc_probe.json"), ("time_actor_cap.txt", "policy_launch_by_cap.txt"), ("soak.txt", "soak.txt"),
             ("sweep_env.md", "nsweep_env.md"), ("sweep_loop.md", "nsweep_loop.md")):
    f = os.path.join(src, a)
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, f"{dst}_{b}")
subprocess.check_call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_summary.py"), src, dst])
print("\n".join(sorted(glob.glob(dst + "_*"))))
