#!/bin/bash
# GPU box, round 4: tools/profile_round.sh (bench lines, rocprofv3 kernel stats + timelines, PMC traffic + SQ pass at N = 65536,
# N-sweeps) + what this round added: the stamp timeline of the loop that ships (tools/step_timeline.py), learn()'s workgroup
# stamps, --updates-per-step 64, graph edge instead of the device-memory hand-over, the peer-to-peer exchange's launches at world
# size 1 (--dp-mode p2p; tools/time_p2p.py), the IPC gate (tools/ipc_probe.py), a soak.  Diagnostic library: built here if missing.
set -o pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo on the GPU box)}"
tag=${1:-r04}
out=gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 ddpg-trucktrailer_amd/build.py --variant stamps > $out/build_stamps.log 2>&1 || { tail -5 $out/build_stamps.log; echo "stamps build failed"; exit 1; }
STAMPS=$PWD/tools/dbg/libttenv_stamps.so
[ -f $STAMPS ] || { echo "$STAMPS missing"; exit 1; }
bash tools/profile_round.sh $tag > $out/profile_round.log 2>&1 || { tail -20 $out/profile_round.log; echo "profile_round failed"; }
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --updates-per-step 64 --steps 300 --warmup 40 --no-cpu-baseline > $out/bench_u64.json 2> $out/bench_u64.err
TT_POLICY_EDGE=graph python3 bench.py --no-cpu-baseline > $out/bench_graph_edge.json 2> $out/bench_graph_edge.err
python3 bench.py --no-cpu-baseline --dp-mode p2p > $out/bench_p2p_world1.json 2> $out/bench_p2p_world1.err
TT_FORCE_DP=1 TT_DP_GRAPH_COLLECTIVES=1 python3 bench.py --no-cpu-baseline > $out/bench_dp_structure_noop_collectives.json 2> $out/bench_dp_structure.err
python3 tools/time_p2p.py > $out/time_p2p.txt 2>&1
timeout -k 10 300 python3 tools/ipc_probe.py > $out/ipc_probe.json 2> $out/ipc_probe.err
TT_LIB_PATH=$STAMPS timeout -k 10 300 python3 tools/learn_blocks.py > $out/learn_blocks.txt 2>&1 || echo "learn_blocks failed"
TT_LIB_PATH=$STAMPS timeout -k 10 300 python3 tools/step_timeline.py > $out/step_timeline.txt 2>&1 || echo "step_timeline failed"
python3 tools/time_actor_cap.py > $out/time_actor_cap.txt 2>&1
timeout -k 10 400 python3 tools/soak.py 65536 200000 > $out/soak.txt 2>&1 || echo "soak failed"
ls $out | head -80
