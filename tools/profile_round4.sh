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
if [ "${PMC4M:-0}" = "1" ]; then      # what bounds k_step at N = 4 M envs (round 3's study: kernel trace + SQ / TCC / FETCH / WRITE passes, each its own run)
N=4194304
B="python3 bench.py --workload env --n-envs $N --graph-steps 1 --steps 40 --warmup 10 --no-cpu-baseline --repeats 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/env4m -- $B > $out/env4m.log 2>&1 || echo "env4m trace failed"
rocprofv3 -L > $out/counters.txt 2>&1
SQ1=$(python3 tools/pmc_pick.py $out/counters.txt 8 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU)
SQ2=$(python3 tools/pmc_pick.py $out/counters.txt 8 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS)
T1=$(python3 tools/pmc_pick.py $out/counters.txt 4 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_sum)
T2=$(python3 tools/pmc_pick.py $out/counters.txt 4 TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_DRAM_sum)
echo "SQ1=$SQ1 | SQ2=$SQ2 | T1=$T1 | T2=$T2" > $out/pmc_sets.txt
i=0
for set in "$SQ1" "$SQ2" "$T1" "$T2" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  [ -z "$set" ] && continue
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/pmc4m_$i -- $B > $out/pmc4m_$i.log 2>&1 || echo "pmc pass $i ($set) failed" >> $out/pmc_sets.txt
done
python3 tools/pmc_summary4m.py $out > $out/pmc4m_summary.json 2> $out/pmc4m_summary.err
for nt in 0 1; do
  echo "TT_NT_ENVS=$nt" >> $out/step_nt.txt
  TT_NT_ENVS=$nt timeout -k 10 200 python3 tools/ab_kernel.py 65536 1048576 4194304 >> $out/step_nt.txt 2>&1
done
fi
ls $out | head -80
