#!/bin/bash
# GPU box, round 3 diagnostics: (1) where learn()'s time goes by workgroup stamps; (2) what bounds k_step at N = 4 M envs
# (kernel stats + SQ and TCC counter passes, each in its own run); (3) k_step variants (workgroup size, non-temporal stores).
set -o pipefail
out=gpurun_out/${1:-r03b}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TT_LIB_PATH=$PWD/tools/dbg/libttenv_stamps.so timeout -k 10 300 python3 tools/learn_blocks.py > $out/learn_blocks.txt 2>&1 || echo "learn_blocks failed"
N=4194304
B="python3 bench.py --workload env --n-envs $N --graph-steps 1 --steps 40 --warmup 10 --no-cpu-baseline --repeats 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/env4m -- $B > $out/env4m.log 2>&1 || echo "env4m trace failed"
rocprofv3 -L > $out/counters.txt 2>&1
SQ1=$(python3 tools/pmc_pick.py $out/counters.txt 8 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU)
SQ2=$(python3 tools/pmc_pick.py $out/counters.txt 8 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE)
T1=$(python3 tools/pmc_pick.py $out/counters.txt 4 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_STALL_sum)
T2=$(python3 tools/pmc_pick.py $out/counters.txt 4 TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_BUSY_sum)
echo "SQ1=$SQ1 | SQ2=$SQ2 | T1=$T1 | T2=$T2" > $out/pmc_sets.txt
i=0
for set in "$SQ1" "$SQ2" "$T1" "$T2" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  [ -z "$set" ] && continue
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/pmc4m_$i -- $B > $out/pmc4m_$i.log 2>&1 || echo "pmc pass $i ($set) failed" >> $out/pmc_sets.txt
done
python3 tools/pmc_summary4m.py $out > $out/pmc4m_summary.txt 2>&1
for v in "" "_b128" "_b512" "_nt" "_nt_b128"; do
  lib=$PWD/tools/dbg/libttenv_step$v.so
  [ -z "$v" ] && lib=$PWD/ddpg-trucktrailer_amd/libttenv.so
  [ -f $lib ] && TT_LIB_PATH=$lib timeout -k 10 200 python3 tools/ab_kernel.py 65536 1048576 4194304 >> $out/step_variants.txt 2>&1
done
cat $out/step_variants.txt
