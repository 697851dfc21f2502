#!/bin/bash
# GPU box: the measurements of one round, written under gpurun_out/$1 (copy what is to be judged into profiles/).
#   bench lines (default = BASELINE config 3; env = config 2), rocprofv3 kernel stats of both, the PMC passes for the
#   step kernel's HBM traffic (FETCH_SIZE and WRITE_SIZE in separate runs) and an SQ pass for the actor kernel.
set -o pipefail
tag=${1:-round}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
python3 bench.py --workload env --no-cpu-baseline > $out/bench_env.json 2> $out/bench_env.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ddpg -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $out/ddpg.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/env -- python3 bench.py --workload env --steps 200 --no-cpu-baseline > $out/env.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --workload env --graph-steps 1 --steps 100 --warmup 20 --no-cpu-baseline > $out/pmc_$c.log 2>&1 || exit 1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 20 --warmup 8 --step-graph 0 --no-cpu-baseline > $out/pmc_sq.log 2>&1 || echo "SQ pass failed (see $out/pmc_sq.log)"
find $out -name "*.csv" | head -40
