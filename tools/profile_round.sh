#!/bin/bash
# GPU box: the measurements of one round, written under gpurun_out/$1 (copy what is to be judged into profiles/).
#   bench lines (default = BASELINE config 3 pipelined; --serial; env = config 2; simv1 = config 5; N = 4096), rocprofv3
#   kernel stats + traces of the ddpg workload in both orders and of the env workload, the PMC passes for the step
#   kernel's HBM traffic (FETCH_SIZE and WRITE_SIZE in separate runs) and an SQ pass for the policy kernel.
set -o pipefail
tag=${1:-round}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver_form.json 2> $out/bench_driver_form.err || exit 1
python3 bench.py --serial --no-cpu-baseline > $out/bench_serial.json 2> $out/bench_serial.err || exit 1
python3 bench.py --workload env --no-cpu-baseline > $out/bench_env.json 2> $out/bench_env.err || exit 1
python3 bench.py --variant simv1 --no-cpu-baseline > $out/bench_simv1.json 2> $out/bench_simv1.err || exit 1
python3 bench.py --variant simv1 --workload env --no-cpu-baseline > $out/bench_simv1_env.json 2> $out/bench_simv1_env.err || exit 1
python3 bench.py --n-envs 4096 --no-cpu-baseline > $out/bench_n4096.json 2> $out/bench_n4096.err || exit 1
python3 bench.py --n-envs 4096 --workload env --no-cpu-baseline > $out/bench_n4096_env.json 2> $out/bench_n4096_env.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ddpg -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --repeats 0 > $out/ddpg.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ddpg_serial -- python3 bench.py --serial --steps 100 --warmup 20 --no-cpu-baseline --repeats 0 > $out/ddpg_serial.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/env -- python3 bench.py --workload env --steps 200 --no-cpu-baseline --repeats 0 > $out/env.log 2>&1 || exit 1
python3 tools/timeline.py $out/ddpg 71 > $out/ddpg_step_timeline.txt 2>&1
python3 tools/timeline.py $out/ddpg_serial 60 > $out/ddpg_serial_step_timeline.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --workload env --graph-steps 1 --steps 100 --warmup 20 --no-cpu-baseline --repeats 0 > $out/pmc_$c.log 2>&1 || exit 1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -- python3 bench.py --serial --steps 20 --warmup 8 --step-graph 0 --no-cpu-baseline --repeats 0 > $out/pmc_sq.log 2>&1 || echo "SQ pass failed (see $out/pmc_sq.log)"
find $out -name "*.csv" | head -40
python3 tools/sweep.py > $out/sweep_env.md 2> $out/sweep_env.err
bash tools/sweep_loop.sh > $out/sweep_loop.md 2> $out/sweep_loop.err
