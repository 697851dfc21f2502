#!/bin/bash
# round-4 experiment 1: kernel-argument placement (HIP_FORCE_DEV_KERNARG) A/B on the bench line and on learn()'s stamps; p2p structure at world size 1
set -o pipefail
out=gpurun_out/r04c; mkdir -p $out
cd ${GRAFT_REPO_ROOT:?}
for v in default 1 0; do
  if [ $v = default ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  python3 bench.py --no-cpu-baseline --repeats 3 > $out/bench_kernarg_$v.json 2> $out/bench_kernarg_$v.err
  python3 -c "import json;d=json.load(open('$out/bench_kernarg_$v.json'));print('kernarg=$v', d['ms_per_step'], d['timing']['cold_ms_per_step'], d['timing']['median_ms_per_step'])"
  TT_LB_SHORT=1 TT_LIB_PATH=$PWD/tools/dbg/libttenv_stamps.so timeout -k 10 200 python3 tools/learn_blocks.py > $out/learn_blocks_kernarg_$v.txt 2>&1
  grep "chain" $out/learn_blocks_kernarg_$v.txt | tail -2
done
unset HIP_FORCE_DEV_KERNARG
python3 bench.py --no-cpu-baseline --repeats 3 --dp-mode p2p > $out/bench_p2p_w1.json 2> $out/bench_p2p_w1.err
python3 -c "import json;d=json.load(open('$out/bench_p2p_w1.json'));print('p2p w1', d['ms_per_step'], d['timing']['median_ms_per_step'], d['config']['launch'][:80])"
TT_FORCE_DP=1 TT_DP_GRAPH_COLLECTIVES=1 python3 bench.py --no-cpu-baseline --repeats 3 > $out/bench_forcedp_w1.json 2> $out/bench_forcedp_w1.err
python3 -c "import json;d=json.load(open('$out/bench_forcedp_w1.json'));print('force-dp graph w1', d['ms_per_step'], d['timing']['median_ms_per_step'])"
python3 bench.py --no-cpu-baseline --repeats 3 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err
python3 -c "import json;d=json.load(open('$out/bench_driver.json'));print('driver form', d['ms_per_step'], d['timing']['cold_ms_per_step'], d['timing']['median_ms_per_step'])"
