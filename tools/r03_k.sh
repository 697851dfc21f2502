#!/bin/bash
# one box: bench driver forms with / without the settle phase (synchronize per unit); learn() stamps incl. the gap between updates
set -o pipefail
out=gpurun_out/${1:-r03k}
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_fused_learn.py tests/test_gpu_rollout.py -x -q -m gpu > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
tail -3 $out/tests.log
grep -q "tests rc=0" $out/tests.log || exit 1
for rep in 1 2 3; do
for sm in 100 0; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --settle-ms $sm > $out/bench_driver_s${sm}_$rep.json 2> $out/bench_driver_s$sm.err
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --n-envs 4096 --settle-ms $sm > $out/bench_driver4096_s${sm}_$rep.json 2> $out/bench_driver4096_s$sm.err
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload env --settle-ms $sm > $out/bench_driverenv_s${sm}_$rep.json 2> $out/bench_driverenv_s$sm.err
done
done
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
timeout -k 10 300 python3 bench.py --no-cpu-baseline --n-envs 4096 > $out/bench_n4096.json 2> $out/bench_n4096.err
timeout -k 10 300 python3 bench.py --no-cpu-baseline --updates-per-step 64 --steps 200 --warmup 40 > $out/bench_u64.json 2> $out/bench_u64.err
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["ms_per_step"],5), "%.3e"%d["value"], d["timing"]["median_ms_per_step"], d["config"].get("setup_vector_steps"))
    except Exception as e: print(f, "failed", e)
PY
TT_LIB_PATH=$PWD/tools/dbg/libttenv_stamps.so timeout -k 10 400 python3 tools/learn_blocks.py > $out/learn_blocks.txt 2>&1 || echo "learn_blocks failed"
grep -B1 -A9 "third of three\|LAST of" $out/learn_blocks.txt | grep -v phases | tail -60
