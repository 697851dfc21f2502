#!/bin/bash
# one box: bench forms with / without the settle phase; kernel sequence of the learn chain at 64 updates per step
set -o pipefail
out=gpurun_out/${1:-r03j}
mkdir -p $out
for rep in 1 2; do
for sm in 60 0; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --settle-ms $sm > $out/bench_driver_s${sm}_$rep.json 2> $out/bench_driver_s$sm.err
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --settle-ms $sm > $out/bench_default_s${sm}_$rep.json 2> $out/bench_default_s$sm.err
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --n-envs 4096 --settle-ms $sm > $out/bench_driver4096_s${sm}_$rep.json 2> $out/bench_driver4096_s$sm.err
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload env --settle-ms $sm > $out/bench_driverenv_s${sm}_$rep.json 2> $out/bench_driverenv_s$sm.err
done
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["ms_per_step"],5), "%.3e"%d["value"], d["timing"]["median_ms_per_step"], d["config"].get("setup_vector_steps"))
    except Exception as e: print(f, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/u64 -- python3 bench.py --updates-per-step 64 --steps 40 --warmup 20 --no-cpu-baseline --repeats 0 --settle-ms 0 > $out/u64.log 2>&1 || echo "u64 trace failed"
f=$(ls -S $out/u64/*/*_kernel_trace.csv | head -1)
python3 tools/trace_gaps.py $f 0.5 24 > $out/u64_gaps.txt 2>&1; cat $out/u64_gaps.txt
