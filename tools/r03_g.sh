#!/bin/bash
set -o pipefail
out=gpurun_out/${1:-r03g}
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
tail -4 $out/tests.log
grep -q "tests rc=0" $out/tests.log || exit 1
for tag in default graphedge; do
  if [ $tag = graphedge ]; then export TT_POLICY_EDGE=graph; else unset TT_POLICY_EDGE; fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline > $out/bench_$tag.json 2> $out/bench_$tag.err
done
unset TT_POLICY_EDGE
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver.json 2> $out/bench_driver.err
timeout -k 10 300 python3 bench.py --no-cpu-baseline --n-envs 4096 > $out/bench_n4096.json 2> $out/bench_n4096.err
timeout -k 10 300 python3 bench.py --no-cpu-baseline --updates-per-step 64 --steps 300 --warmup 40 > $out/bench_u64.json 2> $out/bench_u64.err
python3 - <<PY
import json
for f in ("bench_default","bench_graphedge","bench_driver","bench_n4096","bench_u64"):
    try:
        d=json.load(open("$out/%s.json"%f)); print(f, round(d["ms_per_step"],5), "%.3e"%d["value"], d["timing"]["median_ms_per_step"], d["roofline"]["kernel_ms"], d.get("roofline_mfma",{}).get("kernel_ms"))
    except Exception as e: print(f, "failed", e)
PY
