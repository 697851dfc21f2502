#!/bin/bash
# GPU box: the whole -m gpu suite, build() + smoke() in one process, and the bench line in the driver form.
set -o pipefail
out=gpurun_out/${1:-gpu_check}
mkdir -p $out
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
tail -5 $out/tests.log
grep -q "tests rc=0" $out/tests.log || { grep -n "Error\|assert" $out/tests.log | head -20; exit 1; }
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err; tail -c 600 $out/bench_driver.json
