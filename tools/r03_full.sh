#!/bin/bash
# GPU box: everything after a change -- all GPU tests, learn() stamps, bench lines, a kernel trace of the loop with one step's timeline.
set -o pipefail
out=gpurun_out/${1:-r03e}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
tail -4 $out/tests.log
grep -q "tests rc=0" $out/tests.log || exit 1
TT_LIB_PATH=$PWD/tools/dbg/libttenv_stamps.so timeout -k 10 300 python3 tools/learn_blocks.py > $out/learn_blocks.txt 2>&1; grep -A9 "hipGraph replays (repeat 2)" $out/learn_blocks.txt; grep -A9 "policy grids beside it) (repeat 2)" $out/learn_blocks.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver.json 2> $out/bench_driver.err
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
timeout -k 10 300 python3 bench.py --no-cpu-baseline --n-envs 4096 > $out/bench_n4096.json 2> $out/bench_n4096.err
timeout -k 10 300 python3 bench.py --no-cpu-baseline --updates-per-step 64 --steps 300 --warmup 40 > $out/bench_u64.json 2> $out/bench_u64.err
python3 - <<PY
import json
for f in ("bench_driver","bench_default","bench_n4096","bench_u64"):
    try:
        d=json.load(open("$out/%s.json"%f)); print(f, round(d["ms_per_step"],5), "%.3e"%d["value"], d["timing"]["median_ms_per_step"], d["roofline"]["kernel_ms"], d.get("roofline_mfma",{}).get("kernel_ms"))
    except Exception as e: print(f, "failed", e)
PY
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ddpg -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --repeats 0 > $out/ddpg.log 2>&1 || echo "trace failed"
python3 tools/timeline.py $out/ddpg 71 > $out/ddpg_step_timeline.txt 2>&1; head -40 $out/ddpg_step_timeline.txt
