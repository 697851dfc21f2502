#!/bin/bash
# GPU box: the persistent multi-tile policy kernel -- parity tests, timing alone by cap, stamps, then the loop's bench lines.
set -o pipefail
out=gpurun_out/${1:-r03c}
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_fused_net.py tests/test_gpu_rollout.py -x -q > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
tail -4 $out/tests.log
grep -q "tests rc=0" $out/tests.log || exit 1
timeout -k 10 200 python3 tools/time_actor_cap.py > $out/time_actor_cap.txt 2>&1; cat $out/time_actor_cap.txt
timeout -k 10 200 python3 tools/time_actor.py 65536 262144 > $out/time_actor.txt 2>&1; cat $out/time_actor.txt
TT_LIB_PATH=$PWD/tools/dbg/libttenv_stamps.so timeout -k 10 200 python3 tools/stamps_actor.py 65536 > $out/stamps_actor.txt 2>&1; tail -8 $out/stamps_actor.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver.json 2> $out/bench_driver.err
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
python3 - <<PY
import json
for f in ("bench_driver","bench_default"):
    try:
        d=json.load(open("$out/%s.json"%f)); print(f, d["ms_per_step"], d["value"], d["timing"]["median_ms_per_step"], d["roofline"]["kernel_ms"], d.get("roofline_mfma",{}).get("kernel_ms"))
    except Exception as e: print(f, "failed", e)
PY
