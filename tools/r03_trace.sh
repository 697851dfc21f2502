#!/bin/bash
set -o pipefail
out=gpurun_out/${1:-r03q}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
TT_POLICY_EDGE=graph python3 bench.py --no-cpu-baseline > $out/bench_graph_edge.json 2> $out/bench_graph_edge.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ddpg -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --repeats 0 > $out/ddpg.log 2>&1 || echo trace failed
python3 tools/timeline.py $out/ddpg 71 > $out/ddpg_step_timeline.txt 2>&1; head -28 $out/ddpg_step_timeline.txt
python3 -c "
import json
for f in ('bench_default','bench_graph_edge'):
    d=json.load(open('$out/%s.json'%f)); print(f, round(d['ms_per_step'],5), '%.3e'%d['value'])
print(open('$out/ddpg.log').read()[-400:])"
