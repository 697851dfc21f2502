#!/bin/bash
# GPU box: socket power and shader clock while the loop runs (is the period set by the power cap?)
out=gpurun_out/${1:-power}; mkdir -p $out
rocm-smi --showpower --showclocks --showmaxpower > $out/idle.txt 2>&1
timeout -k 10 300 python3 bench.py --steps 150000 --warmup 20 > $out/bench.json 2> $out/bench.err &
BP=$!
sleep 45
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" >> $out/load.txt; echo "--" >> $out/load.txt; sleep 1; done
wait $BP
tail -c 400 $out/bench.json; echo; cat $out/idle.txt | grep -E "Power|sclk|Max" ; cat $out/load.txt | head -30
