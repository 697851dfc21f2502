# GPU box: does any runtime knob move the loop?  (default bench line under a few HIP / ROCclr environment settings, alternating with the plain run)
set -o pipefail
out=gpurun_out/${1:-knobs}; mkdir -p $out; : > $out/knobs.txt
run() { timeout -k 10 200 env "$@" python3 bench.py --no-cpu-baseline --steps 2000 --warmup 200 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*'.ljust(44), round(d['ms_per_step'],5))" >> $out/knobs.txt || echo "$* failed" >> $out/knobs.txt; }
for k in A=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 A=1 GPU_MAX_HW_QUEUES=2 GPU_MAX_HW_QUEUES=8 A=1 HIP_FORCE_DEV_KERNARG=0 ROC_SIGNAL_POOL_SIZE=256 A=1 DEBUG_HIP_GRAPH_BATCH_SIZE=1 HSA_ENABLE_INTERRUPT=0 A=1; do run $k; done
cat $out/knobs.txt
