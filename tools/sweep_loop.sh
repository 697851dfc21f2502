#!/bin/bash
# GPU box: the DDPG loop (BASELINE config 3) against N envs per GPU; prints a markdown table (copy into profiles/).
echo "| N envs | ms per vector step | env-steps/s | \`k_step\` us (events, in loop) | frac of 8 TB/s | policy forward us | executed-MFMA frac of 2.5 PFLOP/s | useful f32 TFLOP/s |"
echo "|---|---|---|---|---|---|---|---|"
for n in 4096 16384 65536 262144 1048576; do
  steps=2000; warm=200
  if [ $n -ge 262144 ]; then steps=400; warm=40; fi
  python3 bench.py --n-envs $n --steps $steps --warmup $warm --no-cpu-baseline --repeats 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; m=d['roofline_mfma']
print('| %d | %.3f | %.3e | %.1f | %.3f | %.1f | %.3f | %.0f |' % (d['config']['n_envs_per_gpu'], d['ms_per_step'], d['value'], r['kernel_ms']*1e3, r['frac'], m['kernel_ms']*1e3, m['frac'], m['algorithmic_f32_tflops']))" || exit 1
done
