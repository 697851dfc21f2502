#!/bin/bash
# round-4 experiment 2 (one box, A/B): non-temporal obs/reward/done stores at N = 65536 inside the loop; the policy grid's cap
out=gpurun_out/r04q; mkdir -p $out; cd ${GRAFT_REPO_ROOT:?}
line() { python3 -c "import json,sys;d=json.load(open('$1'));print('$2', round(d['ms_per_step'],5), d['timing']['median_ms_per_step'], round(d['roofline']['kernel_ms']*1e3,2), round(d['roofline_mfma']['kernel_ms']*1e3,1))"; }
for rep in 1 2; do
python3 bench.py --no-cpu-baseline --repeats 3 > $out/a_$rep.json 2>/dev/null; line $out/a_$rep.json "default     "
TT_NT_ENVS=1 python3 bench.py --no-cpu-baseline --repeats 3 > $out/nt_$rep.json 2>/dev/null; line $out/nt_$rep.json "nt stores   "
done
for wg in 150 160 171 180 192 210; do
TT_POLICY_WG=$wg python3 bench.py --no-cpu-baseline --repeats 3 > $out/wg_$wg.json 2>/dev/null; line $out/wg_$wg.json "cap $wg     "
done
