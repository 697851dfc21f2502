# GPU box: A/B of two builds of the library on the bench configurations (alternating runs, one box).
# usage: ab_lib.sh <other.so> [tag] [tests]   -- "tree" = ddpg-trucktrailer_amd/libttenv.so, "other" = the library given
set -o pipefail
other=$1; out=gpurun_out/${2:-ab}; mkdir -p $out; : > $out/ab.txt
if [ -n "$3" ]; then
  timeout -k 10 900 python3 -m pytest $3 -x -q -m gpu > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log; tail -3 $out/tests.log
  grep -q "rc=0" $out/tests.log || exit 1
fi
run() { timeout -k 10 200 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$*', round(d['ms_per_step'],5), '%.4g' % d['value'], 'policy_alone_ms', round((d.get('roofline_mfma') or {}).get('kernel_ms') or 0, 5))" >> $out/ab.txt; }
for i in 1 2 3; do
for v in tree other; do
  if [ $v = other ]; then export TT_LIB_PATH=$PWD/$other; else unset TT_LIB_PATH; fi
  run --steps 2000 --warmup 200
  [ "$AB_QUICK" = 1 ] && continue
  run --steps 2000 --warmup 200 --n-envs 4096
  run --steps 500 --warmup 50 --updates-per-step 64
done; done
cat $out/ab.txt
