#!/bin/bash
# A/B of library builds: tools/ab_lib.sh out.log "ENV=.. ENV2=.." ...   (each argument = one bench run's environment)
set -e
out=$1; shift
: > $out
for envs in "$@"; do
  echo "== $envs" >> $out
  env $envs timeout -k 10 150 python bench.py --no-cpu-baseline --steps 20 --warmup 5 ${BENCH_ARGS} 2>>$out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:3], round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['stage_ms'].items() if v is not None}, 'exact', d['config']['exact_arithmetic'])" >> $out
done
cat $out
