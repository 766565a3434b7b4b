#!/bin/bash
# placement search: several processes, what the search sees and what the stage takes afterwards
export EXP_ONLY="${EXP_ONLY:-targets}" SKINRT_DEBUG_PLACE=1
for k in 1 2 3 4; do
  echo "== process $k"
  timeout -k 10 200 python tools/exp_morph.py 2>&1 | grep -E "placement|skin " || exit 1
done
