#!/bin/bash
# output placement search (api.hip tune_output_placement): what it sees at crowd creation and what
# the skin stage takes afterwards, in several processes
export EXP_ONLY="${EXP_ONLY:-targets}" SKINRT_DEBUG_PLACE=1
for k in 1 2 3; do
  echo "== process $k"
  timeout -k 10 200 python tools/exp_morph.py 2>&1 | grep -E "placement|skin " || exit 1
done
