#!/bin/bash
# does the skin stage speed up with sustained load / in later processes?
export EXP_ONLY="${EXP_ONLY:-no targets}"
for k in 1 2; do
echo "== long $k"; EXP_REPS=3000 timeout -k 10 200 python tools/exp_morph.py 2>&1 | grep -v "skinrt plan" || exit 1
done
echo "== short after"; EXP_REPS=20 timeout -k 10 200 python tools/exp_morph.py 2>&1 | grep -v "skinrt plan" || exit 1
