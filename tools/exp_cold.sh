#!/bin/bash
# does the skin stage change speed within one long process?
export EXP_ONLY="${EXP_ONLY:-no targets}"
EXP_REPS=${EXP_REPS:-20000} timeout -k 10 400 python -u tools/exp_morph.py 2>&1 | grep -v "skinrt plan" || exit 1
