#!/bin/bash
# Round profile of the bench command on the GPU box: rocprofv3 kernel trace + stats, then PMC
# passes (each its own run), summarised into gpurun_out/<tag>/summary.txt and traffic.json.
# usage: tools/round_profile.sh <tag> [bench args...]     e.g. tools/round_profile.sh r01_c3 --steps 5 --warmup 2
set -u
TAG=$1; shift
# bench.py's --steps, so that the summary can set the timed frames apart from the warm-up
prev=""; for a in "$@"; do [ "$prev" = "--steps" ] && export PROF_TIMED=$a; prev=$a; done
tools/prof.sh $TAG "$@" || exit 1
python3 tools/prof_summary.py gpurun_out/$TAG > gpurun_out/$TAG/summary.txt || exit 1
python3 tools/make_traffic.py gpurun_out/$TAG "$@" > gpurun_out/$TAG/traffic.json || exit 1
python3 tools/prof_summary.py gpurun_out/$TAG --csv > gpurun_out/$TAG/kernel_stats.csv
grep -E "k_skin|k_pose|k_aabb|==|skin stage" gpurun_out/$TAG/summary.txt | head -40
cat gpurun_out/$TAG/traffic.json
