#!/bin/bash
# k_skin_fix morphing a batch of instances per read of a pair's deltas (product: 4) against one instance at a time
# (make -C renderer_amd/csrc variant NAME=fixb1 FLAGS=-DSKINRT_FIX_BATCH=1): parity first, then C3 / C5 / the face cases.
set -u
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_random_rigs.py tests/test_gpu_multiskin.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1
L=$PWD/renderer_amd/lib
for wl in c3 c5; do
  BENCH_ARGS="--workload $wl --no-extras" tools/ab_lib.sh gpurun_out/r3_ab_fixbatch_$wl.log "SKINRT_LIB=$L/libskinrt.so" "SKINRT_LIB=$L/libskinrt_fixb1.so SKINRT_ALLOW_EXPERIMENTAL=1" "SKINRT_LIB=$L/libskinrt_fixb2.so SKINRT_ALLOW_EXPERIMENTAL=1" \
     "SKINRT_LIB=$L/libskinrt.so" "SKINRT_LIB=$L/libskinrt_fixb1.so SKINRT_ALLOW_EXPERIMENTAL=1" || exit 1
done
for lib in libskinrt.so libskinrt_fixb2.so libskinrt_fixb1.so; do
  echo "== face cases, $lib"
  SKINRT_LIB=$L/$lib SKINRT_ALLOW_EXPERIMENTAL=1 timeout -k 10 300 python3 tools/exp_face.py || exit 1
done
