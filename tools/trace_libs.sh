#!/bin/bash
# per-kernel frame durations for several builds: tools/trace_libs.sh name1 name2 ... ("default" = libskinrt.so)
for n in "$@"; do
  if [ "$n" = default ]; then lib=renderer_amd/lib/libskinrt.so; else lib=renderer_amd/lib/libskinrt_$n.so; fi
  echo "== $n"
  SKINRT_LIB=$PWD/$lib bash tools/trace.sh tl_$n --steps 6 --warmup 3 ${BENCH_ARGS:-} | grep -E "k_skin|skin stage" || exit 1
done
