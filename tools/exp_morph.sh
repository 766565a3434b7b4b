#!/bin/bash
# morph ablation table (tools/exp_morph.py) for several builds: tools/exp_morph.sh name1 name2 ... ("default" = libskinrt.so)
for n in "$@"; do
  if [ "$n" = default ]; then lib=renderer_amd/lib/libskinrt.so; else lib=renderer_amd/lib/libskinrt_$n.so; fi
  echo "== $n"
  SKINRT_LIB=$PWD/$lib timeout -k 10 300 python tools/exp_morph.py 2>&1 | grep -v "skinrt plan" || exit 1
done
