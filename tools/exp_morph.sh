#!/bin/bash
# morph ablations with a timing build and with the stamps build
SKINRT_LIB=$PWD/renderer_amd/lib/libskinrt_${1:-nt}.so timeout -k 10 300 python tools/exp_morph.py > gpurun_out/exp_morph.log 2>&1 || { tail -5 gpurun_out/exp_morph.log; exit 1; }
cat gpurun_out/exp_morph.log
SKINRT_LIB=$PWD/renderer_amd/lib/libskinrt_${2:-st2}.so timeout -k 10 300 python tools/exp_morph.py > gpurun_out/exp_morph_st.log 2>&1 || { tail -5 gpurun_out/exp_morph_st.log; exit 1; }
grep -v "skinrt plan" gpurun_out/exp_morph_st.log
