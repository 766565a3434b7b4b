#!/bin/bash
# Host-side robustness run: glb loader + flattener + rig compiler under ASan/UBSan on mutated files.
# usage: tools/fuzz/run.sh [iterations]      (CPU only; a few seconds per 10,000 variants)
set -e
cd "$(dirname "$0")/../.."
IT=${1:-20000}
B=/tmp/skinrt_fuzz; mkdir -p $B
python3 - <<PY
import sys; sys.path.insert(0, ".")
from renderer_amd import synth
from tests.glb_writer import write_glb
for name, spec in (("a", synth.tiny_spec(seed=51, matrix_nodes=2)),
                   ("b", synth.tiny_spec(seed=52, sets=2, target_mode="sparse", target_frac=0.3, targets=4))):
    glb, _ = write_glb(synth.make_rig(spec), seed=7)
    open("$B/" + name + ".glb", "wb").write(glb)
PY
SAN="-fsanitize=address,undefined,float-cast-overflow -fno-sanitize-recover=undefined,float-cast-overflow -fno-omit-frame-pointer"
g++ -O1 -g -std=c++17 $SAN -ffp-contract=off tools/fuzz/fuzz_glb.cpp renderer_amd/host/host_capi.cpp renderer_amd/host/gltf_loader.cpp \
    renderer_amd/host/flatten.cpp renderer_amd/csrc/rig_compile.cpp -o $B/fuzz_glb
for f in a b; do for seed in 1 2 3; do ASAN_OPTIONS=detect_leaks=1 timeout 900 $B/fuzz_glb $B/$f.glb $IT $seed; done; done
