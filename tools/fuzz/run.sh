#!/bin/bash
# Host-side robustness run: glb and PMX loaders + flattener + rig compiler under ASan/UBSan on mutated files.
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
from tests.glb_writer import write_glb_multi
base = synth.make_rig(synth.tiny_spec(seed=61, verts=300, joints=10, nodes=18, clips=2, keys=4, targets=5, sets=2, weight_channels=1))
multi, _ = synth.make_multi_rig(base, seed=3, fractions=(0.4, 0.35, 0.25), rigid_frac=0.1, rigid_sections=2)
open("$B/c.glb", "wb").write(write_glb_multi(multi, seed=5)[0])   # three skins, two un-skinned mesh nodes, a weights channel
from tests.pmx_writer import write_pmx
from tests.test_pmx_loader import model
pos, nrm, idx, bones, deforms, morphs, _, _ = model()
open("$B/a.pmx", "wb").write(write_pmx(pos, nrm, idx, bones, deforms, morphs, materials=[300, 300],
                                       other_morphs=("group", "bone", "uv", "material", "flip", "impulse")))
PY
SAN="-fsanitize=address,undefined,float-cast-overflow -fno-sanitize-recover=undefined,float-cast-overflow -fno-omit-frame-pointer"
g++ -O1 -g -std=c++17 $SAN -ffp-contract=off tools/fuzz/fuzz_glb.cpp renderer_amd/host/host_capi.cpp renderer_amd/host/gltf_loader.cpp \
    renderer_amd/host/pmx_loader.cpp renderer_amd/host/flatten.cpp renderer_amd/csrc/rig_compile.cpp -o $B/fuzz_glb
g++ -O1 -g -std=c++17 $SAN -ffp-contract=off tools/fuzz/fuzz_pmx.cpp renderer_amd/host/host_capi.cpp renderer_amd/host/gltf_loader.cpp \
    renderer_amd/host/pmx_loader.cpp renderer_amd/host/flatten.cpp renderer_amd/csrc/rig_compile.cpp -o $B/fuzz_pmx
for f in a b c; do for seed in 1 2 3; do ASAN_OPTIONS=detect_leaks=1 timeout 900 $B/fuzz_glb $B/$f.glb $IT $seed; done; done
for seed in 1 2 3; do ASAN_OPTIONS=detect_leaks=1 timeout 900 $B/fuzz_pmx $B/a.pmx $IT $seed; done
