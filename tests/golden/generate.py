#!/usr/bin/env python3
"""Generates the golden fixtures in this directory (run from the repo root:
`python tests/golden/generate.py`).

PARITY UNPINNED: the reference (Rust + WGSL) cannot be built or run in this environment and
holds no tests or fixtures for this path, so these vectors are outputs of the
fp32 CPU oracle (oracle/skin_oracle.c, -O2 -ffp-contract=off) on seeded synthetic rigs — they
freeze the oracle's behaviour (and detect drift of the generator, the oracle or the library),
they do not prove equality with the reference.  What ties the oracle to the reference's
formulas is tests/test_oracle_known_answers.py.

Each fixture is one .npz (plain arrays, no pickles):
  rig_*          the flat rig (renderer_amd/rigfile.py)
  tau, clip_id, morph_weights, root      per-instance inputs
  cfg            [cubic_quirk, normalize_outputs]
  key_index      u32  [I][Cmax]   (bit-exact)
  world          f32  [I][N][12]  rows, skinrt_get_world layout
  palette        f32  [I][P][24]  skinrt_get_palette layout (P = all skins' joints + one entry per un-skinned mesh node)
  position / normal / tangent  f32 [I][V][3];  aabb f32 [I][2][3]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.oracle import Oracle  # noqa: E402
from renderer_amd import rigfile, synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (spec, instances, cubic_quirk, normalize_outputs, with_root[, multi-skin arguments of synth.make_multi_rig])
FIXTURES = {
    # ABI 6: three skins on one node tree + two rigid props (un-skinned sections on two nodes), targets and a weights
    # channel split over the sections; the world rows a caller's InstanceUniform export is made from are in `world`
    "tiny_three_skins_two_props": (synth.tiny_spec(seed=107, verts=160, joints=9, nodes=14, clips=2, keys=4, targets=5, sets=2,
                                                   weight_channels=1), 6, 1, 0, True,
                                   dict(seed=7, fractions=(0.4, 0.35, 0.25), rigid_frac=0.1, rigid_sections=2)),
    # small: <= 64 verts, <= 8 joints, <= 3 keys
    "tiny_all_features": (synth.tiny_spec(seed=101, matrix_nodes=2), 5, 1, 0, True),
    "tiny_spec_hermite": (synth.tiny_spec(seed=102, frac_cubic=0.6), 4, 0, 0, False),
    "tiny_normalized": (synth.tiny_spec(seed=103), 3, 1, 1, False),
    "tiny_two_sets_sparse": (synth.tiny_spec(seed=104, sets=2, target_mode="sparse", target_frac=0.3, targets=4), 4, 1, 0, True),
    "tiny_no_clips_no_targets": (synth.tiny_spec(seed=105, clips=0, targets=0), 2, 1, 0, False),
    # glTF "weights" channels on clips 0 (LINEAR) and 1 (CUBICSPLINE); clip 2 has none
    "tiny_weight_channels": (synth.tiny_spec(seed=106, clips=3, keys=5, weight_channels=2, targets=5), 9, 1, 0, False),
    # one mid-size per config shape, scaled down
    "c1_shape": (synth.CONFIGS["c1"], 1, 1, 0, False),
    "c2_shape": (synth.CONFIGS["c2"], 6, 1, 0, True),
    "c3_shape": (synth.scaled(synth.CONFIGS["c3"], verts=1536, targets=8, keys=6), 3, 1, 0, False),
    "c5_shape": (synth.scaled(synth.CONFIGS["c5"], nodes=96, joints=96, verts=2048, targets=12, target_frac=0.05, keys=4), 2, 1, 0, False),
}


def make(name):
    spec, inst, quirk, norm, with_root = FIXTURES[name][:5]
    rig = synth.make_rig(spec)
    if len(FIXTURES[name]) > 5:
        rig, _ = synth.make_multi_rig(rig, **FIXTURES[name][5])
    crowd = synth.make_crowd(spec, rig, instances=inst, with_root=with_root)
    res = Oracle("f32").evaluate(rig, crowd.tau, crowd.clip_id, crowd.morph_weights, crowd.root,
                                 cubic_quirk=quirk, normalize_outputs=norm)
    extra = dict(tau=crowd.tau, clip_id=crowd.clip_id, cfg=np.array([quirk, norm], np.uint32),
                 key_index=res.key_index, world=res.world_rows(), palette=res.palette_rows(),
                 position=res.position, normal=res.normal, tangent=res.tangent, aabb=res.aabb)
    if crowd.morph_weights is not None: extra["morph_weights"] = crowd.morph_weights
    if crowd.root is not None: extra["root"] = crowd.root
    return rig, extra


def main():
    only = sys.argv[1:]
    for name in FIXTURES:
        if only and name not in only:
            continue
        rig, extra = make(name)
        path = os.path.join(HERE, name + ".npz")
        rigfile.save_rig(path, rig, **extra)
        print(f"{name:28s} N={rig.node_count:4d} J={rig.joint_count:4d} V={rig.vertex_count:5d} K={rig.set_count} "
              f"M={rig.target_count:3d} I={extra['tau'].shape[0]}  {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
