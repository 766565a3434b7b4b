#!/usr/bin/env python3
"""tests/golden/<name>.npz -> a flat little-endian file the Rust cross-check reads (oracle/rust/src/main.rs).
Test infrastructure only.  Layout: magic 'SKRF', u32 version, 16 u32 header words, then the arrays in the order
written below (u32 / i32 / f32, no padding)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from renderer_amd import rigfile  # noqa: E402


def export(npz_path, out_path):
    rig = rigfile.load_rig(npz_path)
    z = np.load(npz_path, allow_pickle=False)
    I = int(z["tau"].shape[0])
    N, J, V, K, M = rig.node_count, rig.joint_count, rig.vertex_count, rig.set_count, rig.target_count
    ts = int(rig.tangent.shape[1])
    cmax = int(z["key_index"].shape[1])
    quirk, norm = (int(x) for x in z["cfg"])
    has_mw, has_root = "morph_weights" in z, "root" in z
    out = bytearray()
    u32 = lambda a: out.extend(np.ascontiguousarray(a, np.uint32).tobytes())
    i32 = lambda a: out.extend(np.ascontiguousarray(a, np.int32).tobytes())
    f32 = lambda a: out.extend(np.ascontiguousarray(a, np.float32).tobytes())
    out.extend(b"SKRF")
    u32([1])
    u32([N, J, V, K, M, len(rig.clips), I, cmax, ts, rig.matrix is not None, rig.inverse_bind is not None,
         rig.default_weights is not None, has_mw, has_root, quirk, norm])
    i32(rig.parent); f32(rig.translation); f32(rig.rotation); f32(rig.scale)
    if rig.matrix is not None:
        f32(rig.matrix); u32(rig.has_matrix)
    u32(rig.joint_node)
    if rig.inverse_bind is not None:
        f32(rig.inverse_bind)
    for clip in rig.clips:
        u32([len(clip)])
        for ch in clip:
            u32([ch.target_node, ch.path, ch.interpolation, ch.times.shape[0]])
            f32(ch.times); f32(ch.values)
    f32(rig.position); f32(rig.normal); f32(rig.tangent)
    for k in range(K):
        u32(rig.joints[k]); f32(rig.weights[k])
    for t in rig.targets:
        cnt = V if t.indices is None else int(t.indices.shape[0])
        u32([cnt, t.indices is not None, t.normal is not None, t.tangent is not None])
        if t.indices is not None:
            u32(t.indices)
        f32(t.position)
        if t.normal is not None:
            f32(t.normal)
        if t.tangent is not None:
            f32(t.tangent)
    if rig.default_weights is not None:
        f32(rig.default_weights)
    f32(z["tau"]); u32(z["clip_id"])
    if has_mw:
        f32(z["morph_weights"])
    if has_root:
        f32(z["root"])
    # expected (C oracle, fp32): device layouts of skinrt_get_world / skinrt_get_palette
    u32(z["key_index"]); f32(z["world"]); f32(z["palette"]); f32(z["position"]); f32(z["normal"]); f32(z["tangent"]); f32(z["aabb"])
    with open(out_path, "wb") as f:
        f.write(out)
    return len(out)


if __name__ == "__main__":
    print(export(sys.argv[1], sys.argv[2]), "bytes")
