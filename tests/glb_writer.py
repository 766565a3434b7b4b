"""Test helper: serialise a RigData as a binary glTF (.glb) the way an exporter would — node
indices in an arbitrary (shuffled) order, hierarchy through `children`, one skin, one skinned
mesh primitive with morph targets, one animation per clip — so that the native loader + flattener
(renderer_amd/host) have to rebuild the flat rig: DFS order, joint -> node mapping, channels.
`write_glb_multi` does the same for a rig with several skins and mesh nodes (RigData.skins / .sections): one glTF
skin per skin, one mesh per section on the section's node, un-skinned sections without JOINTS / WEIGHTS / skin.
Written from the glTF 2.0 specification; it is not a general exporter."""
import json
import struct

import numpy as np

F32, U8, U16, U32 = 5126, 5121, 5123, 5125


class _Bin:
    def __init__(self):
        self.data = bytearray()
        self.views = []
        self.accessors = []

    def add(self, arr, ctype, gtype, stride_with=None):
        """one accessor in its own bufferView (4-byte aligned)"""
        arr = np.ascontiguousarray(arr)
        while len(self.data) % 4: self.data.append(0)
        self.views.append({"buffer": 0, "byteOffset": len(self.data), "byteLength": arr.nbytes})
        self.data += arr.tobytes()
        self.accessors.append({"bufferView": len(self.views) - 1, "componentType": ctype, "count": int(arr.shape[0]), "type": gtype})
        return len(self.accessors) - 1

    def add_interleaved(self, a, b):
        """two float VEC3 attributes interleaved in one strided bufferView (byteStride 24)"""
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        inter = np.concatenate([a, b], axis=1)
        while len(self.data) % 4: self.data.append(0)
        off = len(self.data)
        self.data += inter.tobytes()
        self.data += b"\0" * 12  # the reference's bounds check wants a whole stride after the last element
        self.views.append({"buffer": 0, "byteOffset": off, "byteLength": inter.nbytes + 12, "byteStride": 24})
        v = len(self.views) - 1
        self.accessors.append({"bufferView": v, "componentType": F32, "count": int(a.shape[0]), "type": "VEC3"})
        self.accessors.append({"bufferView": v, "byteOffset": 12, "componentType": F32, "count": int(a.shape[0]), "type": "VEC3"})
        return len(self.accessors) - 2, len(self.accessors) - 1


def _f(x):
    return [float(v) for v in np.asarray(x, np.float32).reshape(-1)]


def write_glb(rig, seed=0, joints_u8=False, weights_u16=False, interleave=True, omit_ibm=False, mesh_node=0, omit_target_tangent=False):
    """Returns (glb bytes, perm) where perm[flat node] = glTF node index."""
    rng = np.random.default_rng(seed)
    n = rig.node_count
    perm = rng.permutation(n)
    b = _Bin()
    nodes = [None] * n
    children = [[] for _ in range(n)]
    roots = []
    for k in range(n):
        p = int(rig.parent[k])
        (roots if p < 0 else children[p]).append(int(perm[k]))
    for k in range(n):
        node = {"name": f"n{k}"}
        if rig.has_matrix is not None and rig.has_matrix[k]:
            node["matrix"] = _f(rig.matrix[k])
        else:
            node["translation"] = _f(rig.translation[k]); node["rotation"] = _f(rig.rotation[k]); node["scale"] = _f(rig.scale[k])
        if children[k]: node["children"] = children[k]
        nodes[int(perm[k])] = node
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": roots}], "nodes": nodes}

    skin = {"joints": [int(perm[j]) for j in rig.joint_node]}
    if rig.inverse_bind is not None and not omit_ibm:
        skin["inverseBindMatrices"] = b.add(rig.inverse_bind.reshape(-1, 16), F32, "MAT4")
    doc["skins"] = [skin]

    v = rig.vertex_count
    tan4 = rig.tangent if rig.tangent.shape[1] == 4 else np.concatenate([rig.tangent, np.ones((v, 1), np.float32)], axis=1)
    attrs = {}
    if interleave:
        attrs["POSITION"], attrs["NORMAL"] = b.add_interleaved(rig.position, rig.normal)
    else:
        attrs["POSITION"] = b.add(rig.position, F32, "VEC3"); attrs["NORMAL"] = b.add(rig.normal, F32, "VEC3")
    attrs["TANGENT"] = b.add(tan4.astype(np.float32), F32, "VEC4")
    for k in range(rig.set_count):
        if joints_u8:
            attrs[f"JOINTS_{k}"] = b.add(rig.joints[k].astype(np.uint8), U8, "VEC4")
        else:
            attrs[f"JOINTS_{k}"] = b.add(rig.joints[k].astype(np.uint16), U16, "VEC4")
        if weights_u16:
            q = np.round(np.clip(rig.weights[k], 0, 1) * 65535.0).astype(np.uint16)
            attrs[f"WEIGHTS_{k}"] = b.add(q, U16, "VEC4")
        else:
            attrs[f"WEIGHTS_{k}"] = b.add(rig.weights[k], F32, "VEC4")
    prim = {"attributes": attrs, "mode": 4}
    targets = []
    for t in rig.targets:
        def dense(a, w):
            if a is None: return None
            if t.indices is None: return np.asarray(a, np.float32).reshape(v, -1)[:, :w]
            d = np.zeros((v, w), np.float32); d[t.indices] = np.asarray(a, np.float32).reshape(len(t.indices), -1)[:, :w]
            return d
        tj = {"POSITION": b.add(dense(t.position, 3), F32, "VEC3")}
        dn = dense(t.normal, 3)
        if dn is not None: tj["NORMAL"] = b.add(dn, F32, "VEC3")
        dt = dense(t.tangent, 3)
        # the reference wants VEC4 here and GENERATES tangents when the accessor is absent: always write one
        dt4 = np.zeros((v, 4), np.float32)
        if dt is not None: dt4[:, :3] = dt
        if not omit_target_tangent: tj["TANGENT"] = b.add(dt4, F32, "VEC4")
        targets.append(tj)
    if targets: prim["targets"] = targets
    mesh = {"primitives": [prim]}
    if rig.default_weights is not None and len(rig.targets): mesh["weights"] = _f(rig.default_weights)
    doc["meshes"] = [mesh]
    nodes[int(perm[mesh_node])]["mesh"] = 0
    nodes[int(perm[mesh_node])]["skin"] = 0

    anims = []
    for clip in rig.clips:
        samplers, channels = [], []
        for ch in clip:
            interp = ["LINEAR", "STEP", "CUBICSPLINE"][ch.interpolation]
            if ch.path == 3:  # "weights": M scalars per key (3 M for the spline triple), aimed at the node that carries the mesh
                out = np.asarray(ch.values, np.float32).reshape(-1)
                samplers.append({"input": b.add(np.asarray(ch.times, np.float32), F32, "SCALAR"),
                                 "output": b.add(out, F32, "SCALAR"), "interpolation": interp})
                channels.append({"sampler": len(samplers) - 1, "target": {"node": int(perm[mesh_node]), "path": "weights"}})
                continue
            w = 4 if ch.path == 1 else 3
            out = np.asarray(ch.values, np.float32).reshape(-1, w)  # cubic: (in, value, out) per key = 3 rows per key
            samplers.append({"input": b.add(np.asarray(ch.times, np.float32), F32, "SCALAR"),
                             "output": b.add(out, F32, "VEC4" if w == 4 else "VEC3"), "interpolation": interp})
            channels.append({"sampler": len(samplers) - 1,
                             "target": {"node": int(perm[ch.target_node]), "path": ["translation", "rotation", "scale"][ch.path]}})
        anims.append({"samplers": samplers, "channels": channels})
    if anims: doc["animations"] = anims

    doc["bufferViews"] = b.views
    doc["accessors"] = b.accessors
    while len(b.data) % 4: b.data.append(0)
    doc["buffers"] = [{"byteLength": len(b.data)}]
    js = json.dumps(doc, separators=(",", ":")).encode("utf-8")
    js += b" " * (-len(js) % 4)
    total = 12 + 8 + len(js) + 8 + len(b.data)
    return (struct.pack("<III", 0x46546C67, 2, total) + struct.pack("<II", len(js), 0x4E4F534A) + js +
            struct.pack("<II", len(b.data), 0x004E4942) + bytes(b.data)), perm


def write_glb_multi(rig, seed=0):
    """A rig in the long form (skins + sections) as .glb.  Returns (glb bytes, perm) as write_glb does."""
    rng = np.random.default_rng(seed)
    n = rig.node_count
    perm = rng.permutation(n)
    b = _Bin()
    nodes = [None] * n
    children = [[] for _ in range(n)]
    roots = []
    for k in range(n):
        p = int(rig.parent[k])
        (roots if p < 0 else children[p]).append(int(perm[k]))
    for k in range(n):
        node = {"name": f"n{k}"}
        if rig.has_matrix is not None and rig.has_matrix[k]:
            node["matrix"] = _f(rig.matrix[k])
        else:
            node["translation"] = _f(rig.translation[k]); node["rotation"] = _f(rig.rotation[k]); node["scale"] = _f(rig.scale[k])
        if children[k]: node["children"] = children[k]
        nodes[int(perm[k])] = node
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": roots}], "nodes": nodes}
    doc["skins"] = []
    for sk in rig.skin_list():
        skin = {"joints": [int(perm[j]) for j in sk.joint_node]}
        if sk.inverse_bind is not None:
            skin["inverseBindMatrices"] = b.add(sk.inverse_bind.reshape(-1, 16), F32, "MAT4")
        doc["skins"].append(skin)
    tan4 = rig.tangent if rig.tangent.shape[1] == 4 else np.concatenate([rig.tangent, np.ones((rig.vertex_count, 1), np.float32)], axis=1)
    doc["meshes"] = []
    for sc in rig.section_list():
        lo, hi = sc.first_vertex, sc.first_vertex + sc.vertex_count
        v = sc.vertex_count
        attrs = {}
        attrs["POSITION"], attrs["NORMAL"] = b.add_interleaved(rig.position[lo:hi], rig.normal[lo:hi])
        attrs["TANGENT"] = b.add(tan4[lo:hi].astype(np.float32), F32, "VEC4")
        if sc.skin >= 0:
            for k in range(rig.set_count):
                attrs[f"JOINTS_{k}"] = b.add(rig.joints[k][lo:hi].astype(np.uint16), U16, "VEC4")
                attrs[f"WEIGHTS_{k}"] = b.add(rig.weights[k][lo:hi], F32, "VEC4")
        prim = {"attributes": attrs, "mode": 4}
        targets = []
        for t in rig.targets[sc.first_target:sc.first_target + sc.target_count]:
            def dense(a, w):
                if a is None: return None
                if t.indices is None: return np.asarray(a, np.float32).reshape(v, -1)[:, :w]
                d = np.zeros((v, w), np.float32); d[t.indices] = np.asarray(a, np.float32).reshape(len(t.indices), -1)[:, :w]
                return d
            tj = {"POSITION": b.add(dense(t.position, 3), F32, "VEC3")}
            dn = dense(t.normal, 3)
            if dn is not None: tj["NORMAL"] = b.add(dn, F32, "VEC3")
            dt4 = np.zeros((v, 4), np.float32)
            dt = dense(t.tangent, 3)
            if dt is not None: dt4[:, :3] = dt
            tj["TANGENT"] = b.add(dt4, F32, "VEC4")
            targets.append(tj)
        if targets: prim["targets"] = targets
        mesh = {"primitives": [prim]}
        if rig.default_weights is not None and sc.target_count:
            mesh["weights"] = _f(rig.default_weights[sc.first_target:sc.first_target + sc.target_count])
        doc["meshes"].append(mesh)
        nodes[int(perm[sc.node])]["mesh"] = len(doc["meshes"]) - 1
        if sc.skin >= 0: nodes[int(perm[sc.node])]["skin"] = int(sc.skin)
    anims = []
    for clip in rig.clips:
        samplers, channels = [], []
        for ch in clip:
            interp = ["LINEAR", "STEP", "CUBICSPLINE"][ch.interpolation]
            if ch.path == 3:
                out, gtype, path = np.asarray(ch.values, np.float32).reshape(-1), "SCALAR", "weights"
            else:
                w = 4 if ch.path == 1 else 3
                out, gtype, path = np.asarray(ch.values, np.float32).reshape(-1, w), ("VEC4" if w == 4 else "VEC3"), ["translation", "rotation", "scale"][ch.path]
            samplers.append({"input": b.add(np.asarray(ch.times, np.float32), F32, "SCALAR"), "output": b.add(out, F32, gtype), "interpolation": interp})
            channels.append({"sampler": len(samplers) - 1, "target": {"node": int(perm[ch.target_node]), "path": path}})
        anims.append({"samplers": samplers, "channels": channels})
    if anims: doc["animations"] = anims
    doc["bufferViews"] = b.views
    doc["accessors"] = b.accessors
    while len(b.data) % 4: b.data.append(0)
    doc["buffers"] = [{"byteLength": len(b.data)}]
    js = json.dumps(doc, separators=(",", ":")).encode("utf-8")
    js += b" " * (-len(js) % 4)
    total = 12 + 8 + len(js) + 8 + len(b.data)
    return (struct.pack("<III", 0x46546C67, 2, total) + struct.pack("<II", len(js), 0x4E4F534A) + js +
            struct.pack("<II", len(b.data), 0x004E4942) + bytes(b.data)), perm
