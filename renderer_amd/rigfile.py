"""Flat rig files: a RigData as one `.npz` of plain arrays (no pickles), used for the golden
fixtures and for handing rigs between tools.  The arrays are exactly the fields of
skinrt_rig_desc (include/skinrt.h); ragged parts (channels, morph targets) are stored as
concatenated pools plus offset tables.  Rigs in the long form (several skins / mesh sections, ABI 6) add `skins`,
`skin_joint_node`, `skin_inverse_bind` and `sections`; rigs in the short form are stored exactly as before."""
from __future__ import annotations

from typing import Dict

import numpy as np

from .abi import Channel, MorphTarget, RigData, Section, Skin

_VALUE_WIDTH = {0: 3, 1: 4, 2: 3}  # translation, rotation, scale; weights (path 3): the rig's target count


def rig_to_arrays(rig: RigData, prefix: str = "rig_") -> Dict[str, np.ndarray]:
    a: Dict[str, np.ndarray] = {}
    a["parent"] = rig.parent
    a["translation"], a["rotation"], a["scale"] = rig.translation, rig.rotation, rig.scale
    if rig.matrix is not None:
        a["matrix"], a["has_matrix"] = rig.matrix, rig.has_matrix
    a["joint_node"] = rig.joint_node
    if rig.inverse_bind is not None:
        a["inverse_bind"] = rig.inverse_bind
    if rig.skins is not None:  # one row per skin {joint_count, has_inverse_bind}; pools in skin order
        a["skins"] = np.array([(sk.joint_node.shape[0], sk.inverse_bind is not None) for sk in rig.skins], np.uint32).reshape(-1, 2)
        a["skin_joint_node"] = np.concatenate([sk.joint_node for sk in rig.skins] + [np.zeros(0, np.uint32)])
        a["skin_inverse_bind"] = np.concatenate([sk.inverse_bind.reshape(-1, 16) for sk in rig.skins if sk.inverse_bind is not None]
                                                + [np.zeros((0, 16), np.float32)])
    if rig.sections is not None:  # {node, skin (-1 = none), first_vertex, vertex_count, first_target, target_count}
        a["sections"] = np.array([(sc.node, sc.skin, sc.first_vertex, sc.vertex_count, sc.first_target, sc.target_count)
                                  for sc in rig.sections], np.int64).reshape(-1, 6)
    a["position"], a["normal"], a["tangent"] = rig.position, rig.normal, rig.tangent
    a["joints"] = np.stack(rig.joints) if rig.joints else np.zeros((0, rig.vertex_count, 4), np.uint16)
    a["weights"] = np.stack(rig.weights) if rig.weights else np.zeros((0, rig.vertex_count, 4), np.float32)
    # channels: one row per channel {clip, node, path, interpolation, key_count}; pools in channel order
    rows, times, values = [], [], []
    for ci, clip in enumerate(rig.clips):
        for ch in clip:
            rows.append((ci, ch.target_node, ch.path, ch.interpolation, ch.times.shape[0]))
            times.append(ch.times.reshape(-1))
            values.append(ch.values.reshape(-1))
    a["clip_count"] = np.array([len(rig.clips)], np.uint32)
    a["channels"] = np.array(rows, np.uint32).reshape(-1, 5)
    a["channel_times"] = np.concatenate(times) if times else np.zeros(0, np.float32)
    a["channel_values"] = np.concatenate(values) if values else np.zeros(0, np.float32)
    # morph targets: one row per target {count, compacted, has_normal, has_tangent}
    trow, tidx, tpos, tnrm, ttan = [], [], [], [], []
    for t in rig.targets:
        cnt = int(t.position.shape[0]) if t.indices is None else int(t.indices.shape[0])  # dense: the section's vertex count
        trow.append((cnt, t.indices is not None, t.normal is not None, t.tangent is not None))
        if t.indices is not None: tidx.append(t.indices)
        tpos.append(t.position.reshape(cnt, 3))
        if t.normal is not None: tnrm.append(t.normal.reshape(cnt, 3))
        if t.tangent is not None: ttan.append(t.tangent.reshape(cnt, -1))
    a["targets"] = np.array(trow, np.uint32).reshape(-1, 4)
    a["target_indices"] = np.concatenate(tidx) if tidx else np.zeros(0, np.uint32)
    a["target_position"] = np.concatenate(tpos) if tpos else np.zeros((0, 3), np.float32)
    a["target_normal"] = np.concatenate(tnrm) if tnrm else np.zeros((0, 3), np.float32)
    a["target_tangent"] = np.concatenate(ttan) if ttan else np.zeros((0, rig.tangent_stride), np.float32)
    if rig.default_weights is not None:
        a["default_weights"] = rig.default_weights
    return {prefix + k: v for k, v in a.items()}


def rig_from_arrays(z, prefix: str = "rig_") -> RigData:
    g = lambda k: z[prefix + k] if (prefix + k) in z else None  # noqa: E731
    clips = [[] for _ in range(int(g("clip_count")[0]))]
    to = vo = 0
    times, values = g("channel_times"), g("channel_values")
    n_targets = int(g("targets").shape[0])
    skins = sections = None
    if g("skins") is not None:
        skins, jo, bo = [], 0, 0
        for jc, has_ibm in g("skins").tolist():
            ibm = None
            if has_ibm:
                ibm = g("skin_inverse_bind")[bo:bo + jc].copy(); bo += jc
            skins.append(Skin(g("skin_joint_node")[jo:jo + jc].copy(), ibm)); jo += jc
    if g("sections") is not None:
        sections = [Section(*(int(x) for x in row)) for row in g("sections").tolist()]

    def weights_width(node):  # a weights channel carries the targets of the section on its node
        if sections is None:
            return n_targets
        return next((sc.target_count for sc in sections if sc.node == node and sc.target_count), 0)

    for ci, node, path, interp, kc in g("channels").tolist():
        w = (weights_width(node) if path == 3 else _VALUE_WIDTH[path]) * (3 if interp == 2 else 1)
        clips[ci].append(Channel(node, path, interp, times[to:to + kc].copy(), values[vo:vo + kc * w].reshape(kc, -1).copy()))
        to += kc; vo += kc * w
    targets = []
    io = po = no = tg = 0
    for cnt, compact, has_n, has_t in g("targets").tolist():
        idx = None
        if compact:
            idx = g("target_indices")[io:io + cnt].copy(); io += cnt
        pos = g("target_position")[po:po + cnt].copy(); po += cnt
        nrm = tan = None
        if has_n:
            nrm = g("target_normal")[no:no + cnt].copy(); no += cnt
        if has_t:
            tan = g("target_tangent")[tg:tg + cnt].copy(); tg += cnt
        targets.append(MorphTarget(position=pos, normal=nrm, tangent=tan, indices=idx))
    joints, weights = g("joints"), g("weights")
    return RigData(parent=g("parent"), translation=g("translation"), rotation=g("rotation"), scale=g("scale"),
                   joint_node=g("joint_node"), inverse_bind=g("inverse_bind"), matrix=g("matrix"),
                   has_matrix=g("has_matrix"), position=g("position"), normal=g("normal"), tangent=g("tangent"),
                   joints=[joints[k] for k in range(joints.shape[0])], weights=[weights[k] for k in range(weights.shape[0])],
                   clips=clips, targets=targets, default_weights=g("default_weights"), skins=skins, sections=sections)


def save_rig(path: str, rig: RigData, **extra: np.ndarray) -> None:
    np.savez_compressed(path, **rig_to_arrays(rig), **extra)


def load_rig(path: str) -> RigData:
    with np.load(path, allow_pickle=False) as z:
        return rig_from_arrays(z)
