"""Seeded synthetic rigs and crowds in the shapes of BASELINE.json's five configs
(SURVEY.md §8d).  No .glb exists on disk, so C1/C2 are CesiumMan-/Fox-SHAPED
stand-ins (node/joint/vertex/channel counts as recalled from the Khronos listings, not
verifiable offline).  Counter-based RNG: numpy Philox keyed by the seed.

What the generator produces is the loader OUTPUT (renderer-asset types flattened, see
renderer_amd/abi.py:RigData); it exercises every feature the path has: DFS pre-order trees,
joint order != node order, per-channel key counts 1..Kf, late first keys / early last keys
(clamping), LINEAR / STEP / CUBICSPLINE, K weight sets, dense and compacted morph targets.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np

from .abi import (INTERP_CUBICSPLINE, INTERP_LINEAR, INTERP_STEP, PATH_ROTATION, PATH_SCALE,
                  PATH_TRANSLATION, PATH_WEIGHTS, Channel, MorphTarget, RigData)


def _rng(seed: int, stream: int = 0) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFFFFFFFFFF, stream]))


def _quat_near_identity(rng, n, max_deg=30.0):
    axis = rng.normal(size=(n, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True) + 1e-12
    ang = np.deg2rad(rng.uniform(-max_deg, max_deg, size=(n, 1)))
    q = np.concatenate([axis * np.sin(ang / 2), np.cos(ang / 2)], axis=1)
    return q.astype(np.float32)


def _trs_mat64(t, q, s):
    x, y, z, w = [q[..., i].astype(np.float64) for i in range(4)]
    m = np.zeros(q.shape[:-1] + (4, 4))
    m[..., 0, 0] = 1 - 2 * (y * y + z * z); m[..., 0, 1] = 2 * (x * y - w * z); m[..., 0, 2] = 2 * (x * z + w * y)
    m[..., 1, 0] = 2 * (x * y + w * z); m[..., 1, 1] = 1 - 2 * (x * x + z * z); m[..., 1, 2] = 2 * (y * z - w * x)
    m[..., 2, 0] = 2 * (x * z - w * y); m[..., 2, 1] = 2 * (y * z + w * x); m[..., 2, 2] = 1 - 2 * (x * x + y * y)
    m[..., :3, :3] *= s[..., None, :].astype(np.float64)
    m[..., :3, 3] = t
    m[..., 3, 3] = 1
    return m


def _tree_parents(rng, n, topology, max_depth):
    """parent[] in DFS pre-order (parent[k] < k): each new node attaches to a node on the
    right-most path of the tree built so far, which keeps pre-order valid."""
    parent = np.full(n, -1, dtype=np.int32)
    depth = np.zeros(n, dtype=np.int32)
    if topology == "chain":
        parent[1:] = np.arange(n - 1)
        return parent
    path = [0]
    for k in range(1, n):
        # bias towards deepening so depth approaches max_depth, sometimes pop back up
        while len(path) > 1 and (depth[path[-1]] + 1 >= max_depth or rng.random() < 0.35):
            path.pop()
        p = path[-1]
        parent[k] = p
        depth[k] = depth[p] + 1
        path.append(k)
    return parent


@dataclass
class SynthSpec:
    name: str
    seed: int
    nodes: int
    joints: int
    verts: int
    sets: int = 1
    targets: int = 0
    target_mode: str = "dense"     # "dense" | "sparse"
    target_frac: float = 0.10
    clips: int = 1
    keys: int = 48
    clip_len: float = 2.0
    topology: str = "tree"
    max_depth: int = 12
    frac_step: float = 0.0
    frac_cubic: float = 0.0
    scale_jitter: float = 0.1
    matrix_nodes: int = 0          # number of static nodes given as Matrix transforms
    joint_spread: int = 3          # influences drawn from primary bone +- spread (joint index space)
    scatter_joints: bool = False   # True: influences uniform over all joints (worst-case locality)
    instances: int = 1
    morph_zero_frac: float = 0.75
    weight_channels: int = 0       # clips 0 .. weight_channels-1 carry a glTF "weights" channel (interpolation cycles LINEAR, CUBICSPLINE, STEP)


CONFIGS: Dict[str, SynthSpec] = {
    # C1 CesiumMan-shaped: N=22, J=19, V=3273, 57 LINEAR channels x 48 keys over 2.0 s, 1 instance
    "c1": SynthSpec("c1", 0x5EED0001 + 1, nodes=22, joints=19, verts=3273, keys=48, max_depth=7, instances=1),
    # C2 Fox-shaped: N=25, J=24, V=1728, 3 clips, 4096 instances
    "c2": SynthSpec("c2", 0x5EED0001 + 2, nodes=25, joints=24, verts=1728, clips=3, keys=24, max_depth=8, instances=4096),
    # C3 VRM-shaped: N=260, J=200, V=60000, 32 dense morph targets, 10k instances
    "c3": SynthSpec("c3", 0x5EED0001 + 3, nodes=260, joints=200, verts=60000, targets=32, keys=48,
                    frac_step=0.1, frac_cubic=0.1, instances=10000),
    # C4 = C3 rig x 100k instances sharded over the node
    "c4": SynthSpec("c4", 0x5EED0001 + 3, nodes=260, joints=200, verts=60000, targets=32, keys=48,
                    frac_step=0.1, frac_cubic=0.1, instances=100000),
    # C5 stress: 4096-joint chain, 8 weights, 128 sparse targets (1 % of verts each), 1M verts, 1k instances
    "c5": SynthSpec("c5", 0x5EED0001 + 5, nodes=4096, joints=4096, verts=1000000, sets=2, targets=128,
                    target_mode="sparse", target_frac=0.01, keys=16, topology="chain", frac_step=0.1,
                    frac_cubic=0.1, scale_jitter=0.002, instances=1000),
}


def scaled(spec: SynthSpec, **kw) -> SynthSpec:
    d = dict(spec.__dict__)
    d.update(kw)
    return SynthSpec(**d)


def make_rig(spec: SynthSpec) -> RigData:
    rng = _rng(spec.seed, 0)
    N, J, V = spec.nodes, spec.joints, spec.verts
    parent = _tree_parents(rng, N, spec.topology, spec.max_depth)

    # rest pose
    t = rng.uniform(-0.1, 0.1, size=(N, 3)).astype(np.float32)
    q = _quat_near_identity(rng, N)
    s = rng.uniform(1 - spec.scale_jitter, 1 + spec.scale_jitter, size=(N, 3)).astype(np.float32)
    if spec.topology == "chain":
        t[:, 1] = np.abs(t[:, 1]) + 0.05  # a chain that grows "up"

    # joints: a subset of nodes in shuffled order (joint order != node order)
    if J == N:
        joint_node = np.arange(N, dtype=np.uint32)
        if spec.topology != "chain":
            joint_node = rng.permutation(N).astype(np.uint32)
    else:
        cand = np.arange(1, N) if N > J else np.arange(N)
        joint_node = rng.permutation(cand)[:J].astype(np.uint32)
        # keep index-space locality meaningful: neighbouring joint ids are neighbouring nodes,
        # then swap a few to make the map a real permutation
        joint_node.sort()
        for _ in range(max(1, J // 8)):
            a, b = rng.integers(0, J, size=2)
            joint_node[a], joint_node[b] = joint_node[b], joint_node[a]

    # static Matrix-form nodes (never animated here; the animated-matrix case is unit-tested)
    local64 = _trs_mat64(t, q, s)
    matrix = None
    has_matrix = None
    if spec.matrix_nodes:
        non_joint = np.setdiff1d(np.arange(N), joint_node)
        pick = non_joint[: spec.matrix_nodes]
        matrix = np.zeros((N, 16), dtype=np.float32)
        has_matrix = np.zeros(N, dtype=np.uint8)
        for n in pick:
            matrix[n] = local64[n].T.reshape(16).astype(np.float32)  # column-major
            has_matrix[n] = 1
            local64[n] = matrix[n].reshape(4, 4).T.astype(np.float64)

    # rest world (fp64) -> inverse bind matrices
    world = np.zeros((N, 4, 4))
    for n in range(N):
        world[n] = local64[n] if parent[n] < 0 else world[parent[n]] @ local64[n]
    ibm = np.linalg.inv(world[joint_node])
    inverse_bind = np.ascontiguousarray(ibm.transpose(0, 2, 1)).reshape(J, 16).astype(np.float32)

    # mesh: vertices in bind space scattered around their primary bone's rest position
    rv = _rng(spec.seed, 1)
    primary = np.minimum((np.arange(V, dtype=np.int64) * J) // max(V, 1), J - 1)
    bone_pos = world[joint_node][:, :3, 3]
    position = (bone_pos[primary] + rv.uniform(-1, 1, size=(V, 3)) * 0.25).astype(np.float32)
    normal = rv.normal(size=(V, 3)); normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    tangent = rv.normal(size=(V, 4)); tangent[:, :3] /= np.linalg.norm(tangent[:, :3], axis=1, keepdims=True)
    tangent[:, 3] = np.where(rv.random(V) < 0.5, 1.0, -1.0)
    K = spec.sets
    nw = 4 * K
    if spec.scatter_joints:
        jidx = rv.integers(0, J, size=(V, nw))
    else:
        off = rv.integers(-spec.joint_spread, spec.joint_spread + 1, size=(V, nw))
        off[:, 0] = 0
        jidx = np.clip(primary[:, None] + off, 0, J - 1)
    w = rv.exponential(size=(V, nw))
    n_active = rv.integers(1, nw + 1, size=V)
    w[np.arange(nw)[None, :] >= n_active[:, None]] = 0.0
    w = -np.sort(-w, axis=1)
    w = (w / w.sum(axis=1, keepdims=True)).astype(np.float32)
    joints = [np.ascontiguousarray(jidx[:, 4 * k:4 * k + 4]).astype(np.uint16) for k in range(K)]
    weights = [np.ascontiguousarray(w[:, 4 * k:4 * k + 4]) for k in range(K)]

    # animation clips: T/R/S channel per joint node
    ra = _rng(spec.seed, 2)
    clips = []
    for c in range(spec.clips):
        length = spec.clip_len * (1.0 + 0.25 * c)
        chans = []
        for j in range(J):
            node = int(joint_node[j])
            for path in (PATH_TRANSLATION, PATH_ROTATION, PATH_SCALE):
                r = ra.random()
                interp = INTERP_STEP if r < spec.frac_step else (
                    INTERP_CUBICSPLINE if r < spec.frac_step + spec.frac_cubic else INTERP_LINEAR)
                r2 = ra.random()
                kf = 1 if r2 < 0.03 else (2 if r2 < 0.06 else int(ra.integers(max(2, spec.keys // 2), spec.keys + 1)))
                times = np.sort(ra.uniform(0, length, size=kf)).astype(np.float32)
                if kf > 1:
                    if ra.random() < 0.8: times[0] = 0.0
                    if ra.random() < 0.8: times[-1] = np.float32(length)
                    # strictly increasing keys (duplicate times are legal input but make td = 0)
                    for k in range(1, kf):
                        if times[k] <= times[k - 1]:
                            times[k] = np.nextafter(times[k - 1], np.float32(np.inf))
                wdt = 4 if path == PATH_ROTATION else 3
                if path == PATH_TRANSLATION:
                    base = t[node] + ra.uniform(-0.05, 0.05, size=(kf, 3))
                elif path == PATH_ROTATION:
                    base = _quat_near_identity(ra, kf)
                else:
                    base = s[node] + ra.uniform(-spec.scale_jitter, spec.scale_jitter, size=(kf, 3)) * 0.5
                base = base.astype(np.float32)
                if interp == INTERP_CUBICSPLINE:
                    vals = np.zeros((kf, 3, wdt), dtype=np.float32)
                    vals[:, 1] = base
                    vals[:, 0] = ra.normal(scale=0.05, size=(kf, wdt))
                    vals[:, 2] = ra.normal(scale=0.05, size=(kf, wdt))
                else:
                    vals = base
                chans.append(Channel(node, path, interp, times, vals))
        # file order of channels is arbitrary in glTF: shuffle
        order = ra.permutation(len(chans))
        clips.append([chans[i] for i in order])
    # glTF "weights" channels (animated expressions): own stream, so rigs without them are unchanged.  Keys are
    # exactly zero for most targets most of the time, as expressions are.
    if spec.weight_channels and spec.targets:
        rw = _rng(spec.seed, 4)
        M = spec.targets
        for c in range(min(spec.weight_channels, len(clips))):
            length = spec.clip_len * (1.0 + 0.25 * c)
            interp = (INTERP_LINEAR, INTERP_CUBICSPLINE, INTERP_STEP)[c % 3]
            kf = max(2, spec.keys // 2)
            times = np.sort(rw.uniform(0, length, size=kf)).astype(np.float32)
            times[0] = np.float32(0.1 * length)   # late first key, early last key: both clamps are exercised
            times[-1] = np.float32(0.9 * length)
            times = np.sort(times)
            for k in range(1, kf):
                if times[k] <= times[k - 1]:
                    times[k] = np.nextafter(times[k - 1], np.float32(np.inf))
            base = rw.uniform(0, 1, size=(kf, M)).astype(np.float32)
            on = rw.random((kf, M)) < 0.5
            on[:, rw.random(M) < 0.6] = False      # most targets stay off for the whole clip
            base[~on] = 0.0
            if interp == INTERP_CUBICSPLINE:
                vals = np.zeros((kf, 3, M), dtype=np.float32)
                vals[:, 1] = base
                tang = rw.normal(scale=0.05, size=(2, kf, M)).astype(np.float32)
                tang[:, ~(on | np.roll(on, 1, axis=0) | np.roll(on, -1, axis=0))] = 0.0  # off segments stay exactly zero
                vals[:, 0] = tang[0]
                vals[:, 2] = tang[1]
            else:
                vals = base
            node = int(joint_node[0])
            clips[c].insert(int(rw.integers(0, len(clips[c]) + 1)), Channel(node, PATH_WEIGHTS, interp, times, vals))

    # morph targets
    rm = _rng(spec.seed, 3)
    targets = []
    M = spec.targets
    for m in range(M):
        cnt = max(1, int(V * spec.target_frac))
        if spec.target_mode == "dense":
            start = int(rm.integers(0, max(1, V - cnt)))
            dp = np.zeros((V, 3), np.float32); dn = np.zeros((V, 3), np.float32); dt = np.zeros((V, 4), np.float32)
            dp[start:start + cnt] = rm.normal(scale=0.02, size=(cnt, 3))
            dn[start:start + cnt] = rm.normal(scale=0.05, size=(cnt, 3))
            dt[start:start + cnt, :3] = rm.normal(scale=0.05, size=(cnt, 3))
            # the loader leaves normal/tangent empty when the file has none (loader/gltf/mod.rs:895-916)
            targets.append(MorphTarget(position=dp, normal=dn if m % 4 != 3 else None, tangent=dt if m % 4 < 2 else None))
        else:
            run = 64
            nruns = max(1, cnt // run)
            starts = np.sort(rm.choice(max(1, V // run), size=min(nruns, max(1, V // run)), replace=False)) * run
            idx = (starts[:, None] + np.arange(run)[None, :]).reshape(-1)
            idx = idx[idx < V].astype(np.uint32)
            c = idx.shape[0]
            targets.append(MorphTarget(position=rm.normal(scale=0.02, size=(c, 3)).astype(np.float32),
                                       normal=rm.normal(scale=0.05, size=(c, 3)).astype(np.float32) if m % 4 != 3 else None,
                                       tangent=rm.normal(scale=0.05, size=(c, 4)).astype(np.float32) if m % 4 < 2 else None,
                                       indices=idx))
    default_weights = rm.uniform(0, 1, size=M).astype(np.float32) if M else None

    return RigData(parent=parent, translation=t, rotation=q, scale=s, joint_node=joint_node,
                   inverse_bind=inverse_bind, matrix=matrix, has_matrix=has_matrix,
                   position=position, normal=normal.astype(np.float32), tangent=tangent.astype(np.float32),
                   joints=joints, weights=weights, clips=clips, targets=targets,
                   default_weights=default_weights, name=spec.name)


@dataclass
class CrowdState:
    tau: np.ndarray                         # [I] f32
    clip_id: np.ndarray                     # [I] u32
    morph_weights: Optional[np.ndarray]     # [I][M] f32 or None
    root: Optional[np.ndarray] = None       # [I][16] f32 or None


def make_crowd(spec: SynthSpec, rig: RigData, instances: Optional[int] = None, first: int = 0,
               with_root: bool = False) -> CrowdState:
    """Per-instance clocks (independent, uniform over the clip), clip choice and morph weights.
    `first` offsets the instance ids so shards of one crowd draw disjoint, reproducible streams."""
    n = spec.instances if instances is None else instances
    # one Philox stream per 65536-instance block keeps shards independent of the shard size
    tau = np.zeros(n, np.float32); clip = np.zeros(n, np.uint32)
    M = rig.target_count
    mw = np.zeros((n, M), np.float32) if M else None
    root = np.zeros((n, 16), np.float32) if with_root else None
    lens = np.array([rig.clip_length(c) for c in range(len(rig.clips))], dtype=np.float64) if rig.clips else None
    blk = 65536
    g0 = first
    while g0 < first + n:
        b = g0 // blk
        r = _rng(spec.seed, 16 + b)
        u = r.random(blk); cc = r.integers(0, max(1, len(rig.clips)), size=blk)
        wv = r.random((blk, max(M, 1))); wz = r.random((blk, max(M, 1)))
        rr = r.uniform(-50, 50, size=(blk, 3)); ra = r.uniform(-np.pi, np.pi, size=blk)
        lo = g0 - b * blk
        hi = min(blk, first + n - b * blk)
        sl = slice(g0 - first, g0 - first + (hi - lo))
        if rig.clips:
            clip[sl] = cc[lo:hi]
            tau[sl] = (u[lo:hi] * lens[cc[lo:hi]]).astype(np.float32)
        if M:
            w = wv[lo:hi, :M].copy()
            w[wz[lo:hi, :M] < spec.morph_zero_frac] = 0.0
            mw[sl] = w
        if with_root:
            c, s_ = np.cos(ra[lo:hi]), np.sin(ra[lo:hi])
            m = np.zeros((hi - lo, 4, 4))
            m[:, 0, 0] = c; m[:, 0, 2] = s_; m[:, 2, 0] = -s_; m[:, 2, 2] = c; m[:, 1, 1] = 1; m[:, 3, 3] = 1
            m[:, :3, 3] = rr[lo:hi]
            root[sl] = m.transpose(0, 2, 1).reshape(-1, 16)
        g0 = b * blk + hi
    return CrowdState(tau, clip, mw, root)


def tiny_spec(seed: int = 7, **kw) -> SynthSpec:
    """Small rig for fixtures and fast tests (<= 64 verts, <= 8 joints by default)."""
    base = SynthSpec("tiny", seed, nodes=10, joints=8, verts=64, sets=1, targets=3, clips=2, keys=3,
                     max_depth=4, frac_step=0.2, frac_cubic=0.3, instances=5)
    return scaled(base, **kw)


def make_multi_rig(base: RigData, seed: int = 11, fractions=(0.5, 0.3, 0.2), rigid_frac: float = 0.06, rigid_sections: int = 1):
    """An avatar-shaped rig with SEVERAL skins on one node tree plus un-skinned props, cut out of a one-skin rig: the
    vertices are dealt to len(fractions) skinned sections (body / face / hair ...) and `rigid_sections` un-skinned ones.
    Each skinned section gets a skin of its own — the joints its vertices use plus a few shared ones, in an order of its
    own, so equal joints sit at different indices in different skins — and its vertex joint indices are rewritten to that
    skin.  Every base morph target becomes one target per section it touches; a base `weights` channel becomes one
    channel per morphed section (on that section's node).

    Returns (rig, target_map): target_map[t] = the base target behind target t, so a base crowd's weights carry over as
    mw[:, target_map].  Skinned sections then reproduce the base rig's results bit for bit (same joint matrices, same
    weights, same target order) — a check that needs no oracle; un-skinned sections follow shader.wgsl:143-152."""
    from .abi import Section, Skin
    rng = _rng(seed, 9)
    V, J, N = base.vertex_count, base.joint_count, base.node_count
    K = base.set_count
    n_rigid = int(V * rigid_frac) if rigid_sections else 0
    n_skinned = V - n_rigid
    cuts = [0]
    for f in fractions[:-1]:
        cuts.append(min(n_skinned, cuts[-1] + max(1, int(n_skinned * f))))
    cuts.append(n_skinned)
    for r in range(rigid_sections):
        cuts.append(n_skinned + (n_rigid * (r + 1)) // rigid_sections)
    n_sk = len(fractions)
    # one mesh per node: distinct nodes for the sections; skinned ones prefer non-joint nodes, props hang off joints (they move)
    joint_nodes = [int(x) for x in base.joint_node]
    free = [n for n in range(N) if n not in set(joint_nodes)] + joint_nodes
    nodes = free[:n_sk] + [joint_nodes[(3 + 5 * r) % J] for r in range(rigid_sections)]
    if len(set(nodes)) != len(nodes):
        nodes = list(dict.fromkeys(nodes + list(range(N))))[: n_sk + rigid_sections]
    joints = [a.copy() for a in base.joints]
    weights = [a.copy() for a in base.weights]
    skins, sections = [], []
    for s in range(n_sk + rigid_sections):
        v0, v1 = cuts[s], cuts[s + 1]
        if s < n_sk:
            used = np.unique(np.concatenate([joints[k][v0:v1].reshape(-1) for k in range(K)])) if v1 > v0 else np.zeros(0, np.int64)
            extra = rng.integers(0, J, size=min(J, 3))
            order = rng.permutation(np.unique(np.concatenate([used.astype(np.int64), extra])))
            local = np.full(J, 0xFFFF, np.int64)
            local[order] = np.arange(order.shape[0])
            for k in range(K):
                joints[k][v0:v1] = local[joints[k][v0:v1].astype(np.int64)].astype(np.uint16)
            skins.append(Skin(base.joint_node[order].copy(), None if base.inverse_bind is None else base.inverse_bind[order].copy()))
            sections.append(Section(nodes[s], s, v0, v1 - v0))
        else:
            for k in range(K):  # not read for an un-skinned section: poison them
                joints[k][v0:v1] = 0xFFFF
                weights[k][v0:v1] = np.float32(np.nan)
            sections.append(Section(nodes[s], -1, v0, v1 - v0))
    # morph targets: one per (section, base target that touches the section), base order inside a section
    targets, target_map = [], []
    for sc in sections:
        sc.first_target = len(targets)
        v0, v1 = sc.first_vertex, sc.first_vertex + sc.vertex_count
        for m, t in enumerate(base.targets):
            if t.indices is None:
                sl = slice(v0, v1)
                parts = [a[sl] for a in (t.position, t.normal, t.tangent) if a is not None]
                if not any(p.any() for p in parts):
                    continue
                targets.append(MorphTarget(position=None if t.position is None else t.position[sl].copy(),
                                           normal=None if t.normal is None else t.normal[sl].copy(),
                                           tangent=None if t.tangent is None else t.tangent[sl].copy()))
            else:
                sel = (t.indices >= v0) & (t.indices < v1)
                if not sel.any():
                    continue
                targets.append(MorphTarget(position=None if t.position is None else t.position[sel].copy(),
                                           normal=None if t.normal is None else t.normal[sel].copy(),
                                           tangent=None if t.tangent is None else t.tangent[sel].copy(),
                                           indices=(t.indices[sel] - v0).astype(np.uint32)))
            target_map.append(m)
        sc.target_count = len(targets) - sc.first_target
    target_map = np.array(target_map, np.int64)
    clips = []
    for clip in base.clips:
        out = []
        for ch in clip:
            if ch.path != PATH_WEIGHTS:
                out.append(ch)
                continue
            for sc in sections:
                if not sc.target_count:
                    continue
                cols = target_map[sc.first_target:sc.first_target + sc.target_count]
                out.append(Channel(sc.node, PATH_WEIGHTS, ch.interpolation, ch.times.copy(),
                                   np.ascontiguousarray(ch.values.reshape(ch.times.shape[0], -1, base.target_count)[..., cols])))
        clips.append(out)
    dw = None if base.default_weights is None or not len(target_map) else base.default_weights[target_map].copy()
    rig = RigData(parent=base.parent, translation=base.translation, rotation=base.rotation, scale=base.scale,
                  joint_node=skins[0].joint_node, inverse_bind=skins[0].inverse_bind, matrix=base.matrix, has_matrix=base.has_matrix,
                  position=base.position, normal=base.normal, tangent=base.tangent, joints=joints, weights=weights,
                  clips=clips, targets=targets, default_weights=dw, name=base.name + "-multi", skins=skins, sections=sections)
    return rig, target_map
