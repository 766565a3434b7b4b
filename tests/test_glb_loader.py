"""Native .glb loader + scene flattener (renderer_amd/host).  A synthetic rig
is serialised as a binary glTF with shuffled node indices (tests/glb_writer.py), loaded back by
the C++ loader and flattened; the flat rig must equal the original — node ORDER bit-exact (DFS
pre-order rebuilt from `children`), joints, keyframes, vertex data — and evaluate identically.
The loader follows the reference's conventions (renderer-asset/src/loader/gltf/mod.rs); its
quirks are tested one by one below.  PARITY UNPINNED: the reference loader cannot run here."""
import json
import struct

import numpy as np
import pytest

from renderer_amd import abi, api, hostlib, synth
from tests import helpers
from tests.glb_writer import write_glb


def dense_target(rig, t, attr, w):
    a = getattr(t, attr)
    v = rig.vertex_count
    if a is None:
        return np.zeros((v, w), np.float32)
    if t.indices is None:
        return np.asarray(a, np.float32).reshape(v, -1)[:, :w]
    d = np.zeros((v, w), np.float32)
    d[t.indices] = np.asarray(a, np.float32).reshape(len(t.indices), -1)[:, :w]
    return d


def assert_same_rig(a, b, weights_quantised=False):
    assert np.array_equal(a.parent, b.parent)
    has_a = a.has_matrix if a.has_matrix is not None else np.zeros(a.node_count, np.uint8)
    has_b = b.has_matrix if b.has_matrix is not None else np.zeros(b.node_count, np.uint8)
    assert np.array_equal(has_a, has_b)
    trs = has_a == 0
    for k in ("translation", "rotation", "scale"):
        assert np.array_equal(getattr(a, k)[trs].view(np.uint32), getattr(b, k)[trs].view(np.uint32)), k
    if has_a.any():
        assert np.array_equal(a.matrix[has_a != 0], b.matrix[has_b != 0])
    assert np.array_equal(a.joint_node, b.joint_node)
    if a.inverse_bind is not None:
        assert np.array_equal(a.inverse_bind, b.inverse_bind)
    assert np.array_equal(a.position, b.position) and np.array_equal(a.normal, b.normal)
    assert np.array_equal(a.tangent[:, :3], b.tangent[:, :3])
    assert a.set_count == b.set_count
    for k in range(a.set_count):
        assert np.array_equal(a.joints[k], b.joints[k])
        if weights_quantised:
            q = np.round(np.clip(a.weights[k], 0, 1) * 65535.0).astype(np.uint16)
            assert np.array_equal((q.astype(np.float32) / np.float32(65535.0)), b.weights[k])  # u16 / 65535 (loader/gltf/mod.rs:343-347)
        else:
            assert np.array_equal(a.weights[k], b.weights[k])
    assert len(a.clips) == len(b.clips)
    for ca, cb in zip(a.clips, b.clips):
        assert len(ca) == len(cb)
        for x, y in zip(ca, cb):
            assert (x.target_node, x.path, x.interpolation) == (y.target_node, y.path, y.interpolation)
            assert np.array_equal(x.times, y.times)
            assert np.array_equal(np.asarray(x.values).reshape(-1), np.asarray(y.values).reshape(-1))
    assert a.target_count == b.target_count
    for ta, tb in zip(a.targets, b.targets):
        assert np.array_equal(dense_target(a, ta, "position", 3), dense_target(b, tb, "position", 3))
        assert np.array_equal(dense_target(a, ta, "normal", 3), dense_target(b, tb, "normal", 3))
        assert np.array_equal(dense_target(a, ta, "tangent", 3), dense_target(b, tb, "tangent", 3))


SPECS = {
    "tiny": synth.tiny_spec(seed=51, matrix_nodes=2),
    "two_sets_sparse_targets": synth.tiny_spec(seed=52, sets=2, target_mode="sparse", target_frac=0.3, targets=4),
    "c1_shape": synth.CONFIGS["c1"],
    "c3_shape": synth.scaled(synth.CONFIGS["c3"], verts=1200, targets=6, keys=5),
}


@pytest.mark.parametrize("name", list(SPECS))
def test_round_trip_through_glb(name, oracle32):
    spec = SPECS[name]
    rig = synth.make_rig(spec)
    glb, perm = write_glb(rig, seed=7)
    loaded, order, lengths = hostlib.load_glb(glb)
    # node ordering: flat node k is glTF node perm[k] — the flattener rebuilt the DFS pre-order
    assert np.array_equal(order, perm.astype(np.uint32))
    assert_same_rig(rig, loaded)
    for c in range(len(rig.clips)):
        assert lengths[c] == np.float32(rig.clip_length(c))
    info = api.validate_rig(loaded)
    assert (info.node_count, info.joint_count, info.vertex_count) == (rig.node_count, rig.joint_count, rig.vertex_count)
    crowd = synth.make_crowd(spec, rig, instances=3)
    a = helpers.run_oracle(oracle32, rig, crowd)
    b = helpers.run_oracle(oracle32, loaded, crowd)
    for k in a:
        helpers.assert_equal(np.asarray(a[k]).view(np.uint32), np.asarray(b[k]).view(np.uint32), k)


def test_u8_joints_and_u16_weights_follow_the_reference_constants():
    spec = synth.tiny_spec(seed=53)
    rig = synth.make_rig(spec)
    glb, _ = write_glb(rig, seed=1, joints_u8=True, weights_u16=True, interleave=False)
    loaded, _, _ = hostlib.load_glb(glb)
    assert loaded.joints[0].dtype == np.uint16
    assert_same_rig(rig, loaded, weights_quantised=True)


def test_missing_inverse_bind_matrices_mean_identity():
    spec = synth.tiny_spec(seed=54)
    rig = synth.make_rig(spec)
    glb, _ = write_glb(rig, seed=2, omit_ibm=True)
    loaded, _, _ = hostlib.load_glb(glb)
    assert loaded.inverse_bind is None


def _edit(glb, fn):
    """re-pack a GLB after editing its JSON document"""
    jl = struct.unpack_from("<I", glb, 12)[0]
    doc = json.loads(glb[20:20 + jl].decode())
    rest = glb[20 + jl:]
    fn(doc)
    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * (-len(js) % 4)
    return struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(js) + len(rest)) + struct.pack("<II", len(js), 0x4E4F534A) + js + rest


def _expect(glb, code, needle):
    with pytest.raises(hostlib.HostError) as e:
        hostlib.load_glb(glb)
    assert e.value.code == code and needle in e.value.message, e.value


def test_errors_where_the_reference_panics():
    spec = synth.tiny_spec(seed=55)
    rig = synth.make_rig(spec)
    glb, _ = write_glb(rig, seed=3)
    _expect(b"nope" + glb[4:], abi.E_INVALID, "magic")
    _expect(_edit(glb, lambda d: d["meshes"][0]["primitives"][0]["attributes"].pop("POSITION")), abi.E_INVALID, "No positions")
    # a "weights" channel whose output is not SCALAR, or not keys x targets long
    _expect(_edit(glb, lambda d: d["animations"][0]["channels"][0]["target"].update(path="weights")), abi.E_INVALID, "BadAccessorDimensions (animation weights)")
    _expect(_edit(glb, lambda d: d["accessors"][d["meshes"][0]["primitives"][0]["attributes"]["WEIGHTS_0"]].update(componentType=5122)),
            abi.E_INVALID, "signed normalised")
    _expect(_edit(glb, lambda d: d["bufferViews"][0].update(byteOffset=10 ** 9)), abi.E_RANGE, "past its buffer")
    _expect(_edit(glb, lambda d: d["meshes"][0]["primitives"][0]["targets"][0].update(
        TANGENT=d["meshes"][0]["primitives"][0]["targets"][0]["POSITION"])), abi.E_INVALID, "BadAccessorDimensions")
    _expect(_edit(glb, lambda d: d["meshes"][0]["primitives"][0].update(mode=6)), abi.E_INVALID, "UnsupportedPrimitiveMode")
    # an unknown animation target node is caught by the flattener
    _expect(_edit(glb, lambda d: d["animations"][0]["channels"][0]["target"].update(node=10 ** 6)), abi.E_RANGE, "unknown node")


def test_hostile_files_are_rejected_not_followed():
    """Where the reference would overflow its stack or wrap an index computation, the loader returns an
    error (found with tools/fuzz/run.sh: loader + flattener + rig compiler under ASan/UBSan)."""
    spec = synth.tiny_spec(seed=58)
    rig = synth.make_rig(spec)
    glb, _ = write_glb(rig, seed=6)
    # a node that lists itself as a child: load_node (loader/gltf/mod.rs:1168-1207) recurses forever
    _expect(_edit(glb, lambda d: d["nodes"][d["scenes"][0]["nodes"][0]].update(children=[d["scenes"][0]["nodes"][0]])),
            abi.E_INVALID, "cycle")
    # count * stride wraps 64 bits: the bounds check must not be fooled
    def wrap(d):
        acc = d["accessors"][d["meshes"][0]["primitives"][0]["attributes"]["POSITION"]]
        acc["count"] = 2 ** 61
        d["bufferViews"][acc["bufferView"]]["byteStride"] = 16
    _expect(_edit(glb, wrap), abi.E_RANGE, "past its buffer")
    _expect(_edit(glb, lambda d: d["accessors"][0].update(byteOffset=2 ** 64 - 8)), abi.E_RANGE, "past its buffer")
    # an accessor without bufferView and an absurd count
    def huge(d):
        acc = d["accessors"][d["meshes"][0]["primitives"][0]["attributes"]["NORMAL"]]
        acc.pop("bufferView"); acc["count"] = 2 ** 40
    _expect(_edit(glb, huge), abi.E_RANGE, "too large")


def test_mutated_files_never_crash_the_loader():
    """A light version of tools/fuzz/run.sh without the sanitizers: every mutated file loads or is
    rejected with an error code."""
    spec = synth.tiny_spec(seed=59, targets=2)
    glb, _ = write_glb(synth.make_rig(spec), seed=8)
    rng = np.random.default_rng(4)
    jl = struct.unpack_from("<I", glb, 12)[0]
    outcomes = [0, 0]
    for _ in range(300):
        m = bytearray(glb)
        kind = int(rng.integers(0, 3))
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            digits = [k for k in range(20, 20 + jl) if 48 <= m[k] <= 57]
            for k in rng.choice(digits, size=2):
                m[int(k)] = 48 + int(rng.integers(0, 10))
        else:
            m = m[:int(rng.integers(0, len(m)))]
        try:
            hostlib.load_glb(bytes(m))
            outcomes[0] += 1
        except hostlib.HostError:
            outcomes[1] += 1
    assert outcomes[0] > 0 and outcomes[1] > 0, outcomes


def test_accessor_without_buffer_view_is_count_zero_bytes():
    # the reference yields `count` zero BYTES (not count x item size): a VEC3 float accessor of count 12
    # becomes 12 bytes = one zero vertex (loader/gltf/mod.rs:241-245)
    spec = synth.tiny_spec(seed=56, targets=0)
    rig = synth.make_rig(spec)
    glb, _ = write_glb(rig, seed=4, interleave=False)

    def strip(d):
        acc = d["accessors"][d["meshes"][0]["primitives"][0]["attributes"]["NORMAL"]]
        acc.pop("bufferView"); acc["count"] = 12
    with pytest.raises(hostlib.HostError) as e:   # 1 normal for 64 vertices: the flattener refuses the mismatch
        hostlib.load_glb(_edit(glb, strip))
    assert "attribute lengths differ" in e.value.message


def test_missing_normal_and_tangent_are_generated_like_the_reference():
    # tangent := normalised sum of unit face normals, w = 1; normal := that tangent's xyz
    # (tangent.rs:15-58, loader/gltf/mod.rs:1086-1089)
    spec = synth.tiny_spec(seed=57, targets=0, verts=6)
    rig = synth.make_rig(spec)
    rig.position = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 1, 1], [1, 0, 1]], np.float32)
    glb, _ = write_glb(rig, seed=5, interleave=False)

    def strip(d):
        a = d["meshes"][0]["primitives"][0]["attributes"]
        a.pop("NORMAL"); a.pop("TANGENT")
    loaded, _, _ = hostlib.load_glb(_edit(glb, strip))
    # triangle (p0,p1,p2): (p2-p1) x (p0-p1) = (-1,1,0) x (-1,0,0) = (0,0,1)
    assert loaded.tangent[:3, :3].tolist() == [[0, 0, 1]] * 3 and loaded.tangent[:, 3].tolist() == [1.0] * 6
    assert np.array_equal(loaded.normal, loaded.tangent[:, :3])
    want = hostlib.calculate_tangent(4, rig.position)
    assert np.array_equal(loaded.tangent[:, :3], want)


def test_calculate_tangent_indexed_accumulates_per_vertex():
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    idx = np.array([0, 1, 2, 0, 3, 1], np.uint32)
    t = hostlib.calculate_tangent(4, pos, idx)
    # faces: (0,1,2) -> (0,0,1); (0,3,1): (p1-p3) x (p0-p3) = (1,0,-1) x (0,0,-1) = (0,1,0)
    s = np.float32(1.0) / np.sqrt(np.float32(2.0))
    np.testing.assert_allclose(t[0], [0, s, s], rtol=1e-7)
    assert t[2].tolist() == [0, 0, 1] and t[3].tolist() == [0, 1, 0]
    with pytest.raises(hostlib.HostError):
        hostlib.calculate_tangent(3, pos)   # TriangleStrip: todo!() in the reference
    assert np.isnan(hostlib.calculate_tangent(0, pos)).all()   # points: zero vectors normalised -> NaN, as the reference


def test_host_calculate_tangent_equals_the_oracle_restatement(oracle32):
    """The product's host restatement (used by the loader for primitives and morph targets without TANGENT)
    against oracle/skin_oracle.c orc_calculate_tangent, bit for bit: a shuffled grid, repeated corners,
    degenerate triangles (NaN) and vertices no triangle touches."""
    rng = np.random.default_rng(8)
    n = 23
    xs, ys = np.meshgrid(np.arange(n + 1, dtype=np.float32), np.arange(n + 1, dtype=np.float32))
    pos = np.stack([xs, ys, rng.normal(scale=0.3, size=xs.shape).astype(np.float32)], axis=-1).reshape(-1, 3)
    q = np.arange(n * n); a = (q // n) * (n + 1) + q % n
    tris = np.concatenate([np.stack([a, a + 1, a + n + 1], 1), np.stack([a + 1, a + n + 2, a + n + 1], 1), [[5, 5, 9], [3, 3, 3]]])
    rng.shuffle(tris)
    idx = tris.astype(np.uint32).reshape(-1)
    for args in ((4, pos, idx), (4, pos[:300]), (0, pos[:10]), (2, pos[:10])):
        got, want = hostlib.calculate_tangent(*args), oracle32.calculate_tangent(*args)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_weights_channels_round_trip_through_glb(oracle32):
    """glTF "weights" animation channels (the reference: Property::MorphTargetWeights => todo!(), loader/gltf/mod.rs:1312):
    written as SCALAR outputs aimed at the mesh node, loaded into AnimationSampler::MorphTargetWeights, flattened to
    SKINRT_PATH_WEIGHTS with width = target count; the loaded rig evaluates like the source rig."""
    spec = synth.tiny_spec(seed=61, clips=3, keys=5, targets=4, weight_channels=3)
    rig = synth.make_rig(spec)
    glb, perm = write_glb(rig, seed=4)
    loaded, order, _ = hostlib.load_glb(glb)
    src = [[ch for ch in clip if ch.path == abi.PATH_WEIGHTS] for clip in rig.clips]
    got = [[ch for ch in clip if ch.path == abi.PATH_WEIGHTS] for clip in loaded.clips]
    assert [len(c) for c in got] == [1, 1, 1]
    for a, b in zip(src, got):
        assert a[0].interpolation == b[0].interpolation
        assert np.array_equal(a[0].times, b[0].times)
        assert np.array_equal(np.asarray(a[0].values).reshape(-1), np.asarray(b[0].values).reshape(-1))
    assert api.validate_rig(loaded).weight_channel_clips == 3
    crowd = synth.make_crowd(spec, rig, instances=9)
    ra = oracle32.evaluate(rig, crowd.tau, crowd.clip_id, crowd.morph_weights, want=("position", "normal", "tangent"))
    rb = oracle32.evaluate(loaded, crowd.tau, crowd.clip_id, crowd.morph_weights, want=("position", "normal", "tangent"))
    for k in ("position", "normal", "tangent"):
        assert np.array_equal(getattr(ra, k).view(np.uint32), getattr(rb, k).view(np.uint32)), k
    # wrong width: drop one target from the mesh, the channel still carries 4 weights per key
    bad = _edit(glb, lambda d: d["meshes"][0]["primitives"][0]["targets"].pop())
    _expect(bad, abi.E_INVALID, "values per key != morph target count")


def test_targets_without_tangent_accessor_load_and_validate(oracle32):
    """a morph target without a TANGENT accessor gets calculate_tangent(delta positions) from the
    loader, as the reference does (loader/gltf/mod.rs:911-916) — NaN for every vertex whose triangles do not move.  The
    flattener replaces those NaN deltas by zero (documented deviation), so such a file passes skinrt_validate_rig; the
    finite generated tangents are kept as the reference computes them."""
    spec = synth.tiny_spec(seed=62, targets=3, target_frac=0.2)   # most vertices are not moved by a target
    rig = synth.make_rig(spec)
    glb, _ = write_glb(rig, seed=5, omit_target_tangent=True)
    loaded, _, _ = hostlib.load_glb(glb)
    assert loaded.sanitized_tangent_deltas > 0
    for t in loaded.targets:
        assert t.tangent is not None and np.isfinite(t.tangent).all()
    info = api.validate_rig(loaded)
    assert info.target_count == 3
    crowd = synth.make_crowd(spec, rig, instances=4)
    r = oracle32.evaluate(loaded, crowd.tau, crowd.clip_id, crowd.morph_weights, want=("position", "tangent"))
    assert np.isfinite(r.tangent).all() and np.isfinite(r.position).all()


@pytest.mark.gpu
def test_glb_without_target_tangents_evaluates_on_the_gpu(oracle32):
    spec = synth.scaled(SPECS["c3_shape"], weight_channels=1)
    rig = synth.make_rig(spec)
    glb, _ = write_glb(rig, seed=6, omit_target_tangent=True)
    loaded, _, _ = hostlib.load_glb(glb)
    assert loaded.sanitized_tangent_deltas > 0
    crowd = synth.make_crowd(spec, rig, instances=7)
    got = helpers.run_gpu(loaded, crowd, exact_arithmetic=1)
    ref = helpers.run_oracle(oracle32, loaded, crowd)
    helpers.assert_equal(got["key_index"], ref["key_index"], "key_index")
    for k in ("world", "palette", "position", "normal", "tangent", "aabb"):
        helpers.assert_equal(got[k].view(np.uint32), np.asarray(ref[k]).view(np.uint32), k)


@pytest.mark.gpu
def test_loaded_glb_evaluates_on_the_gpu(oracle32):
    spec = SPECS["c3_shape"]
    rig = synth.make_rig(spec)
    glb, _ = write_glb(rig, seed=9)
    loaded, _, _ = hostlib.load_glb(glb)
    crowd = synth.make_crowd(spec, rig, instances=5)
    got = helpers.run_gpu(loaded, crowd, exact_arithmetic=1)
    ref = helpers.run_oracle(oracle32, rig, crowd)
    helpers.assert_equal(got["key_index"], ref["key_index"], "key_index")
    for k in ("world", "palette", "position", "normal", "tangent", "aabb"):
        helpers.assert_equal(got[k].view(np.uint32), np.asarray(ref[k]).view(np.uint32), k)


# ---- several skins and mesh nodes in one file (ABI 6) ----------------------------------------------------------------
def _multi_case(seed=61):
    from tests.glb_writer import write_glb_multi
    spec = synth.tiny_spec(seed=seed, verts=300, joints=10, nodes=18, clips=2, keys=4, targets=5, sets=2, weight_channels=1)
    base = synth.make_rig(spec)
    rig, tmap = synth.make_multi_rig(base, seed=3, fractions=(0.4, 0.35, 0.25), rigid_frac=0.1, rigid_sections=2)
    glb, perm = write_glb_multi(rig, seed=5)
    loaded, order, _ = hostlib.load_glb(glb)
    crowd = synth.make_crowd(spec, base, instances=4)
    mw = crowd.morph_weights[:, tmap]
    return rig, loaded, order, perm, crowd, mw


def _sections_by_node(rig):
    return {sc.node: sc for sc in rig.section_list()}


def _weights_for(loaded, rig, mw):
    """the original rig's [I][M] weights in the loaded rig's target order (sections come back in DFS order)"""
    out = np.zeros((mw.shape[0], loaded.target_count), np.float32)
    a, b = _sections_by_node(rig), _sections_by_node(loaded)
    for node, sb in b.items():
        sa = a[node]
        out[:, sb.first_target:sb.first_target + sb.target_count] = mw[:, sa.first_target:sa.first_target + sa.target_count]
    return out


def test_three_skins_and_two_props_round_trip_through_glb(oracle32):
    """A file with three skinned mesh nodes (three glTF skins over one skeleton) and two un-skinned mesh nodes hanging off
    joints: the loader + flattener keep EVERY mesh (until round 2 all but the first skin's were dropped) — one section
    per mesh node in DFS order, one skin per glTF skin in the order the walk meets them, targets and `weights` channels
    with their own section — and the flat rig evaluates, section by section, to the original's results bit for bit."""
    rig, loaded, order, perm, crowd, mw = _multi_case()
    assert np.array_equal(order, perm.astype(np.uint32))
    a, b = _sections_by_node(rig), _sections_by_node(loaded)
    assert sorted(a) == sorted(b) and len(loaded.skin_list()) == 3
    nodes_dfs = [sc.node for sc in loaded.section_list()]
    assert nodes_dfs == sorted(nodes_dfs), "sections follow the DFS order of their nodes"
    first = 0
    for sc in loaded.section_list():   # they tile the vertex array
        assert sc.first_vertex == first
        first += sc.vertex_count
    assert first == loaded.vertex_count == rig.vertex_count
    for node, sb in b.items():
        sa = a[node]
        assert (sa.vertex_count, sa.target_count, sa.skin >= 0) == (sb.vertex_count, sb.target_count, sb.skin >= 0)
        ra, rb = slice(sa.first_vertex, sa.first_vertex + sa.vertex_count), slice(sb.first_vertex, sb.first_vertex + sb.vertex_count)
        assert np.array_equal(rig.position[ra], loaded.position[rb]) and np.array_equal(rig.normal[ra], loaded.normal[rb])
        if sa.skin >= 0:
            ka, kb = rig.skin_list()[sa.skin], loaded.skin_list()[sb.skin]
            assert np.array_equal(ka.joint_node, kb.joint_node) and np.array_equal(ka.inverse_bind, kb.inverse_bind)
            for k in range(rig.set_count):
                assert np.array_equal(rig.joints[k][ra], loaded.joints[k][rb]) and np.array_equal(rig.weights[k][ra], loaded.weights[k][rb])
    info = api.validate_rig(loaded)
    assert (info.skin_count, info.section_count, info.palette_count) == (3, 5, rig.palette_count)
    want = oracle32.evaluate(rig, crowd.tau, crowd.clip_id, mw)
    got = oracle32.evaluate(loaded, crowd.tau, crowd.clip_id, _weights_for(loaded, rig, mw))
    helpers.assert_equal(got.world_rows().view(np.uint32), want.world_rows().view(np.uint32), "world")
    for node, sb in b.items():
        sa = a[node]
        ra, rb = slice(sa.first_vertex, sa.first_vertex + sa.vertex_count), slice(sb.first_vertex, sb.first_vertex + sb.vertex_count)
        for k in ("position", "normal", "tangent"):
            helpers.assert_equal(getattr(got, k)[:, rb].view(np.uint32), getattr(want, k)[:, ra].view(np.uint32), f"{k} of the section on node {node}")
    helpers.assert_equal(got.aabb.view(np.uint32), want.aabb.view(np.uint32), "bounds")


@pytest.mark.gpu
def test_loaded_multi_skin_glb_evaluates_on_the_gpu(oracle32):
    rig, loaded, _, _, crowd, mw = _multi_case(seed=62)
    w = _weights_for(loaded, rig, mw)
    state = synth.CrowdState(crowd.tau, crowd.clip_id, w, None)
    out = helpers.run_gpu(loaded, state, exact_arithmetic=1)
    ref = helpers.run_oracle(oracle32, loaded, state)
    helpers.assert_equal(out["key_index"], ref["key_index"], "key_index")
    for k in ("world", "palette", "position", "normal", "tangent", "aabb"):
        helpers.assert_equal(np.asarray(out[k]).view(np.uint32), np.asarray(ref[k]).view(np.uint32), k)
