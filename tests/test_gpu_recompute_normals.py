"""Per-frame normals from the DEFORMED positions (second half; include/skinrt.h
skinrt_create_topology / skinrt_recompute_normals): for every instance of a crowd, calculate_tangent
(renderer-asset/src/tangent.rs:15-58) of the positions the frame produced.  The checker is the ORACLE's restatement
(oracle/skin_oracle.c orc_calculate_tangent) run on each instance's downloaded positions: the comparison is bit for bit,
NaNs included (vertices no triangle touches).  PARITY UNPINNED against the reference itself (no toolchain here)."""
import numpy as np
import pytest

from renderer_amd import abi, api, synth

pytestmark = pytest.mark.gpu

TRIANGLE_LIST = 4


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _crowd(spec, instances, **cfg):
    rig = synth.make_rig(spec)
    crowd = synth.make_crowd(spec, rig, instances=instances)
    ctx = api.Context(**cfg)
    r = api.Rig(ctx, rig)
    c = api.Crowd(ctx, r, instances)
    c.set_clocks(crowd.tau, crowd.clip_id)
    if crowd.morph_weights is not None:
        c.set_morph_weights(crowd.morph_weights)
    c.evaluate(sync=True)
    return rig, ctx, r, c


def _triangles(V, rng, faces, pole=0):
    """a random triangle soup over V vertices: repeated corners, a vertex named by `pole` triangles (heapsort branch when
    > 32 corners), the last two vertices touched by nothing"""
    tris = rng.integers(0, V - 2, size=(faces, 3))
    if pole:
        fan = np.stack([np.full(pole, 7), rng.integers(0, V - 2, pole), rng.integers(0, V - 2, pole)], 1)
        tris = np.concatenate([tris, fan])
        rng.shuffle(tris)
    tris[3] = [5, 5, 9]   # a vertex twice in one triangle: the sequential loop adds that normal twice
    return tris.astype(np.uint32).reshape(-1)


@pytest.mark.parametrize("exact", [0, 1])
def test_every_instance_matches_calculate_tangent_of_its_own_positions(oracle32, exact):
    spec = synth.tiny_spec(seed=5, verts=300, targets=4)
    I = 37
    rig, ctx, r, c = _crowd(spec, I, exact_arithmetic=exact)
    V = rig.vertex_count
    rng = np.random.default_rng(17)
    idx = _triangles(V, rng, faces=900, pole=120)
    before_n, before_t = c.normals().copy(), c.tangents().copy()
    with api.Topology(ctx, V, idx) as topo:
        c.recompute_normals(topo, output=0, sync=True)
        pos = c.positions().reshape(I, V, 3)
        got = c.normals().reshape(I, V, 3)
        for i in range(I):
            want = oracle32.calculate_tangent(TRIANGLE_LIST, pos[i], idx)
            assert np.array_equal(bits(got[i]), bits(want)), f"instance {i}"
        assert np.isnan(got[:, -1]).all() and np.isfinite(got[:, 7]).all()
        assert np.array_equal(bits(c.tangents()), bits(before_t))          # the other array is not touched
        # a sub-range into the tangents: only those instances change, the normals stay what the call above made them
        c.recompute_normals(topo, first=5, count=11, output=1, sync=True)
        tan = c.tangents().reshape(I, V, 3)
        assert np.array_equal(bits(tan[5:16]), bits(got[5:16]))
        assert np.array_equal(bits(tan[:5]), bits(before_t.reshape(I, V, 3)[:5])) and np.array_equal(bits(tan[16:]), bits(before_t.reshape(I, V, 3)[16:]))
        # a new frame overwrites the normals again (evaluate owns the arrays); recomputing follows the new positions
        c.evaluate(sync=True)
        assert np.array_equal(bits(c.normals()), bits(before_n))
    c.close(); r.close(); ctx.close()


def test_unindexed_topology_and_follows_the_frame(oracle32):
    """indices = NULL: vertices 3f, 3f+1, 3f+2 form triangle f (tangent.rs:42-51); after the clocks move, the recomputed
    normals are those of the NEW positions."""
    spec = synth.tiny_spec(seed=9, verts=96)
    I = 8
    rig, ctx, r, c = _crowd(spec, I)
    V = rig.vertex_count
    with api.Topology(ctx, V) as topo:
        for frame in range(2):
            c.recompute_normals(topo, sync=True)
            pos = c.positions().reshape(I, V, 3)
            got = c.normals().reshape(I, V, 3)
            for i in range(I):
                assert np.array_equal(bits(got[i]), bits(oracle32.calculate_tangent(TRIANGLE_LIST, pos[i])))
            c.advance_clocks(0.37)
            c.evaluate(sync=True)
    c.close(); r.close(); ctx.close()


def test_argument_errors():
    spec = synth.tiny_spec(seed=1, verts=64)
    rig, ctx, r, c = _crowd(spec, 4)
    V = rig.vertex_count
    with pytest.raises(api.SkinrtError) as e:
        api.Topology(ctx, V, np.array([0, 1, V], np.uint32))
    assert e.value.code == abi.E_INVALID and "out of range" in e.value.message
    with pytest.raises(api.SkinrtError) as e:
        api.Topology(ctx, 0)
    assert e.value.code == abi.E_INVALID
    with api.Topology(ctx, V + 1, np.array([0, 1, 2], np.uint32)) as other:
        with pytest.raises(api.SkinrtError) as e:
            c.recompute_normals(other)
        assert e.value.code == abi.E_INVALID and "vertex count" in e.value.message
    with api.Topology(ctx, V, np.array([0, 1, 2, 3], np.uint32)) as topo:   # one whole triangle; the fourth index is not used
        with pytest.raises(api.SkinrtError) as e:
            c.recompute_normals(topo, first=3, count=2)
        assert e.value.code == abi.E_RANGE
        with pytest.raises(api.SkinrtError) as e:
            c.recompute_normals(topo, output=2)
        assert e.value.code == abi.E_INVALID
        c.recompute_normals(topo, first=4, count=0)   # empty range: nothing to do
        with api.Context() as ctx2:
            with api.Topology(ctx2, V, np.array([0, 1, 2], np.uint32)) as foreign:
                with pytest.raises(api.SkinrtError) as e:
                    c.recompute_normals(foreign)
                assert e.value.code == abi.E_INVALID and "another ctx" in e.value.message
    c.close(); r.close(); ctx.close()


def test_c3_shape_rate(capsys):
    """C3's avatar (60,000 vertices) with a grid-like triangle list of valence ~6, 512 instances: the kernel's rate.  Gather
    work out of L2: per (instance, vertex) ~6 faces x 3 positions read, one 12-byte vector written."""
    import time
    spec = synth.CONFIGS["c3"]
    I = 512
    rig, ctx, r, c = _crowd(spec, I)
    V = rig.vertex_count
    n = 244   # 245 x 245 = 60,025 grid points >= V: clip to V
    q = np.arange(n * n)
    a = (q // n) * (n + 1) + q % n
    tris = np.concatenate([np.stack([a, a + 1, a + n + 1], 1), np.stack([a + 1, a + n + 2, a + n + 1], 1)])
    tris = tris[(tris < V).all(axis=1)].astype(np.uint32).reshape(-1)
    with api.Topology(ctx, V, tris) as topo:
        c.recompute_normals(topo, sync=True)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            c.recompute_normals(topo)
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        got = c.normals(first=0, count=1).reshape(V, 3)
        touched = np.zeros(V, bool); touched[tris] = True
        assert np.isfinite(got[touched]).all()
    with capsys.disabled():
        print(f"\n[recompute normals] {I} x {V} vertices, {tris.size // 3} triangles: {ms:.3f} ms = {I * V / ms / 1e6:.2f} G vertex-instances/s")
    c.close(); r.close(); ctx.close()
