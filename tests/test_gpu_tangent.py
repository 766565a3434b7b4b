"""calculate_tangent on the device (include/skinrt.h skinrt_calculate_tangent)
against the ORACLE's restatement of renderer-asset/src/tangent.rs:15-58 (oracle/skin_oracle.c
orc_calculate_tangent; pinned by hand-computed cases in tests/test_oracle_known_answers.py).  The sum
per vertex follows the reference's order, so the comparison is bit for bit — NaNs included (degenerate
triangles, vertices no triangle touches).  PARITY UNPINNED against the reference itself (no toolchain
here).  The product's own host restatement (renderer_amd/host, used by the .glb loader) is checked
against the same oracle in tests/test_glb_loader.py."""
import numpy as np
import pytest

from renderer_amd import abi, api

pytestmark = pytest.mark.gpu

TRIANGLE_LIST = 4


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def grid_mesh(n, rng):
    """n x n quads of a bumpy height field, two triangles each, shuffled triangle order"""
    xs, ys = np.meshgrid(np.arange(n + 1, dtype=np.float32), np.arange(n + 1, dtype=np.float32))
    pos = np.stack([xs, ys, rng.normal(scale=0.3, size=xs.shape).astype(np.float32)], axis=-1).reshape(-1, 3)
    q = np.arange(n * n)
    a = (q // n) * (n + 1) + q % n
    tris = np.concatenate([np.stack([a, a + 1, a + n + 1], 1), np.stack([a + 1, a + n + 2, a + n + 1], 1)])
    rng.shuffle(tris)
    return pos, tris.astype(np.uint32).reshape(-1)


@pytest.mark.parametrize("n", [1, 7, 64, 1001])   # 1001: an even vertex count spanning many scan chunks
def test_indexed_mesh_matches_the_sequential_loop_bit_for_bit(oracle32, n):
    rng = np.random.default_rng(n)
    pos, idx = grid_mesh(n, rng)
    with api.Context() as ctx:
        got = ctx.calculate_tangent(TRIANGLE_LIST, pos, idx)
    want = oracle32.calculate_tangent(TRIANGLE_LIST, pos, idx)
    assert np.array_equal(bits(got), bits(want))
    assert np.isfinite(got).all()


def test_unindexed_degenerate_and_untouched_vertices(oracle32):
    rng = np.random.default_rng(3)
    pos = rng.normal(size=(3 * 50 + 2, 3)).astype(np.float32)   # the last two vertices belong to no triangle
    pos[9:12] = pos[9]                                           # a triangle with three equal corners: 0/0
    pos[30] = pos[31]                                            # a sliver
    with api.Context() as ctx:
        got = ctx.calculate_tangent(TRIANGLE_LIST, pos)
    want = oracle32.calculate_tangent(TRIANGLE_LIST, pos)
    assert np.array_equal(bits(got), bits(want))
    assert np.isnan(got[-1]).all() and np.isnan(got[9]).all()


def test_high_valence_and_repeated_corners(oracle32):
    # a fan of 300 triangles around vertex 0 (its corner list is sorted per vertex on the device), plus
    # triangles that name a vertex twice (the sequential loop then adds that normal twice)
    rng = np.random.default_rng(5)
    k = 300
    ang = np.linspace(0, 2 * np.pi, k, endpoint=False)
    pos = np.concatenate([[[0, 0, 1]], np.stack([np.cos(ang), np.sin(ang), rng.normal(scale=0.2, size=k)], 1)]).astype(np.float32)
    fan = np.stack([np.zeros(k, np.int64), 1 + np.arange(k), 1 + (np.arange(k) + 1) % k], 1)
    rng.shuffle(fan)
    odd = np.array([[5, 5, 9], [0, 7, 0], [3, 3, 3]])
    idx = np.concatenate([fan, odd]).astype(np.uint32).reshape(-1)
    with api.Context() as ctx:
        got = ctx.calculate_tangent(TRIANGLE_LIST, pos, idx)
    want = oracle32.calculate_tangent(TRIANGLE_LIST, pos, idx)
    assert np.array_equal(bits(got), bits(want))


def _fan(k, rng):
    ang = np.linspace(0, 2 * np.pi, k, endpoint=False)
    pos = np.concatenate([[[0, 0, 1]], np.stack([np.cos(ang), np.sin(ang), rng.normal(scale=0.2, size=k)], 1)]).astype(np.float32)
    fan = np.stack([np.zeros(k, np.int64), 1 + np.arange(k), 1 + (np.arange(k) + 1) % k], 1)
    rng.shuffle(fan)
    return pos, fan.astype(np.uint32).reshape(-1)


def test_a_pole_of_thousands_of_triangles_is_sorted_in_n_log_n(oracle32):
    """the per-vertex corner sort was quadratic.  A pole with 12,000 triangles (its vertex is named by 12,000
    corners, every rim vertex by two) goes through the heapsort branch and still adds its normals in the sequential
    loop's order: bit-equal.  More corners on one vertex than SKINRT_TANGENT_MAX_VALENCE is refused where it can be seen."""
    rng = np.random.default_rng(8)
    pos, idx = _fan(12000, rng)
    with api.Context() as ctx:
        got, ms = ctx.calculate_tangent(TRIANGLE_LIST, pos, idx, with_ms=True)
        want = oracle32.calculate_tangent(TRIANGLE_LIST, pos, idx)
        assert np.array_equal(bits(got), bits(want))
        print(f"[tangent] pole of 12,000 triangles: kernels {ms:.2f} ms")
        assert ms < 2000.0
        big_pos, big_idx = _fan(49152 + 1, rng)
        with pytest.raises(api.SkinrtError) as e:
            ctx.calculate_tangent(TRIANGLE_LIST, big_pos, big_idx)
        assert e.value.code == abi.E_LIMIT and "SKINRT_TANGENT_MAX_VALENCE" in e.value.message


def test_modes_and_errors():
    pos = np.eye(3, dtype=np.float32)
    with api.Context() as ctx:
        for mode in (0, 1, 2):   # points and lines: no faces, zero sums normalise to NaN as in the reference
            assert np.isnan(ctx.calculate_tangent(mode, pos)).all()
        with pytest.raises(api.SkinrtError) as e:
            ctx.calculate_tangent(3, pos)    # TriangleStrip: todo!() in the reference
        assert e.value.code == abi.E_INVALID and "todo!()" in e.value.message
        with pytest.raises(api.SkinrtError) as e:
            ctx.calculate_tangent(TRIANGLE_LIST, pos, np.array([0, 1, 3], np.uint32))
        assert e.value.code == abi.E_INVALID and "out of range" in e.value.message
        with pytest.raises(api.SkinrtError):
            ctx.calculate_tangent(9, pos)
        assert ctx.calculate_tangent(TRIANGLE_LIST, np.zeros((0, 3), np.float32)).shape == (0, 3)


def test_million_vertex_mesh_and_its_rate(oracle32, capsys):
    """C5-sized mesh (1M vertices, 2M triangles): bit-equal to the host loop; the kernels' time is printed
    (HBM-bound integer/float gather work: positions 12 B + indices 12 B read per triangle corner set,
    face normals written and read back once per corner)."""
    rng = np.random.default_rng(11)
    pos, idx = grid_mesh(1000, rng)
    with api.Context() as ctx:
        ctx.calculate_tangent(TRIANGLE_LIST, pos, idx)                      # warm-up
        got, ms = ctx.calculate_tangent(TRIANGLE_LIST, pos, idx, with_ms=True)
    want = oracle32.calculate_tangent(TRIANGLE_LIST, pos, idx)
    assert np.array_equal(bits(got), bits(want))
    V, C3 = pos.shape[0], idx.shape[0]
    algorithmic = V * 12 + C3 * 4 + V * 12        # positions and indices in, tangents out
    with capsys.disabled():
        print(f"\n[tangent] {V} vertices, {C3 // 3} triangles: kernels {ms:.3f} ms, {algorithmic / ms / 1e6:.1f} GB/s of algorithmic bytes")
