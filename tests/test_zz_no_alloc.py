"""skinrt_calculate_tangent_noalloc (include/skinrt.h): the device-pointer variant of skinrt_calculate_tangent whose
work area is a static array inside the library — for callers that cannot call hipMalloc.  It has no
counterpart in the reference; the bar is the existing entry point's output, bit for bit, on two calls in
a row, the second with a smaller input (the shared work area keeps the first call's contents)."""
import ctypes as C

import numpy as np
import pytest

from renderer_amd import abi, api
from tests.helpers import DeviceBuffers

pytestmark = pytest.mark.gpu

TRIANGLE_LIST = 4


def grid_mesh(n, rng):
    xs, ys = np.meshgrid(np.arange(n + 1, dtype=np.float32), np.arange(n + 1, dtype=np.float32))
    pos = np.stack([xs, ys, rng.normal(scale=0.3, size=xs.shape).astype(np.float32)], axis=-1).reshape(-1, 3)
    q = np.arange(n * n)
    a = (q // n) * (n + 1) + q % n
    tris = np.concatenate([np.stack([a, a + 1, a + n + 1], 1), np.stack([a + 1, a + n + 2, a + n + 1], 1)])
    rng.shuffle(tris)
    return pos, tris.astype(np.uint32).reshape(-1)




def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_noalloc_equals_the_allocating_entry_point_twice_in_a_row():
    rng = np.random.default_rng(41)
    big = grid_mesh(300, rng)        # 90,601 vertices, 180,000 triangles
    small = grid_mesh(37, rng)       # 1,444 vertices: second call, smaller, on the work area the first one left
    flat = rng.normal(size=(3 * 211 + 1, 3)).astype(np.float32)   # third: no index buffer, one dangling vertex
    dev = DeviceBuffers()
    try:
        with api.Context() as ctx:
            for pos, idx in (big, small, (flat, None)):
                want = ctx.calculate_tangent(TRIANGLE_LIST, pos, idx)
                d_pos = dev.upload(pos)
                d_idx = dev.upload(idx) if idx is not None else 0
                d_out = dev.empty(pos.nbytes)
                ctx.calculate_tangent_noalloc(TRIANGLE_LIST, d_pos, pos.shape[0], d_idx, 0 if idx is None else idx.shape[0], d_out)
                ctx.synchronize()
                got = dev.download(d_out, pos.shape, np.float32)
                assert np.array_equal(bits(got), bits(want)), f"{pos.shape[0]} vertices"
    finally:
        dev.close()


def test_noalloc_limits_and_errors():
    dev = DeviceBuffers()
    try:
        with api.Context() as ctx:
            p = dev.empty(64)
            for args, code in (((TRIANGLE_LIST, p, 262145, 0, 0, p), abi.E_LIMIT),          # too many vertices
                               ((TRIANGLE_LIST, p, 10, p, 1572864 + 3, p), abi.E_LIMIT),    # too many indices
                               ((3, p, 3, 0, 0, p), abi.E_INVALID),                         # TriangleStrip: todo!() in the reference
                               ((TRIANGLE_LIST, 0, 3, 0, 0, p), abi.E_INVALID),             # NULL positions
                               ((TRIANGLE_LIST, p, 3, 0, 3, p), abi.E_INVALID)):            # index_count without indices
                with pytest.raises(api.SkinrtError) as e:
                    ctx.calculate_tangent_noalloc(*args)
                assert e.value.code == code, args
            # an index past the vertex count cannot be seen on the host: its triangle contributes nothing
            pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [5, 5, 5]], np.float32)
            idx = np.array([0, 1, 2, 0, 1, 99], np.uint32)
            d_pos, d_idx, d_out = dev.upload(pos), dev.upload(idx), dev.empty(pos.nbytes)
            ctx.calculate_tangent_noalloc(TRIANGLE_LIST, d_pos, 4, d_idx, 6, d_out)
            ctx.synchronize()
            got = dev.download(d_out, (4, 3), np.float32)
            assert got[:3].tolist() == [[0, 0, 1]] * 3 and np.isnan(got[3]).all()
    finally:
        dev.close()
