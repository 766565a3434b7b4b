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


def _mem_free(hip):
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_noalloc_allocates_nothing_and_runs_under_graph_capture():
    """the property itself.  (i) hipMemGetInfo reports the same free memory before and after the calls;
    (ii) the entry point is legal inside hipStreamBeginCapture / EndCapture — where hipMalloc is not — and the graph,
    launched twice, reproduces the allocating entry point's output bit for bit; (iii) two contexts on one GPU calling
    it back to back on their own streams (they share the GPU's work area) both get the right result: the library
    orders the second call's kernels behind the first's."""
    rng = np.random.default_rng(43)
    pos, idx = grid_mesh(120, rng)
    pos2, idx2 = grid_mesh(90, rng)
    dev = DeviceBuffers()
    hip = dev.hip
    try:
        with api.Context() as ctx, api.Context() as ctx2:
            want = ctx.calculate_tangent(TRIANGLE_LIST, pos, idx)
            want2 = ctx.calculate_tangent(TRIANGLE_LIST, pos2, idx2)
            d_pos, d_idx, d_out = dev.upload(pos), dev.upload(idx), dev.empty(pos.nbytes)
            e_pos, e_idx, e_out = dev.upload(pos2), dev.upload(idx2), dev.empty(pos2.nbytes)
            ctx.synchronize()
            assert hip.hipDeviceSynchronize() == 0
            before = _mem_free(hip)
            for _ in range(3):
                ctx.calculate_tangent_noalloc(TRIANGLE_LIST, d_pos, pos.shape[0], d_idx, idx.shape[0], d_out)
            ctx.synchronize()
            assert _mem_free(hip) == before, "the allocation-free entry point changed the device's free memory"
            assert np.array_equal(bits(dev.download(d_out, pos.shape, np.float32)), bits(want))
            # (iii) two streams, one work area, no host wait in between
            for _ in range(4):
                ctx.calculate_tangent_noalloc(TRIANGLE_LIST, d_pos, pos.shape[0], d_idx, idx.shape[0], d_out)
                ctx2.calculate_tangent_noalloc(TRIANGLE_LIST, e_pos, pos2.shape[0], e_idx, idx2.shape[0], e_out)
            ctx.synchronize(); ctx2.synchronize()
            assert np.array_equal(bits(dev.download(d_out, pos.shape, np.float32)), bits(want)), "first stream"
            assert np.array_equal(bits(dev.download(e_out, pos2.shape, np.float32)), bits(want2)), "second stream"
            # (ii) graph capture on the context's own stream
            stream = C.c_void_p(ctx.stream)
            graph, gexec = C.c_void_p(), C.c_void_p()
            assert hip.hipMemset(C.c_void_p(d_out), 0, C.c_size_t(pos.nbytes)) == 0
            assert hip.hipDeviceSynchronize() == 0
            assert hip.hipStreamBeginCapture(stream, C.c_int(2)) == 0          # hipStreamCaptureModeRelaxed
            ctx.calculate_tangent_noalloc(TRIANGLE_LIST, d_pos, pos.shape[0], d_idx, idx.shape[0], d_out)
            assert hip.hipStreamEndCapture(stream, C.byref(graph)) == 0
            assert hip.hipDeviceSynchronize() == 0
            assert not dev.download(d_out, pos.shape, np.float32).any(), "capture must not run anything"
            assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)) == 0
            for _ in range(2):
                assert hip.hipMemset(C.c_void_p(d_out), 0, C.c_size_t(pos.nbytes)) == 0
                assert hip.hipDeviceSynchronize() == 0
                assert hip.hipGraphLaunch(gexec, stream) == 0
                ctx.synchronize()
                assert np.array_equal(bits(dev.download(d_out, pos.shape, np.float32)), bits(want)), "graph launch"
            hip.hipGraphExecDestroy(gexec)
            hip.hipGraphDestroy(graph)
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
            # more corners on one vertex than the kernels sort (SKINRT_TANGENT_MAX_VALENCE): that vertex gets NaN, the rest is right
            k = 49152 // 3 + 1
            ang = np.linspace(0, 2 * np.pi, k, endpoint=False)
            fpos = np.concatenate([[[0, 0, 1]], np.stack([np.cos(ang), np.sin(ang), np.zeros(k)], 1)]).astype(np.float32)
            fidx = np.stack([np.zeros(k, np.int64), 1 + np.arange(k), 1 + (np.arange(k) + 1) % k,
                             np.zeros(k, np.int64), 1 + np.arange(k), 1 + (np.arange(k) + 1) % k,
                             np.zeros(k, np.int64), 1 + np.arange(k), 1 + (np.arange(k) + 1) % k], 1).astype(np.uint32).reshape(-1)
            d_pos, d_idx, d_out = dev.upload(fpos), dev.upload(fidx), dev.empty(fpos.nbytes)
            ctx.calculate_tangent_noalloc(TRIANGLE_LIST, d_pos, fpos.shape[0], d_idx, fidx.shape[0], d_out)
            ctx.synchronize()
            got = dev.download(d_out, fpos.shape, np.float32)
            assert np.isnan(got[0]).all() and np.isfinite(got[1:]).all()
    finally:
        dev.close()
