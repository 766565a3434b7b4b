"""A host that holds HIP streams of its own.  HIP maps streams onto a few hardware queues; skinrt_init measures whether
its side stream (k_skin_fix beside k_skin2) really runs beside the main stream and replaces one that shares the main
stream's queue (renderer_amd/csrc/api.hip: side_by_side).  Here the process creates a crowd of streams
first — every one of them used, so that it owns a queue — and the context that comes after must still produce the
oracle's bits on a rig whose frames use both of its streams (work lists: three or more active morph pairs per tile)."""
import ctypes as C

import numpy as np
import pytest

from renderer_amd import synth
from tests.helpers import assert_equal, run_gpu, run_oracle

pytestmark = pytest.mark.gpu


def test_context_created_among_many_host_streams(oracle32):
    hip = C.CDLL("libamdhip64.so")
    streams = []
    buf = C.c_void_p()
    assert hip.hipMalloc(C.byref(buf), C.c_size_t(1 << 20)) == 0
    try:
        for k in range(9):
            s = C.c_void_p()
            assert hip.hipStreamCreateWithFlags(C.byref(s), C.c_uint(1)) == 0  # hipStreamNonBlocking
            assert hip.hipMemsetAsync(buf, C.c_int(k), C.c_size_t(1 << 20), s) == 0  # a stream owns a queue once it is used
            streams.append(s)
        for s in streams:
            assert hip.hipStreamSynchronize(s) == 0
        spec = synth.scaled(synth.tiny_spec(seed=77), verts=1500, targets=12, target_mode="dense")
        rig = synth.make_rig(spec)
        crowd = synth.make_crowd(spec, rig, instances=300)
        crowd.morph_weights[:] = np.float32(0.25)  # every target active: every tile with three or more pairs goes on the work lists
        for rep in range(3):  # several contexts in a row, each with its own check
            got = run_gpu(rig, crowd, exact_arithmetic=1)
            ref = run_oracle(oracle32, rig, crowd)
            for k in ("position", "normal", "tangent", "aabb"):
                assert_equal(got[k], ref[k], f"context {rep}: {k}")
    finally:
        for s in streams:
            hip.hipStreamDestroy(s)
        hip.hipFree(buf)
