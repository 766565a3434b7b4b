"""does the "skin stage after another kernel" effect (the stage takes 0.2-0.4 ms
longer inside a frame than back to back on its own) come from dispatch gaps?  The frame is captured into a HIP graph
(skinrt_evaluate enqueues kernels, a memset, events and a fork / join over a side stream: all capturable) and replayed;
a graph launch has no per-kernel host dispatch.  Wall time per frame from HIP events around bursts of 30 frames.
usage: python3 tools/exp_graph.py [c3|c5]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import abi, api, synth

hip = C.CDLL("libamdhip64.so")


def ok(rc):
    assert rc == 0, f"HIP error {rc}"


name = sys.argv[1] if len(sys.argv) > 1 else "c3"
spec = synth.CONFIGS[name]
rig = synth.make_rig(spec)
I = spec.instances
crowd = synth.make_crowd(spec, rig, instances=I)
with api.Context() as ctx:
    r = api.Rig(ctx, rig)
    c = api.Crowd(ctx, r, I)
    c.set_clocks(crowd.tau, crowd.clip_id)
    if crowd.morph_weights is not None:
        c.set_morph_weights(crowd.morph_weights)
    stream = C.c_void_p(ctx.stream)
    e0, e1 = C.c_void_p(), C.c_void_p()
    ok(hip.hipEventCreate(C.byref(e0))); ok(hip.hipEventCreate(C.byref(e1)))

    def burst(fn, n=30):
        for _ in range(10):
            fn()
        ctx.synchronize()
        ok(hip.hipEventRecord(e0, stream))
        for _ in range(n):
            fn()
        ok(hip.hipEventRecord(e1, stream))
        ok(hip.hipEventSynchronize(e1))
        ms = C.c_float()
        ok(hip.hipEventElapsedTime(C.byref(ms), e0, e1))
        return ms.value / n

    def capture(mask):
        graph, gexec = C.c_void_p(), C.c_void_p()
        ctx.synchronize()
        ok(hip.hipStreamBeginCapture(stream, C.c_int(2)))  # relaxed: the library calls hipSetDevice etc. meanwhile
        c.evaluate(mask)
        ok(hip.hipStreamEndCapture(stream, C.byref(graph)))
        ok(hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)))
        return graph, gexec

    for label, mask in (("full frame", abi.STAGE_ALL), ("skin + bounds", abi.STAGE_SKIN | abi.STAGE_AABB)):
        plain = [burst(lambda: c.evaluate(mask)) for _ in range(3)]
        stage = c.stage_ms()
        graph, gexec = capture(mask)
        replay = [burst(lambda: ok(hip.hipGraphLaunch(gexec, stream))) for _ in range(3)]
        ref = c.aabbs().copy()
        c.evaluate(mask, sync=True)
        same = np.array_equal(ref.view(np.uint32), c.aabbs().view(np.uint32))
        print(f"{name} {label:14s}: stream launches {min(plain):.3f} ms/frame (skin stage of the last frame {stage[1]:.3f}), "
              f"graph replay {min(replay):.3f} ms/frame, same bounds: {same}", flush=True)
        hip.hipGraphExecDestroy(gexec); hip.hipGraphDestroy(graph)
    c.close(); r.close()
