#!/usr/bin/env python3
"""What slows the skin stage when it follows the pose stage?  Sequences of stage calls, mean skin-stage ms per position."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import abi, api, synth

spec = synth.CONFIGS["c3"]
rig_data = synth.make_rig(spec)
crowd_in = synth.make_crowd(spec, rig_data, instances=spec.instances)
ctx = api.Context(); rig = api.Rig(ctx, rig_data); crowd = api.Crowd(ctx, rig, spec.instances)
crowd.set_clocks(crowd_in.tau, crowd_in.clip_id); crowd.set_morph_weights(crowd_in.morph_weights)
P, S = abi.STAGE_POSE, abi.STAGE_SKIN | abi.STAGE_AABB

def seq(name, steps, n=25):
    acc = [[] for _ in steps]
    for k in range(n + 3):
        for q, st in enumerate(steps):
            if isinstance(st, float):
                ctx.synchronize(); time.sleep(st); continue
            crowd.evaluate(st)
            ms = crowd.stage_ms()
            if k >= 3: acc[q].append(ms[1] if st & abi.STAGE_SKIN else ms[0])
    print(f"{name:34s}", "  ".join(f"{np.mean(a):.3f}" if a else "sleep" for a in acc), flush=True)

seq("ALL", [abi.STAGE_ALL])
seq("P, S", [P, S])
seq("P, S, S", [P, S, S])
seq("P, S, S, S", [P, S, S, S])
seq("P, P, S", [P, P, S])
seq("P, sleep 0.3ms, S", [P, 0.0003, S])
seq("S only", [S])
seq("ALL again", [abi.STAGE_ALL])

# what precedes the skin stage, other than the pose stage: a fill of 0.6 GB / 4 GB on the same stream (caches flushed, chip busy)
import ctypes as C
hip = C.CDLL("libamdhip64.so")
buf = C.c_void_p(); assert hip.hipMalloc(C.byref(buf), C.c_size_t(4 << 30)) == 0
stream = C.c_void_p(ctx.stream)
def seq_fill(name, nbytes, n=25):
    acc = []
    for k in range(n + 3):
        assert hip.hipMemsetAsync(buf, C.c_int(1), C.c_size_t(nbytes), stream) == 0
        crowd.evaluate(S); ms = crowd.stage_ms()
        if k >= 3: acc.append(ms[1])
    print(f"{name:34s} {np.mean(acc):.3f}", flush=True)
seq_fill("fill 0.6 GB, S", 600 << 20)
seq_fill("fill 4 GB, S", 4 << 30)
seq_fill("fill 64 MB, S", 64 << 20)
seq("S only", [S])
