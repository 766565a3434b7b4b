#!/usr/bin/env python3
"""Does reading the palette once before the skin stage (so that it sits in the memory-side cache) remove the
first-skin-after-pose penalty?"""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import abi, api, synth
spec = synth.CONFIGS["c3"]
rig_data = synth.make_rig(spec)
crowd_in = synth.make_crowd(spec, rig_data, instances=spec.instances)
ctx = api.Context(); rig = api.Rig(ctx, rig_data); crowd = api.Crowd(ctx, rig, spec.instances)
crowd.set_clocks(crowd_in.tau, crowd_in.clip_id); crowd.set_morph_weights(crowd_in.morph_weights)
P, S = abi.STAGE_POSE, abi.STAGE_SKIN | abi.STAGE_AABB
hip = C.CDLL("libamdhip64.so")
stream = C.c_void_p(ctx.stream)
pal, npal = crowd.device_pointer("palette")
scratch = C.c_void_p(); assert hip.hipMalloc(C.byref(scratch), C.c_size_t(npal * 4)) == 0

def run(name, between, n=25):
    acc = []
    for k in range(n + 3):
        crowd.evaluate(P)
        between()
        crowd.evaluate(S); ms = crowd.stage_ms()
        if k >= 3: acc.append(ms[1])
    print(f"{name:44s} {np.mean(acc):.3f}", flush=True)

run("P, S", lambda: None)
run("P, copy palette -> scratch (192 MB), S", lambda: hip.hipMemcpyAsync(scratch, C.c_void_p(pal), C.c_size_t(npal * 4), C.c_int(3), stream))
run("P, copy first 64 MB of palette, S", lambda: hip.hipMemcpyAsync(scratch, C.c_void_p(pal), C.c_size_t(64 << 20), C.c_int(3), stream))
run("P, S", lambda: None)
def s_only(n=25):
    acc = []
    for k in range(n + 3):
        crowd.evaluate(S); ms = crowd.stage_ms()
        if k >= 3: acc.append(ms[1])
    print(f"{'S only':44s} {np.mean(acc):.3f}", flush=True)
s_only()
