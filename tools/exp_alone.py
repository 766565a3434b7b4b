"""Skin stage timed inside full frames and back to back on its own (same crowd, same inputs): the "after another
kernel" effect of DESIGN.md §3.2, re-measured on the current build.  usage: python3 tools/exp_alone.py [c3|c5]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import abi, api, synth

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
spec = synth.CONFIGS[name]
rig = synth.make_rig(spec)
I = spec.instances
crowd = synth.make_crowd(spec, rig, instances=I)
with api.Context() as ctx:
    r = api.Rig(ctx, rig); c = api.Crowd(ctx, r, I)
    c.set_clocks(crowd.tau, crowd.clip_id)
    if crowd.morph_weights is not None: c.set_morph_weights(crowd.morph_weights)
    for mode, mask in (("full frames", abi.STAGE_ALL), ("skin + bounds only", abi.STAGE_SKIN | abi.STAGE_AABB), ("full frames", abi.STAGE_ALL)):
        ms = []
        for rep in range(25):
            c.evaluate(mask)
        ctx.synchronize()
        for rep in range(20):
            c.evaluate(mask); ctx.synchronize(); ms.append(c.stage_ms()[1])
        burst = []
        for rep in range(20):
            c.evaluate(mask)
        ctx.synchronize(); burst.append(c.stage_ms()[1])
        print(f"{name} {mode:20s}: skin stage median {np.median(ms):.3f} ms (sync after each), last of a burst of 20: {burst[0]:.3f} ms", flush=True)
    c.close(); r.close()
