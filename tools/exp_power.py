#!/usr/bin/env python3
"""Is the skin stage's time a function of what ran just before it?  (a) whole frames back to back, (b) the skin
stage alone back to back, (c) whole frames with the host idle between them, (d) skin alone with idle gaps."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import abi, api, synth

spec = synth.CONFIGS["c3"]
rig_data = synth.make_rig(spec)
crowd_in = synth.make_crowd(spec, rig_data, instances=spec.instances)
ctx = api.Context(); rig = api.Rig(ctx, rig_data); crowd = api.Crowd(ctx, rig, spec.instances)
crowd.set_clocks(crowd_in.tau, crowd_in.clip_id); crowd.set_morph_weights(crowd_in.morph_weights)

def run(mask, gap, n=30):
    out = []
    for k in range(n + 5):
        crowd.evaluate(mask)
        ms = crowd.stage_ms()
        if gap: time.sleep(gap)
        if k >= 5: out.append(ms)
    return np.array(out).mean(axis=0)

for name, mask, gap in (("frames back to back", abi.STAGE_ALL, 0), ("skin+aabb only", abi.STAGE_SKIN | abi.STAGE_AABB, 0),
                        ("frames, 5 ms idle between", abi.STAGE_ALL, 0.005), ("skin only, 5 ms idle", abi.STAGE_SKIN | abi.STAGE_AABB, 0.005),
                        ("frames, 20 ms idle between", abi.STAGE_ALL, 0.02), ("frames back to back again", abi.STAGE_ALL, 0)):
    m = run(mask, gap)
    print(f"{name:28s} pose {m[0]:.3f}  skin {m[1]:.3f}  aabb {m[2]:.3f}  span {m[3]:.3f}", flush=True)
