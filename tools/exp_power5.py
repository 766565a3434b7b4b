#!/usr/bin/env python3
"""Is the first-skin-after-pose penalty a transient (constant in ms) or a steady-state property (proportional)?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import abi, api, synth
spec = synth.CONFIGS["c3"]
rig_data = synth.make_rig(spec)
P, S = abi.STAGE_POSE, abi.STAGE_SKIN | abi.STAGE_AABB
ctx = api.Context(); rig = api.Rig(ctx, rig_data)
for inst in [int(x) for x in os.environ.get("INSTS", "2500,5000,10000,20000,40000").split(",")]:
    crowd_in = synth.make_crowd(spec, rig_data, instances=inst)
    crowd = api.Crowd(ctx, rig, inst)
    crowd.set_clocks(crowd_in.tau, crowd_in.clip_id); crowd.set_morph_weights(crowd_in.morph_weights)
    res = {}
    for name, steps in (("ALL", [abi.STAGE_ALL]), ("S", [S])):
        acc = []
        for k in range(18):
            for st in steps:
                crowd.evaluate(st); ms = crowd.stage_ms()
            if k >= 3: acc.append(ms[1])
        res[name] = np.mean(acc)
    print(f"I={inst:6d}  skin in frames {res['ALL']:.3f}  skin alone {res['S']:.3f}  diff {res['ALL'] - res['S']:.3f} ms = {100 * (res['ALL'] / res['S'] - 1):.1f} %", flush=True)
    crowd.close()
