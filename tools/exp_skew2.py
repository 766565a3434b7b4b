"""Experiment: is the slow state tied to the bounds stores?  (fresh process = slow state)"""
import os, sys, numpy as np
sys.path.insert(0, ".")
from renderer_amd import abi, api, synth
os.environ["SKINRT_OUT_OFFS"] = "0,0"
spec = synth.scaled(synth.CONFIGS["c3"], targets=0)
inst = 10000
rig = synth.make_rig(spec); crowd = synth.make_crowd(spec, rig, instances=inst)
ctx = api.Context(); r = api.Rig(ctx, rig)
for rep in range(4):
    c = api.Crowd(ctx, r, inst)
    c.set_clocks(crowd.tau, crowd.clip_id)
    def t(mask, n=4):
        ms = []
        for _ in range(n):
            c.evaluate(mask, sync=True); ms.append(c.stage_ms()[1])
        return np.mean(ms[1:])
    a = t(abi.STAGE_ALL)
    b = t(abi.STAGE_POSE | abi.STAGE_SKIN)
    d = t(abi.STAGE_ALL)
    print(f"crowd {rep}: with bounds {a:.3f}  without bounds {b:.3f}  with bounds {d:.3f}", flush=True)
    c.close()
