#!/usr/bin/env python3
"""Kernel trace of the sequence P, S, S, S (run under rocprofv3 --kernel-trace): which kernel of the first skin stage
after the pose stage is longer than in the following ones?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import abi, api, synth
spec = synth.CONFIGS["c3"]
rig_data = synth.make_rig(spec)
crowd_in = synth.make_crowd(spec, rig_data, instances=spec.instances)
ctx = api.Context(); rig = api.Rig(ctx, rig_data); crowd = api.Crowd(ctx, rig, spec.instances)
crowd.set_clocks(crowd_in.tau, crowd_in.clip_id); crowd.set_morph_weights(crowd_in.morph_weights)
P, S = abi.STAGE_POSE, abi.STAGE_SKIN | abi.STAGE_AABB
for k in range(12):
    for st in (P, S, S, S):
        crowd.evaluate(st)
ctx.synchronize()
