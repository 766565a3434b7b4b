import sys, numpy as np
sys.path.insert(0, ".")
from renderer_amd import abi, api, synth
spec = synth.CONFIGS["c3"]; rig = synth.make_rig(spec); crowd = synth.make_crowd(spec, rig, instances=10000)
for pp in (1, 2):
    ctx = api.Context(pose_path=pp); r = api.Rig(ctx, rig); c = api.Crowd(ctx, r, 10000)
    c.set_clocks(crowd.tau, crowd.clip_id); c.set_morph_weights(crowd.morph_weights)
    for _ in range(5): c.evaluate(abi.STAGE_POSE, sync=True)
    ms = []
    for _ in range(20):
        c.evaluate(abi.STAGE_POSE, sync=True); ms.append(c.stage_ms()[0])
    print("pose_path", pp, "pose ms", round(float(np.median(ms)), 4), flush=True)
    c.close(); r.close(); ctx.close()
