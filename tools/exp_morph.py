"""Experiment: skin-stage time of the C3 rig as a function of morph-target count / weights."""
import sys, numpy as np
sys.path.insert(0, ".")
from renderer_amd import abi, api, synth
import os
ONLY = os.environ.get("EXP_ONLY", "")
REPS = int(os.environ.get("EXP_REPS", "24"))
def run(spec, inst, zero_w=False, tag=""):
    if ONLY and ONLY not in tag: return
    rig = synth.make_rig(spec); crowd = synth.make_crowd(spec, rig, instances=inst)
    ctx = api.Context(); r = api.Rig(ctx, rig); c = api.Crowd(ctx, r, inst)
    c.set_clocks(crowd.tau, crowd.clip_id)
    if crowd.morph_weights is not None:
        c.set_morph_weights(np.zeros_like(crowd.morph_weights) if zero_w else crowd.morph_weights)
    for _ in range(10): c.evaluate(sync=True)
    ms = []
    for _ in range(REPS):
        c.evaluate(sync=True); ms.append(c.stage_ms()[1])
    info = r.info
    if REPS >= 1000: print("   per 1000:", np.round(np.mean(np.array(ms[:REPS // 1000 * 1000]).reshape(-1, 1000), axis=1), 3))
    print(f"{tag:28s} skin {np.median(ms):6.3f} ms (min {np.min(ms):6.3f})  max_pairs? morph_entries={info.morph_entries}")
    c.close(); r.close(); ctx.close()
base = synth.CONFIGS["c3"]
run(base, 10000, tag="c3 (32 targets)")
run(base, 10000, zero_w=True, tag="c3 all weights zero")
run(synth.scaled(base, targets=12), 10000, tag="c3 12 targets")
run(synth.scaled(base, targets=0), 10000, tag="c3 no targets")
run(synth.scaled(base, targets=32, target_frac=0.03), 10000, tag="c3 32 targets x 3% verts")
