"""The stages against the number of clips the crowd's instances play (C3; instances pick clips at random): lanes of a sampling wave
that play different clips read different channel tables.   usage: python3 tools/exp_clips.py"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
from renderer_amd import abi, api, synth
for clips in (1, 2, 4, 8, 16):
    spec = synth.scaled(synth.CONFIGS["c3"], clips=clips)
    rig = synth.make_rig(spec)
    I = spec.instances
    crowd = synth.make_crowd(spec, rig, instances=I)
    with api.Context() as ctx:
        r = api.Rig(ctx, rig); c = api.Crowd(ctx, r, I)
        c.set_clocks(crowd.tau, crowd.clip_id); c.set_morph_weights(crowd.morph_weights)
        for rep in range(10): c.evaluate(abi.STAGE_ALL)
        ctx.synchronize()
        ms = []
        for rep in range(10):
            c.evaluate(abi.STAGE_ALL); ctx.synchronize(); ms.append(c.stage_ms())
        m = np.median(np.array(ms), axis=0)
        print(f"clips {clips:2d} (instances spread over them: {len(np.unique(crowd.clip_id))} in use): pose {m[0]:.3f} ms, skin {m[1]:.3f} ms", flush=True)
        c.close(); r.close()
