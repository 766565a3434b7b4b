"""C3's avatar with wider joint neighbourhoods per vertex (SynthSpec.joint_spread: influences drawn from the primary bone
+- spread): how does the skin stage take tiles with longer joint lists — k_skin2's ring up to 21 joints, the loader-wave
kernel k_skin above?   usage: python3 tools/exp_spread.py [spread ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import abi, api, synth

targets = int(os.environ.get("SPREAD_TARGETS", "32"))
spreads = [int(a) for a in sys.argv[1:]] or [3, 6, 9, 12, 16]
for sp in spreads:
    spec = synth.scaled(synth.CONFIGS["c3"], joint_spread=sp, targets=targets)
    rig = synth.make_rig(spec)
    I = spec.instances
    crowd = synth.make_crowd(spec, rig, instances=I)
    with api.Context() as ctx:
        r = api.Rig(ctx, rig); c = api.Crowd(ctx, r, I)
        info = r.info
        c.set_clocks(crowd.tau, crowd.clip_id)
        if crowd.morph_weights is not None: c.set_morph_weights(crowd.morph_weights)
        for rep in range(12): c.evaluate(abi.STAGE_ALL)
        ctx.synchronize()
        ms = []
        for rep in range(12):
            c.evaluate(abi.STAGE_ALL); ctx.synchronize(); ms.append(c.stage_ms()[1])
        print(f"joint_spread {sp:2d}: max tile joints {info.max_tile_joints:3d}, lds tiles {info.lds_tiles}/{info.tile_count}, handed over {info.handed_over_tiles}: "
              f"skin stage {np.median(ms):.3f} ms", flush=True)
        c.close(); r.close()
