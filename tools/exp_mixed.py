"""C3's avatar with a FRACTION of its tiles bound loosely (joints drawn from a window of `window` bones): most tiles keep their
8-joint lists, some get long ones.  What does sizing k_skin2's palette ring by the long lists cost the short tiles, against
sending the long tiles to the loader-wave kernel?   usage: python3 tools/exp_mixed.py [percent ...]   (env MIX_WINDOW, default 36)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import abi, api, synth

window = int(os.environ.get("MIX_WINDOW", "36"))
percents = [float(a) for a in sys.argv[1:]] or [0, 1, 3, 5, 10, 25]
base_spec = synth.CONFIGS["c3"]
for pc in percents:
    rig = synth.make_rig(base_spec)
    rng = np.random.default_rng(17)
    V = rig.vertex_count
    tiles = (V + 255) // 256
    pick = rng.choice(tiles, size=int(round(tiles * pc / 100.0)), replace=False)
    for t in pick:
        v0, v1 = t * 256, min(V, (t + 1) * 256)
        lo = int(rng.integers(0, rig.joint_count - window))
        rig.joints[0][v0:v1] = rng.integers(lo, lo + window, size=(v1 - v0, 4)).astype(np.uint16)
    I = base_spec.instances
    crowd = synth.make_crowd(base_spec, rig, instances=I)
    with api.Context() as ctx:
        r = api.Rig(ctx, rig); c = api.Crowd(ctx, r, I)
        info = r.info
        c.set_clocks(crowd.tau, crowd.clip_id)
        c.set_morph_weights(crowd.morph_weights)
        for rep in range(12): c.evaluate(abi.STAGE_ALL)
        ctx.synchronize()
        ms = []
        for rep in range(12):
            c.evaluate(abi.STAGE_ALL); ctx.synchronize(); ms.append(c.stage_ms()[1])
        print(f"{pc:5.1f} % of the tiles with a {window}-bone window: max tile joints {info.max_tile_joints:3d}, handed over {info.handed_over_tiles}: skin stage {np.median(ms):.3f} ms", flush=True)
        c.close(); r.close()
