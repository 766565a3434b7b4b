import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from renderer_amd import abi, api, synth
spec = synth.CONFIGS["c5"]
rig = synth.make_rig(spec)
I = 1000
crowd = synth.make_crowd(spec, rig, instances=I)
ctx = api.Context(); r = api.Rig(ctx, rig); c = api.Crowd(ctx, r, I)
c.set_clocks(crowd.tau, crowd.clip_id); c.set_morph_weights(crowd.morph_weights)
outs = []
for rep in range(3):
    c.evaluate(sync=True)
    outs.append(np.stack([c.positions(i, 1)[0] for i in (0, 1, 2, 255, 256, 511, 999)]))
for rep in (0, 1):
    d = (outs[rep].view(np.uint32) != outs[2].view(np.uint32)).any(axis=2)   # [inst][V]
    for k, i in enumerate((0, 1, 2, 255, 256, 511, 999)):
        bad = np.nonzero(d[k])[0]
        tiles = np.unique(bad // 256)
        print(f"eval {rep} vs eval 2: instance {i}: {bad.size} vertices differ in {tiles.size} tiles", tiles[:12],
              "tile%8 histogram", np.bincount(tiles % 8, minlength=8), flush=True)
mw = crowd.morph_weights
print("active targets of instance 0:", np.nonzero(mw[0])[0].size, "of", mw.shape[1])
