"""How does the skin stage take a VRM-like FACE?  C3's avatar, but the morph targets sit where a face's do: 52 targets
(ARKit's count), all inside the first 15 % of the vertices, each covering a random 2-10 % slice of the mesh inside that
region.  Three weight patterns per instance: (a) the benchmark's (75 % of the weights zero), (b) an idle face — 3 active
targets (blink L / R, one mouth shape), (c) face tracking — all 52 non-zero.  Skin stage from the library's events.
usage: python3 tools/exp_face.py [instances]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import abi, api, synth
from renderer_amd.abi import MorphTarget

I = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
NT = int(os.environ.get("FACE_TARGETS", "52"))  # (more than ~80: face tiles exceed 32 pairs, the limit of k_skin2's per-instance masks)
spec = synth.scaled(synth.CONFIGS["c3"], targets=0)
rig = synth.make_rig(spec)
V = rig.vertex_count
rng = np.random.default_rng(5)
face = int(0.15 * V)
targets = []
for m in range(NT):
    cnt = int(V * rng.uniform(0.02, 0.10))
    start = int(rng.integers(0, face - cnt))
    dp = np.zeros((V, 3), np.float32); dn = np.zeros((V, 3), np.float32); dt = np.zeros((V, 4), np.float32)
    dp[start:start + cnt] = rng.normal(scale=0.02, size=(cnt, 3))
    dn[start:start + cnt] = rng.normal(scale=0.05, size=(cnt, 3))
    dt[start:start + cnt, :3] = rng.normal(scale=0.05, size=(cnt, 3))
    targets.append(MorphTarget(position=dp, normal=dn, tangent=dt))
rig.targets = targets
rig.default_weights = np.zeros(NT, np.float32)
crowd = synth.make_crowd(spec, rig, instances=I)
patterns = {}
w = rng.uniform(0, 1, size=(I, NT)).astype(np.float32)
w[rng.uniform(size=(I, NT)) < 0.75] = 0
patterns["benchmark-like (25 % non-zero)"] = w
w = np.zeros((I, NT), np.float32)
for k in range(3):
    w[np.arange(I), rng.integers(0, NT, size=I)] = rng.uniform(0.1, 1, size=I).astype(np.float32)
patterns["idle face (3 active)"] = w
patterns["face tracking (all 52 active)"] = rng.uniform(0.01, 1, size=(I, NT)).astype(np.float32)
patterns["no morphing (all zero)"] = np.zeros((I, NT), np.float32)
with api.Context() as ctx:
    r = api.Rig(ctx, rig)
    info = r.info
    print(f"tiles {info.tile_count}, morph entries {info.morph_entries}, handed-over tiles {info.handed_over_tiles}", flush=True)
    c = api.Crowd(ctx, r, I)
    c.set_clocks(crowd.tau, crowd.clip_id)
    for name, mw in patterns.items():
        c.set_morph_weights(mw)
        acc = []
        for k in range(25):
            c.evaluate(abi.STAGE_ALL, sync=True)
            if k >= 10:
                acc.append(c.stage_ms()[1])
        print(f"{name:34s}: skin stage {np.mean(acc):.3f} ms (min {np.min(acc):.3f})", flush=True)
    c.close(); r.close()
