"""One crowd of I instances against the same instances as G crowds of I / G evaluated one after the other on the same
stream (pose, skin, bounds of part 0, then of part 1, ...): does the skin stage run faster when the palettes it reads
were written just before it and are few enough to stay in the memory-side cache (256 MB)?  The crowd-size cliff of
DESIGN.md §3.2 ("measured and not explained") suggests so.  usage: python3 tools/exp_split.py [c3|c5] [parts ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import abi, api, synth

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
# a part count with a trailing "s" = every part's pose stage first, then every part's skin stage and bounds
parts_list = sys.argv[2:] or ["1", "2", "2s", "3", "4", "4s", "1", "2", "2s", "3", "4", "4s"]
spec = synth.CONFIGS[name]
rig = synth.make_rig(spec)
I = spec.instances
whole = synth.make_crowd(spec, rig, instances=I)
with api.Context() as ctx:
    r = api.Rig(ctx, rig)
    def frame(crowds, split):
        if split:
            for c in crowds: c.evaluate(abi.STAGE_POSE)
            for c in crowds: c.evaluate(abi.STAGE_SKIN | abi.STAGE_AABB)
        else:
            for c in crowds: c.evaluate(abi.STAGE_ALL)
    for arg in parts_list:
        split = arg.endswith("s"); G = int(arg.rstrip("s"))
        crowds = []
        for g in range(G):
            lo, n = api.shard_range(I, g, G); hi = lo + n
            c = api.Crowd(ctx, r, hi - lo)
            c.set_clocks(whole.tau[lo:hi], whole.clip_id[lo:hi])
            if whole.morph_weights is not None: c.set_morph_weights(whole.morph_weights[lo:hi])
            crowds.append(c)
        for rep in range(15):
            frame(crowds, split)
        ctx.synchronize()
        K = 30
        t0 = time.perf_counter()
        for rep in range(K):
            frame(crowds, split)
        ctx.synchronize()
        frame_ms = (time.perf_counter() - t0) / K * 1e3
        skin = sum(c.stage_ms()[1] for c in crowds); pose = sum(c.stage_ms()[0] for c in crowds)
        strat = [c.placement.get("strategy") for c in crowds]
        print(f"{name} parts {arg:>3s}: frame {frame_ms:.3f} ms  pose {pose:.3f}  skin (last frame, summed) {skin:.3f}  placement {strat}", flush=True)
        for c in crowds: c.close()
    r.close()
