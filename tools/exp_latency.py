"""Frame latency of small crowds (launch-bound): evaluate + synchronize, wall clock and device span.
usage: python3 tools/exp_latency.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import abi, api, synth

for name, I in (("c1", 1), ("c1", 64), ("c2", 4096), ("c3", 16)):
    spec = synth.CONFIGS[name]
    rig = synth.make_rig(spec)
    crowd = synth.make_crowd(spec, rig, instances=I)
    with api.Context() as ctx:
        r = api.Rig(ctx, rig); c = api.Crowd(ctx, r, I)
        c.set_clocks(crowd.tau, crowd.clip_id)
        if crowd.morph_weights is not None: c.set_morph_weights(crowd.morph_weights)
        for rep in range(20): c.evaluate(sync=True)
        wall, span = [], []
        for rep in range(200):
            t0 = time.perf_counter(); c.evaluate(sync=True); wall.append(time.perf_counter() - t0)
            span.append(c.stage_ms()[3])
        t0 = time.perf_counter()
        for rep in range(200): c.evaluate()
        ctx.synchronize(); burst = (time.perf_counter() - t0) / 200
        print(f"{name} x {I:5d}: evaluate+sync wall median {np.median(wall)*1e3:.3f} ms, device span {np.median(span):.3f} ms, "
              f"200 frames enqueued back to back: {burst*1e3:.3f} ms per frame", flush=True)
        c.close(); r.close()
