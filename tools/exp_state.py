"""Experiment: skin-stage speed state vs the addresses of the crowd's buffers."""
import os, sys, numpy as np, ctypes
sys.path.insert(0, ".")
os.environ["SKINRT_DEBUG_PTRS"] = "1"
from renderer_amd import abi, api, synth
spec = synth.scaled(synth.CONFIGS["c3"], targets=0)
inst = 10000
rig = synth.make_rig(spec); crowd = synth.make_crowd(spec, rig, instances=inst)
def measure(c):
    c.set_clocks(crowd.tau, crowd.clip_id)
    for _ in range(8): c.evaluate(sync=True)
    ms = []
    for _ in range(12):
        c.evaluate(sync=True); ms.append(c.stage_ms()[1])
    return float(np.median(ms))
ctx = api.Context(); r = api.Rig(ctx, rig)
hip = ctypes.CDLL("libamdhip64.so")
rng = np.random.default_rng(5)
for k in range(24):
    p = ctypes.c_void_p()
    sz = int(rng.integers(1, 4000)) * (1 << 20) + int(rng.integers(0, 512)) * 4096
    hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(sz))
    c = api.Crowd(ctx, r, inst)
    sys.stderr.flush()
    print(f"state {measure(c):.3f}", flush=True); c.close()
    hip.hipFree(p)
