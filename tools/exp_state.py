"""Experiment behind api.hip tune_output_placement: the skin stage's speed is a property of the
crowd's output allocations.  With SKINRT_PLACEMENT_TRIALS=0 (no search) the same rig is evaluated
on freshly created crowds, a dummy allocation of random size in front of each so that the arrays
land on other pages; the stage time falls into two groups about 8 % apart (C3 without targets:
~3.85 / ~4.15 ms) and stays put for the life of a crowd."""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, ".")
os.environ["SKINRT_PLACEMENT_TRIALS"] = "0"
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
for k in range(16):
    p = ctypes.c_void_p()
    sz = int(rng.integers(1, 4000)) * (1 << 20) + int(rng.integers(0, 512)) * 4096
    hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(sz))
    c = api.Crowd(ctx, r, inst)
    print(f"crowd {k:2d}: skin stage {measure(c):.3f} ms", flush=True); c.close()
    hip.hipFree(p)
