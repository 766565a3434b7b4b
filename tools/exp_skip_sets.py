"""What stopping a wave after its last used weight set is worth (wave_influences, kernels_skin.hip): a C5-shaped rig
(8 influences, exact arithmetic) whose second weight set is empty on a fraction of the mesh, skin stage timed with the
kernels that test for it (SKINRT_SKIP_SETS=1) and without (=0).  usage: python3 tools/exp_skip_sets.py"""
import os, subprocess, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def child(frac, skip, targets):
    from renderer_amd import api, synth
    spec = synth.scaled(synth.CONFIGS["c5"], verts=250_000, joints=1024, nodes=1024, targets=targets)
    rig = synth.make_rig(spec)
    V = rig.vertex_count
    rig.weights[1][: int(V * frac)] = 0.0
    I = 2000
    crowd = synth.make_crowd(spec, rig, instances=I)
    with api.Context(exact_arithmetic=1) as ctx:
        r = api.Rig(ctx, rig); c = api.Crowd(ctx, r, I)
        c.set_clocks(crowd.tau, crowd.clip_id)
        if crowd.morph_weights is not None: c.set_morph_weights(crowd.morph_weights)
        ms = []
        for rep in range(8):
            c.evaluate(sync=True); ms.append(c.stage_ms()[1])
        print(json.dumps({"targets": targets, "empty_fraction": frac, "skip": skip, "kernels_chosen": int(r.info.skip_weight_sets),
                          "skin_ms": round(float(np.median(ms[2:])), 3)}), flush=True)
        c.close(); r.close()

if len(sys.argv) == 4:
    child(float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
else:
  for targets in (0, 16):
    for frac in (0.0, 0.5, 0.9, 1.0):
        for skip in (0, 1):
            subprocess.run([sys.executable, __file__, str(frac), str(skip), str(targets)], env=dict(os.environ, SKINRT_SKIP_SETS=str(skip)), check=True, timeout=300)
