"""The pose stage against the node count of a SHALLOW skeleton (tree, depth <= 12): the wavefront-per-instance kernel holds a whole
instance's node matrices in LDS (up to 1,024 nodes); above that the split path takes over.  Is there a cliff?
usage: python3 tools/exp_nodes.py [nodes ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import abi, api, synth

I = int(os.environ.get("NODES_INSTANCES", "10000"))
for N in [int(a) for a in sys.argv[1:]] or [260, 512, 1000, 1024, 1100, 2000, 4000]:
    spec = synth.scaled(synth.CONFIGS["c3"], nodes=N, joints=min(N - 10, 200), verts=2560, targets=0, instances=I)
    rig = synth.make_rig(spec)
    crowd = synth.make_crowd(spec, rig, instances=I)
    with api.Context() as ctx:
        r = api.Rig(ctx, rig); c = api.Crowd(ctx, r, I)
        c.set_clocks(crowd.tau, crowd.clip_id)
        for rep in range(10): c.evaluate(abi.STAGE_ALL)
        ctx.synchronize()
        ms = []
        for rep in range(10):
            c.evaluate(abi.STAGE_ALL); ctx.synchronize(); ms.append(c.stage_ms()[0])
        info = r.info
        gb = I * N * (48 * 3 + 12) / 1e9  # locals out and in, world matrices, key indices (palette: 200 joints)
        print(f"nodes {N:5d} (levels {info.level_count:3d}): pose stage {np.median(ms):.3f} ms for {I} instances = {np.median(ms) * 1e3 / N:.3f} us per node-thousand-instances... {gb / (np.median(ms) * 1e-3):.0f} GB/s of its own bytes", flush=True)
        c.close(); r.close()
