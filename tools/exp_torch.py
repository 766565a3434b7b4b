"""Does torch / RCCL in the process change the skin stage?  (The one-rank torchrun rehearsal of bench.py runs C3's stage
0.25 ms slower than plain bench.py on the same box, with the same confirmed placement.)
usage: python3 tools/exp_torch.py none|torch|rccl [c3]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
name = sys.argv[2] if len(sys.argv) > 2 else "c3"
if mode in ("torch", "rccl"):
    import torch
    torch.cuda.init()
    x = torch.zeros(1 << 20, device="cuda:0")
    torch.cuda.synchronize()
if mode == "rccl":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    dist.all_reduce(x)
    torch.cuda.synchronize()
from renderer_amd import abi, api, synth
spec = synth.CONFIGS[name]
rig = synth.make_rig(spec)
I = spec.instances
crowd = synth.make_crowd(spec, rig, instances=I)
with api.Context() as ctx:
    r = api.Rig(ctx, rig); c = api.Crowd(ctx, r, I)
    c.set_clocks(crowd.tau, crowd.clip_id)
    if crowd.morph_weights is not None: c.set_morph_weights(crowd.morph_weights)
    for rep in range(15): c.evaluate(abi.STAGE_ALL)
    ctx.synchronize()
    ms = []
    t0 = time.perf_counter()
    for rep in range(30):
        c.evaluate(abi.STAGE_ALL)
    ctx.synchronize()
    frame = (time.perf_counter() - t0) / 30 * 1e3
    for rep in range(10):
        c.evaluate(abi.STAGE_ALL); ctx.synchronize(); ms.append(c.stage_ms()[1])
    print(f"{mode:5s} {name}: frame {frame:.3f} ms, skin stage {np.median(ms):.3f} ms, placement {c.placement}", flush=True)
    c.close(); r.close()
if mode == "rccl":
    dist.destroy_process_group()
