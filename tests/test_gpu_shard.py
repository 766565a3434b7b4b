"""The N>1 exchange step rehearsed on ONE GPU: a world-size-1 `nccl` (= RCCL) process group
all-gathers the library's own device buffers (zero-copy torch views of hipMalloc'ed memory via
__cuda_array_interface__), exactly as bench.py does per frame at N>1.  Checks the view, the
stream hand-off and the collective against the host download; byte-identical."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["SKINRT_ROOT"])
from renderer_amd import abi, api, shard, synth

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
spec = synth.tiny_spec(seed=41, verts=300)
rig = synth.make_rig(spec)
I = 7
crowd = synth.make_crowd(spec, rig, instances=I)
with api.Context(device=0) as ctx:
    r = api.Rig(ctx, rig); c = api.Crowd(ctx, r, I)
    c.set_clocks(crowd.tau, crowd.clip_id); c.set_morph_weights(crowd.morph_weights)
    c.evaluate(abi.STAGE_ALL)
    ctx.synchronize()                      # library stream -> torch stream hand-off (as bench.py)
    ap, an = c.device_pointer("aabbs"); pp, pn = c.device_pointer("positions")
    assert an == I * 6 and pn == I * rig.vertex_count * 3
    a_t = torch.as_tensor(shard.DeviceArray(ap, (I, 6)), device="cuda:0")
    p_t = torch.as_tensor(shard.DeviceArray(pp, (I, rig.vertex_count * 3)), device="cuda:0")
    assert a_t.data_ptr() == ap and p_t.data_ptr() == pp, "view must be zero-copy"
    ga = shard.all_gather_rows(a_t, I, 1); gp = shard.all_gather_rows(p_t, I, 1)
    torch.cuda.synchronize()
    assert ga.data_ptr() != ap, "the collective must have produced a new buffer"
    assert np.array_equal(ga.cpu().numpy().view(np.uint32), c.aabbs().reshape(I, 6).view(np.uint32))
    assert np.array_equal(gp.cpu().numpy().view(np.uint32), c.positions().reshape(I, -1).view(np.uint32))
    # the per-frame exchange object: RCCL on a side stream behind an event of the library's stream, three frames
    # enqueued back to back with changing clocks and no host synchronisation in between; result() = the last frame's
    ex = shard.OverlappedRowGather(a_t, I, 1, ctx.stream)
    for frame in range(3):
        c.set_clocks(crowd.tau * (0.5 + 0.25 * frame), crowd.clip_id)
        ex.before_frame(); c.evaluate(abi.STAGE_ALL); ex.after_frame()
    got = ex.result().cpu().numpy()
    ctx.synchronize()
    assert np.array_equal(got.view(np.uint32), c.aabbs().reshape(I, 6).view(np.uint32)), "overlapped gather of the last frame"
    c.close(); r.close()
dist.destroy_process_group()
print("RCCL_GATHER_OK")
'''


@pytest.mark.gpu
def test_rccl_gather_of_library_buffers_world_size_1(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SKINRT_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "RCCL_GATHER_OK" in p.stdout
