"""Instance sharding, CPU only.  The N>1 path is: every rank builds the same rig,
draws ITS instance range of the crowd, evaluates it alone (no data-path collective), then one
all-gather assembles per-instance rows.  Here two `gloo` ranks stand in for two GPUs and the
oracle stands in for the kernels; the gathered result must be byte-identical to one rank
evaluating the whole crowd."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from renderer_amd import shard, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("instances,world", [(10, 2), (11, 2), (1, 2), (0, 4), (100000, 8), (7, 8), (8, 8), (9, 8)])
def test_shard_ranges_partition_the_crowd(instances, world):
    per = shard.padded_count(instances, world)
    assert per * world >= instances
    seen = []
    for r in range(world):
        lo, hi = shard.shard_range(instances, r, world)
        assert 0 <= lo <= hi <= instances and hi - lo <= per
        seen += list(range(lo, hi))
    assert seen == list(range(instances))


@pytest.mark.parametrize("instances,world", [(10, 2), (11, 2), (1, 2), (0, 4), (100000, 8), (7, 8), (9, 8), (4000000000, 3)])
def test_c_abi_shard_rule_is_the_python_rule(instances, world):
    from renderer_amd import abi, api
    for r in range(world):
        lo, hi = shard.shard_range(instances, r, world)
        assert api.shard_range(instances, r, world) == (lo, hi - lo)
    with pytest.raises(api.SkinrtError) as e:
        api.shard_range(10, 2, 2)
    assert e.value.code == abi.E_RANGE


def test_crowd_shards_draw_the_same_streams_as_the_whole_crowd():
    spec = synth.tiny_spec(seed=21)
    rig = synth.make_rig(spec)
    whole = synth.make_crowd(spec, rig, instances=37, with_root=True)
    for world in (2, 3, 8):
        for r in range(world):
            lo, hi = shard.shard_range(37, r, world)
            part = synth.make_crowd(spec, rig, instances=hi - lo, first=lo, with_root=True)
            assert np.array_equal(part.tau, whole.tau[lo:hi])
            assert np.array_equal(part.clip_id, whole.clip_id[lo:hi])
            assert np.array_equal(part.morph_weights, whole.morph_weights[lo:hi])
            assert np.array_equal(part.root, whole.root[lo:hi])


WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["SKINRT_ROOT"])
from oracle.oracle import Oracle
from renderer_amd import shard, synth

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
I = int(os.environ["SKINRT_I"])
spec = synth.tiny_spec(seed=22)
rig = synth.make_rig(spec)
lo, hi = shard.shard_range(I, rank, world)
part = synth.make_crowd(spec, rig, instances=hi - lo, first=lo)
orc = Oracle("f32")
res = orc.evaluate(rig, part.tau, part.clip_id, part.morph_weights, want=("position", "aabb")) if hi > lo else None
V = rig.vertex_count
aabb = torch.from_numpy(res.aabb.reshape(hi - lo, 6)) if res is not None else torch.zeros((0, 6))
pos = torch.from_numpy(res.position) if res is not None else torch.zeros((0, V, 3))
all_aabb = shard.all_gather_rows(aabb, I, world)
all_pos = shard.all_gather_rows(pos, I, world)
if rank == 0:
    whole = synth.make_crowd(spec, rig, instances=I)
    ref = orc.evaluate(rig, whole.tau, whole.clip_id, whole.morph_weights, want=("position", "aabb"))
    assert all_aabb.shape == (I, 6) and all_pos.shape == (I, V, 3)
    assert np.array_equal(all_aabb.numpy().view(np.uint32), ref.aabb.reshape(I, 6).view(np.uint32))
    assert np.array_equal(all_pos.numpy().view(np.uint32), ref.position.view(np.uint32))
    print("GATHER_OK", I, world)
# the per-frame exchange object of bench.py (on the CPU: no streams, the gather runs in after_frame), two frames in a row
ex = shard.OverlappedRowGather(aabb, I, world)
for frame in range(2):
    ex.before_frame()
    ex.after_frame()
    if rank == 0:
        assert np.array_equal(ex.result().numpy().view(np.uint32), ref.aabb.reshape(I, 6).view(np.uint32))
if rank == 0:
    print("OVERLAPPED_OK", I, world)
# the chunked gather of several attributes through one bounded buffer (bench.py's vertex all-gather)
per = shard.padded_count(I, world)
flat = pos.reshape(hi - lo, V * 3)
got = [np.zeros((world * per, V * 3), np.float32) for _ in range(2)]
def consume(k, r0, r1, view):
    for g in range(world):
        got[k][g * per + r0: g * per + r1] = view[g].numpy()
moved = shard.all_gather_rows_chunked([flat, flat * 2.0], I, world, budget_bytes=world * V * 12 * 2, consume=consume)
assert moved == 2 * world * per * V * 12
if rank == 0:
    assert np.array_equal(got[0][:I].view(np.uint32), ref.position.reshape(I, V * 3).view(np.uint32))
    assert np.array_equal(got[1][:I], ref.position.reshape(I, V * 3) * 2.0)
    print("CHUNKED_OK", I, world)
dist.barrier()
dist.destroy_process_group()
'''


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("instances", [9, 2])   # ragged last shard; one instance per rank
def test_world_size_2_gather_is_byte_identical(instances, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, SKINRT_ROOT=ROOT, SKINRT_I=str(instances), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert f"GATHER_OK {instances} 2" in p.stdout
    assert f"CHUNKED_OK {instances} 2" in p.stdout
    assert f"OVERLAPPED_OK {instances} 2" in p.stdout
