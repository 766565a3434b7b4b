"""bench.py's own frame loop (FrameLoop: step / fence / bounds exchange / max-over-ranks timing) driven by two `gloo`
ranks on the CPU with the ORACLE standing in for crowd.evaluate (until now only shard.py was covered,
not the bench's wiring).  Also: the parent of the ranks counts GPUs without opening one, and the CPU baseline sizes its
thread pool from the box's share of the host."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["SKINRT_ROOT"])
import bench
from oracle.oracle import Oracle
from renderer_amd import abi, shard, synth

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
I = int(os.environ["SKINRT_I"])
spec = synth.tiny_spec(seed=22)
rig = synth.make_rig(spec)
lens = np.array([rig.clip_length(c) for c in range(len(rig.clips))], np.float32)
lo, hi = shard.shard_range(I, rank, world)
part = synth.make_crowd(spec, rig, instances=hi - lo, first=lo)
orc = Oracle("f32")


class OracleCrowd:
    """what FrameLoop needs of api.Crowd, computed by the oracle into a torch CPU tensor"""
    def __init__(self):
        self.tau = part.tau.copy()
        self.aabb = torch.zeros((hi - lo, 6))
        self.frames = self.advances = 0
    def advance_clocks(self, dt):
        self.tau = np.fmod((self.tau + np.float32(dt)).astype(np.float32), lens[part.clip_id]).astype(np.float32)
        self.advances += 1
    def evaluate(self, stages):
        assert stages == abi.STAGE_ALL
        if hi > lo:
            res = orc.evaluate(rig, self.tau, part.clip_id, part.morph_weights, want=("aabb",))
            self.aabb.copy_(torch.from_numpy(res.aabb.reshape(hi - lo, 6)))
        self.frames += 1


class Ctx:
    stream = 0
    def synchronize(self):
        pass


crowd = OracleCrowd()
loop = bench.FrameLoop(crowd, Ctx(), crowd.aabb, I, world, launched=True)
for _ in range(2):
    loop.step()
stamps = []
dt = loop.timed(3, per_step=stamps.append)
assert stamps == [0, 1, 2] and crowd.frames == 5 and crowd.advances == 0 and dt > 0
dt_k = loop.timed(2, gather=False)
assert crowd.frames == 7
whole = synth.make_crowd(spec, rig, instances=I)
ref = orc.evaluate(rig, whole.tau, whole.clip_id, whole.morph_weights, want=("aabb",)).aabb.reshape(I, 6)
got = loop.gathered_bounds().numpy()
assert got.shape == (I, 6) and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "gathered bounds of the last exchanged frame"
# frames that advance the clocks first: every rank's clocks move, the gathered bounds follow
dt_a = loop.timed(2, advance=True)
assert crowd.advances == 2 and crowd.frames == 9
tau = whole.tau.copy()
for _ in range(2):
    tau = np.fmod((tau + np.float32(1.0 / 60.0)).astype(np.float32), lens[whole.clip_id]).astype(np.float32)
ref2 = orc.evaluate(rig, tau, whole.clip_id, whole.morph_weights, want=("aabb",)).aabb.reshape(I, 6)
assert np.array_equal(loop.gathered_bounds().numpy().view(np.uint32), ref2.view(np.uint32))
assert not np.array_equal(ref, ref2)
# the time every rank reports is the slowest rank's
t = loop.max_over_ranks(1.0 + rank)
assert t == float(world)
if rank == 0:
    print("BENCH_LOOP_OK", I, world)
dist.barrier()
dist.destroy_process_group()
'''


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_frame_loop_with_two_gloo_ranks_and_the_oracle_as_the_crowd(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, SKINRT_ROOT=ROOT, SKINRT_I="9", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "BENCH_LOOP_OK 9 2" in p.stdout


def test_frame_loop_without_a_launcher_has_no_exchange():
    sys.path.insert(0, ROOT)
    import bench

    class Crowd:
        frames = 0
        def evaluate(self, stages):
            self.frames += 1

    class Ctx:
        stream = 0
        def synchronize(self):
            pass

    c = Crowd()
    loop = bench.FrameLoop(c, Ctx(), None, 5, 1, launched=False)
    assert loop.timed(4) > 0 and c.frames == 4 and loop.gathered_bounds() is None and loop.max_over_ranks(2.5) == 2.5


def test_parent_counts_gpus_without_opening_one(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    n = bench.visible_gpus()
    assert isinstance(n, int) and n >= 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert bench.visible_gpus() <= 1
    src = open(os.path.join(ROOT, "bench.py")).read()
    spawn = src[src.index("def spawn_ranks"):src.index("def main")]
    assert "import torch" not in spawn and "torch.cuda" not in spawn, "the parent of the ranks must not touch torch.cuda"


def test_cpu_baseline_threads_follow_the_box_share(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("OMP_NUM_THREADS", "3")
    assert bench.host_threads() == min(3, os.cpu_count())
    monkeypatch.setenv("OMP_NUM_THREADS", "100000")
    assert bench.host_threads() == os.cpu_count()
    monkeypatch.delenv("OMP_NUM_THREADS")
    assert bench.host_threads() == os.cpu_count()
