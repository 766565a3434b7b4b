"""A frame loop that never blocks the host: stream-ordered clock updates from device
memory, clocks advanced on the device with the wrap rule of SURVEY.md §8a a4, and the morph weights a skin stage reads
after a setter that was not followed by a pose stage."""
import ctypes as C
import time

import numpy as np
import pytest

from renderer_amd import abi, api, synth
from tests.helpers import DeviceBuffers, assert_close, assert_equal, run_oracle

pytestmark = pytest.mark.gpu


def _rig(seed=4, **kw):
    spec = synth.tiny_spec(seed=seed, verts=600, joints=10, nodes=16, clips=3, keys=6, targets=4, **kw)
    return spec, synth.make_rig(spec)


def test_advance_clocks_wraps_like_fmodf_bit_for_bit(oracle32):
    """tau = fmodf(tau + dt * rate, clip_length): product and sum rounded once each in fp32, then the exact remainder —
    numpy's float32 fmod is C's fmodf.  Instances at rest keep their clock.  After many frames the crowd's pose equals
    the oracle's at the host-side replay of the same clocks."""
    spec, rig = _rig()
    n = 257
    crowd = synth.make_crowd(spec, rig, instances=n)
    crowd.clip_id[5] = abi.NO_CLIP
    crowd.clip_id[6] = 77          # out of range = at rest too (the async setter's rule)
    lens = np.array([rig.clip_length(c) for c in range(len(rig.clips))], np.float32)
    rate = np.random.default_rng(1).uniform(0.25, 3.0, size=n).astype(np.float32)
    dev = DeviceBuffers()
    drate = dev.upload(rate)
    with api.Context(exact_arithmetic=1) as ctx:
        r = api.Rig(ctx, rig)
        c = api.Crowd(ctx, r, n)
        clip = crowd.clip_id.copy()
        clip[6] = 0                 # the blocking setter validates; put the out-of-range id in through the async one below
        c.set_clocks(crowd.tau, clip)
        dclip = dev.upload(crowd.clip_id)
        dtau = dev.upload(crowd.tau)
        c.set_clocks_async(dtau, dclip)
        tau = crowd.tau.copy()
        active = crowd.clip_id < len(rig.clips)
        ln = lens[np.where(active, crowd.clip_id, 0)]
        for frame in range(40):
            dt = np.float32(0.0166667 * (1 + frame % 3))
            use_rate = frame % 2 == 0
            c.advance_clocks(float(dt), drate if use_rate else 0)
            step = (dt * rate).astype(np.float32) if use_rate else np.full(n, dt, np.float32)
            nxt = np.fmod((tau + step).astype(np.float32), ln).astype(np.float32)
            tau = np.where(active, nxt, tau)
            if frame in (0, 17, 39):
                assert_equal(c.clocks().view(np.uint32), tau.view(np.uint32), f"clocks after frame {frame}")
        assert (tau[active] < ln[active]).all() and (tau != crowd.tau)[active].all()
        assert tau[5] == crowd.tau[5] and tau[6] == crowd.tau[6]
        c.set_morph_weights(crowd.morph_weights)
        c.evaluate(abi.STAGE_ALL, sync=True)
        ref = oracle32.evaluate(rig, tau, crowd.clip_id, crowd.morph_weights)
        assert_equal(c.keyframe_index(), ref.key_index, "key_index")
        assert_equal(c.palette(), ref.palette_rows(), "palette")
        assert_equal(c.positions(), ref.position, "position")
        c.close(); r.close()
    dev.close()


def test_async_clock_setter_takes_device_memory_and_does_not_wait(oracle32):
    spec, rig = _rig(seed=8)
    n = 4096
    crowd = synth.make_crowd(spec, rig, instances=n)
    other = synth.make_crowd(synth.scaled(spec, seed=99), rig, instances=n)
    dev = DeviceBuffers()
    with api.Context() as ctx:
        r = api.Rig(ctx, rig)
        c = api.Crowd(ctx, r, n)
        c.set_morph_weights(crowd.morph_weights)
        d0, c0 = dev.upload(crowd.tau), dev.upload(crowd.clip_id)
        d1 = dev.upload(other.tau)
        # two frames enqueued back to back without a host wait: each evaluate sees the clocks set just before it
        c.set_clocks_async(d0, c0)
        c.evaluate(abi.STAGE_ALL)
        a0 = c.aabbs().copy()          # download synchronises
        c.set_clocks_async(d1)          # clip ids stay
        c.evaluate(abi.STAGE_ALL)
        a1 = c.aabbs().copy()
        assert_equal(c.clocks(), other.tau, "clocks")
        c.set_clocks(crowd.tau, crowd.clip_id)
        c.evaluate(abi.STAGE_ALL, sync=True)
        assert_equal(c.aabbs().view(np.uint32), a0.view(np.uint32), "frame 0: async setter == blocking setter")
        c.set_clocks(other.tau, crowd.clip_id)
        c.evaluate(abi.STAGE_ALL, sync=True)
        assert_equal(c.aabbs().view(np.uint32), a1.view(np.uint32), "frame 1")
        assert not np.array_equal(a0, a1)
        with pytest.raises(api.SkinrtError):
            c.set_clocks_async(0)
        c.close(); r.close()
    dev.close()


def test_skin_stage_without_pose_sees_new_morph_weights_on_a_weights_channel_rig(oracle32):
    """on rigs with `weights` channels the skin stage reads the sampled copy of the weights; a setter followed
    by evaluate(SKIN | AABB) — no pose stage — must not skin with the stale copy (instances whose clip has no weights
    channel take the host's weights)."""
    spec, rig = _rig(seed=12, weight_channels=1)
    n = 64
    crowd = synth.make_crowd(spec, rig, instances=n)
    assert (crowd.clip_id != 0).any() and (crowd.clip_id == 0).any()
    new_w = np.random.default_rng(5).random((n, rig.target_count), dtype=np.float32)
    with api.Context(exact_arithmetic=1) as ctx:
        r = api.Rig(ctx, rig)
        assert r.info.weight_channel_clips == 1
        c = api.Crowd(ctx, r, n)
        c.set_clocks(crowd.tau, crowd.clip_id)
        c.set_morph_weights(crowd.morph_weights)
        c.evaluate(abi.STAGE_ALL, sync=True)
        first = c.positions().copy()
        c.set_morph_weights(new_w)
        c.evaluate(abi.STAGE_SKIN | abi.STAGE_AABB, sync=True)
        ref = oracle32.evaluate(rig, crowd.tau, crowd.clip_id, new_w)
        assert_equal(c.positions(), ref.position, "positions after set_morph_weights + SKIN")
        assert_equal(c.aabbs(), ref.aabb, "bounds")
        assert not np.array_equal(first, ref.position)
        # defaults restored, again without a pose stage
        c.set_morph_weights(None)
        c.evaluate(abi.STAGE_SKIN, sync=True)
        ref = oracle32.evaluate(rig, crowd.tau, crowd.clip_id, None)
        assert_equal(c.positions(), ref.position, "positions after restoring the defaults")
        c.close(); r.close()


def test_frame_loop_with_device_side_clocks_keeps_the_host_ahead():
    """evaluate + advance_clocks per frame, nothing waits: the host enqueues N frames in far less time than the GPU needs
    to run them (a blocking setter per frame would hold it to the GPU's pace)."""
    spec = synth.scaled(synth.CONFIGS["c3"], instances=3000)
    rig = synth.make_rig(spec)
    crowd = synth.make_crowd(spec, rig, instances=spec.instances)
    with api.Context() as ctx:
        r = api.Rig(ctx, rig)
        c = api.Crowd(ctx, r, spec.instances)
        c.set_clocks(crowd.tau, crowd.clip_id)
        c.set_morph_weights(crowd.morph_weights)
        for _ in range(3):
            c.evaluate(abi.STAGE_ALL, sync=True)
        frames = 30
        t0 = time.perf_counter()
        for _ in range(frames):
            c.advance_clocks(1.0 / 60.0)
            c.evaluate(abi.STAGE_ALL)
        t_enqueue = time.perf_counter() - t0
        ctx.synchronize()
        t_total = time.perf_counter() - t0
        gpu_ms = c.stage_ms()[3]
        print(f"{frames} frames: enqueued in {t_enqueue * 1e3:.2f} ms, finished after {t_total * 1e3:.2f} ms; last frame {gpu_ms:.3f} ms on the GPU")
        assert t_enqueue < 0.5 * t_total
        c.close(); r.close()


@pytest.mark.parametrize("long_lists", [False, True])
def test_a_frame_is_capturable_into_a_hip_graph_and_replays_bit_for_bit(long_lists):
    """A caller with a graph-driven loop: advance_clocks + evaluate(ALL) recorded once with hipStreamBeginCapture on the
    context's stream (kernels, a memset, events and the fork / join over the library's side stream are all capturable;
    nothing is allocated once the crowd has run a frame) and replayed; clocks, vertices and bounds after five replays
    equal those after five frames of plain stream launches.  long_lists: tiles of about 60 joints — the skin kernel's
    blocks then take more than 64 KiB of LDS, which the launch announces with hipFuncSetAttribute (inside the capture too)."""
    if long_lists:
        spec = synth.scaled(synth.tiny_spec(seed=326), nodes=80, joints=64, verts=1400, sets=1, targets=10, target_mode="dense",
                            joint_spread=26, clips=2, keys=6, max_depth=10)
        rig = synth.make_rig(spec)
    else:
        spec, rig = _rig(seed=21)
    n = 513
    crowd = synth.make_crowd(spec, rig, instances=n)
    dev = DeviceBuffers()
    hip = dev.hip
    dt = 1.0 / 60.0
    with api.Context() as ctx:
        r = api.Rig(ctx, rig)
        c = api.Crowd(ctx, r, n)
        assert not long_lists or r.info.max_tile_joints > 56, r.info.max_tile_joints
        c.set_morph_weights(crowd.morph_weights)
        c.set_clocks(crowd.tau, crowd.clip_id)
        for _ in range(5):
            c.advance_clocks(dt)
            c.evaluate(abi.STAGE_ALL)
        want = (c.clocks().copy(), c.positions().copy(), c.normals().copy(), c.aabbs().copy())
        c.set_clocks(crowd.tau, crowd.clip_id)
        ctx.synchronize()
        stream = C.c_void_p(ctx.stream)
        graph, gexec = C.c_void_p(), C.c_void_p()
        assert hip.hipStreamBeginCapture(stream, C.c_int(2)) == 0          # hipStreamCaptureModeRelaxed
        c.advance_clocks(dt)
        c.evaluate(abi.STAGE_ALL)
        assert hip.hipStreamEndCapture(stream, C.byref(graph)) == 0
        assert hip.hipDeviceSynchronize() == 0
        assert_equal(c.clocks(), crowd.tau, "capturing runs nothing")
        assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)) == 0
        for _ in range(5):
            assert hip.hipGraphLaunch(gexec, stream) == 0
        ctx.synchronize()
        got = (c.clocks(), c.positions(), c.normals(), c.aabbs())
        for name, g, w in zip(("clocks", "position", "normal", "aabb"), got, want):
            assert_equal(np.ascontiguousarray(g).view(np.uint32), np.ascontiguousarray(w).view(np.uint32), name)
        hip.hipGraphExecDestroy(gexec)
        hip.hipGraphDestroy(graph)
        c.close(); r.close()
    dev.close()
