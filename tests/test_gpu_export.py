"""Reference-layout exports written on the device: whole instance
ranges in one kernel each, into caller-owned device memory.  Checked field by field against the oracle (layouts:
renderer/src/renderer/buffer/vertex.rs:60-102, renderer/src/renderer/uniform/skin.rs:12-17,41-50) and byte for byte
against the per-instance host exports; the C3-shaped run reports the kernels' HBM rate."""
import ctypes as C
import time

import numpy as np
import pytest

from renderer_amd import abi, api, synth
from tests.helpers import DeviceBuffers, assert_equal

pytestmark = pytest.mark.gpu


def _crowd(spec, instances, **cfg):
    rig = synth.make_rig(spec)
    crowd = synth.make_crowd(spec, rig, instances=instances)
    ctx = api.Context(**cfg)
    r = api.Rig(ctx, rig)
    c = api.Crowd(ctx, r, instances)
    c.set_clocks(crowd.tau, crowd.clip_id)
    if crowd.morph_weights is not None:
        c.set_morph_weights(crowd.morph_weights)
    c.evaluate(sync=True)
    return rig, crowd, ctx, r, c


@pytest.mark.parametrize("verts", [64, 63, 61])   # V * 17 and V * 19 dwords: multiples of 4 and not (instance bases off 16-byte alignment)
def test_device_exports_against_the_oracle(oracle32, verts):
    spec = synth.tiny_spec(seed=2, verts=verts)
    I = 5
    rig, crowd, ctx, r, c = _crowd(spec, I, exact_arithmetic=1)
    ref = oracle32.evaluate(rig, crowd.tau, crowd.clip_id, crowd.morph_weights)
    V, J = rig.vertex_count, rig.joint_count
    dev = DeviceBuffers()
    try:
        first, count = 1, 3
        # [count][128] SkinMatrix
        su_dev = dev.empty(count * 128 * 112)
        c.skin_uniform_device(first, count, su_dev)
        ctx.synchronize()
        su = dev.download(su_dev, (count, 128, 28), np.float32)
        assert_equal(su[:, :J], ref.palette[first:first + count], "SkinMatrix entries")
        assert (su[:, J:] == 0).all()
        for k in range(count):
            assert np.array_equal(su[k].view(np.uint32), c.skin_uniform(first + k).view(np.uint32))
        # TextureSkinVertex, tex_coords from a device array
        uv = np.arange(V * 2, dtype=np.float32).reshape(V, 2) + 0.5
        uv_dev = dev.upload(uv)
        tv_dev = dev.empty(count * V * 68 + 64)
        c.interleaved_device(first, count, abi.LAYOUT_TEXTURE_SKIN, uv_dev, tv_dev)
        ctx.synchronize()
        raw = dev.download(tv_dev, (count * V * 68 + 64,), np.uint8)
        assert (raw[count * V * 68:] == 0xAB).all(), "wrote past the end of the destination"
        tv = raw[:count * V * 68].reshape(count, V, 68)
        f = tv[:, :, :44].copy().view(np.float32).reshape(count, V, 11)
        assert_equal(f[:, :, 0:3], ref.position[first:first + count], "TextureSkinVertex.position")
        assert_equal(f[:, :, 3:5], np.broadcast_to(uv, (count, V, 2)), "TextureSkinVertex.tex_coords")
        assert_equal(f[:, :, 5:8], ref.normal[first:first + count], "TextureSkinVertex.normal")
        assert_equal(f[:, :, 8:11], ref.tangent[first:first + count], "TextureSkinVertex.tangent")
        assert_equal(tv[:, :, 44:52].copy().view(np.uint16).reshape(count, V, 4), np.broadcast_to(rig.joints[0], (count, V, 4)), "joint_index")
        assert_equal(tv[:, :, 52:68].copy().view(np.float32).reshape(count, V, 4), np.broadcast_to(rig.weights[0], (count, V, 4)), "joint_weight")
        for k in range(count):
            assert np.array_equal(tv[k].reshape(-1), c.interleaved(first + k, abi.LAYOUT_TEXTURE_SKIN, uv))
        # ColorSkinVertex, no colour array: zeros
        cv_dev = dev.empty(I * V * 76 + 64)
        c.interleaved_device(0, I, abi.LAYOUT_COLOR_SKIN, 0, cv_dev)
        ctx.synchronize()
        raw = dev.download(cv_dev, (I * V * 76 + 64,), np.uint8)
        assert (raw[I * V * 76:] == 0xAB).all()
        cv = raw[:I * V * 76].reshape(I, V, 76)
        g = cv[:, :, :52].copy().view(np.float32).reshape(I, V, 13)
        assert_equal(g[:, :, 0:3], ref.position, "ColorSkinVertex.position")
        assert (g[:, :, 3:7] == 0).all()
        assert_equal(g[:, :, 7:10], ref.normal, "ColorSkinVertex.normal")
        assert_equal(g[:, :, 10:13], ref.tangent, "ColorSkinVertex.tangent")
        for k in (0, I - 1):
            assert np.array_equal(cv[k].reshape(-1), c.interleaved(k, abi.LAYOUT_COLOR_SKIN, None))
    finally:
        dev.close()
        c.close(); r.close(); ctx.close()


def test_device_export_argument_errors():
    spec = synth.tiny_spec(seed=3)
    rig, crowd, ctx, r, c = _crowd(spec, 2)
    dev = DeviceBuffers()
    try:
        dst = dev.empty(2 * 128 * 112)
        for call in (lambda: c.skin_uniform_device(1, 2, dst), lambda: c.interleaved_device(2, 1, abi.LAYOUT_COLOR_SKIN, 0, dst)):
            with pytest.raises(api.SkinrtError) as e:
                call()
            assert e.value.code == abi.E_RANGE
        with pytest.raises(api.SkinrtError) as e:
            c.interleaved_device(0, 1, 7, 0, dst)
        assert e.value.code == abi.E_INVALID
        with pytest.raises(api.SkinrtError) as e:
            c.skin_uniform_device(0, 1, dst + 4)   # not 16-byte aligned
        assert e.value.code == abi.E_INVALID
        c.skin_uniform_device(0, 0, 0)              # empty range: nothing to do
    finally:
        dev.close()
        c.close(); r.close(); ctx.close()


def test_device_export_c3_shape_rate():
    """C3's avatar x 1,500 instances: TextureSkinVertex records (6.1 GB written, 3.2 GB read) and the SkinMatrix
    palettes of all instances; sampled instances byte-equal to the host export."""
    spec = synth.CONFIGS["c3"]
    I = 1500
    rig, crowd, ctx, r, c = _crowd(spec, I, output_placement=1)
    V, J = rig.vertex_count, rig.joint_count
    assert J > 128   # C3 has 200 joints: the 128-entry uniform cannot hold it, as in the reference (uniform/skin.rs:40)
    dev = DeviceBuffers()
    hip = dev.hip
    try:
        with pytest.raises(api.SkinrtError) as e:
            c.skin_uniform_device(0, 1, dev.empty(128 * 112))
        assert e.value.code == abi.E_LIMIT
        uv_dev = dev.upload(np.zeros((V, 2), np.float32))
        dst = dev.empty(I * V * 68)
        e0, e1 = C.c_void_p(), C.c_void_p()
        assert hip.hipEventCreate(C.byref(e0)) == 0 and hip.hipEventCreate(C.byref(e1)) == 0
        stream = C.c_void_p(ctx.stream)
        c.interleaved_device(0, I, abi.LAYOUT_TEXTURE_SKIN, uv_dev, dst)   # warm-up
        assert hip.hipEventRecord(e0, stream) == 0
        reps = 3
        for _ in range(reps):
            c.interleaved_device(0, I, abi.LAYOUT_TEXTURE_SKIN, uv_dev, dst)
        assert hip.hipEventRecord(e1, stream) == 0
        ctx.synchronize()
        ms = C.c_float()
        assert hip.hipEventElapsedTime(C.byref(ms), e0, e1) == 0
        per = ms.value / reps
        moved = I * V * (36 + 68)
        print(f"\n[export] TextureSkinVertex x {I} instances x {V} vertices: {per:.3f} ms, {moved / per / 1e6:.0f} GB/s (read 36 B + write 68 B per vertex)")
        for k in (0, 777, I - 1):
            got = dev.download(dst + k * V * 68, (V * 68,), np.uint8)
            assert np.array_equal(got, c.interleaved(k, abi.LAYOUT_TEXTURE_SKIN, np.zeros((V, 2), np.float32)))
        hip.hipEventDestroy(e0); hip.hipEventDestroy(e1)
    finally:
        dev.close()
        c.close(); r.close(); ctx.close()


def test_storage_buffer_palette_lifts_the_128_joint_limit(oracle32):
    """skinrt_export_skin_storage_device ("lifts MAX_JOINTS=128 by switching the palette binding to a
    storage buffer"): a 300-joint chain rig — refused by the uniform-array exports, as the reference's assert!
    (uniform/skin.rs:40,60) refuses it — exported as `slots` SkinMatrix records per instance; entries equal the oracle's
    SkinMatrix palette bit for bit, the tail is zero, too few slots is SKINRT_E_LIMIT, and with slots = 128 a small rig
    gives the uniform array's bytes."""
    spec = synth.scaled(synth.CONFIGS["c5"], nodes=300, joints=300, verts=512, targets=0, keys=4)
    I = 4
    rig, crowd, ctx, r, c = _crowd(spec, I, exact_arithmetic=1)
    ref = oracle32.evaluate(rig, crowd.tau, crowd.clip_id, crowd.morph_weights, want=("palette",))
    J = rig.joint_count
    assert J == 300
    dev = DeviceBuffers()
    try:
        with pytest.raises(api.SkinrtError) as e:
            c.skin_uniform_device(0, 1, dev.empty(128 * 112))
        assert e.value.code == abi.E_LIMIT
        for slots in (J, 320):
            dst = dev.empty(3 * slots * 112 + 64)
            c.skin_storage_device(0, 1, 3, slots, dst)
            ctx.synchronize()
            raw = dev.download(dst, (3 * slots * 112 + 64,), np.uint8)
            assert (raw[3 * slots * 112:] == 0xAB).all(), "wrote past the end of the destination"
            su = raw[:3 * slots * 112].copy().view(np.float32).reshape(3, slots, 28)
            assert_equal(su[:, :J].view(np.uint32), np.ascontiguousarray(ref.palette[1:4]).view(np.uint32), f"{slots} slots: SkinMatrix entries")
            assert (su[:, J:] == 0).all()
        with pytest.raises(api.SkinrtError) as e:
            c.skin_storage_device(0, 0, 1, J - 1, dst)
        assert e.value.code == abi.E_LIMIT
    finally:
        dev.close()
        c.close(); r.close(); ctx.close()
    # a small rig: 128 slots = the uniform array
    rig, crowd, ctx, r, c = _crowd(synth.tiny_spec(seed=2), 3)
    dev = DeviceBuffers()
    try:
        a, b = dev.empty(3 * 128 * 112), dev.empty(3 * 128 * 112)
        c.skin_uniform_device(0, 3, a)
        c.skin_storage_device(0, 0, 3, 128, b)
        ctx.synchronize()
        assert np.array_equal(dev.download(a, (3 * 128 * 112,), np.uint8), dev.download(b, (3 * 128 * 112,), np.uint8))
    finally:
        dev.close()
        c.close(); r.close(); ctx.close()
