"""Tiles whose joint list has 22 .. 64 entries: served by the self-staging skin kernel's LONG instantiations (up to six LDS-DMA
requests per wave and instance) and, for the instances on the work lists, by k_skin_fix with 256- / 384-piece staging buffers —
until round 3 such tiles fell to the loader-wave kernel at 1.5 - 2 x the cost (tools/exp_spread.py).  Bit-equal to the
fp32 oracle in exact arithmetic, inside 1e-5 rel / 1e-6 abs in the default."""
import numpy as np
import pytest

from renderer_amd import synth
from tests.helpers import assert_close, assert_equal, run_gpu, run_oracle

pytestmark = pytest.mark.gpu


def _spec(spread, sets, targets, seed):
    return synth.scaled(synth.tiny_spec(seed=seed), nodes=80, joints=64, verts=1400, sets=sets, targets=targets, target_mode="dense",
                        joint_spread=spread, clips=2, keys=6, max_depth=10)


@pytest.mark.parametrize("spread,sets,targets", [(5, 1, 0), (6, 1, 10), (9, 1, 10), (12, 1, 10), (15, 1, 10), (20, 1, 10), (26, 1, 10), (26, 1, 0), (5, 2, 10), (8, 2, 10), (12, 2, 6), (24, 2, 6)])
def test_long_joint_lists_bit_equal(oracle32, spread, sets, targets):
    spec = _spec(spread, sets, targets, seed=300 + spread + sets)
    rig = synth.make_rig(spec)
    crowd = synth.make_crowd(spec, rig, instances=173)
    if targets:
        crowd.morph_weights[:, : targets // 2] = np.float32(0.5)  # several active pairs per tile: the work lists are used
    got = run_gpu(rig, crowd, exact_arithmetic=1)
    info = got["info"]
    assert 21 < info.max_tile_joints <= 64, f"the rig must exercise the long lists: max tile joints {info.max_tile_joints}"
    assert info.lds_tiles == info.tile_count
    ref = run_oracle(oracle32, rig, crowd)
    for k in ("key_index", "world", "palette", "position", "normal", "tangent", "aabb"):
        assert_equal(got[k], ref[k], f"spread {spread}, {sets} sets, {targets} targets: {k}")


def test_long_joint_lists_default_arithmetic(oracle32):
    spec = _spec(12, 1, 10, seed=341)
    rig = synth.make_rig(spec)
    crowd = synth.make_crowd(spec, rig, instances=300)
    got = run_gpu(rig, crowd)
    assert got["info"].max_tile_joints > 21
    ref = run_oracle(oracle32, rig, crowd)
    for k in ("position", "normal", "tangent"):
        assert_close(got[k], ref[k], k)
    assert_equal(got["aabb"][:, 0], got["position"].min(axis=1), "aabb.min")
    assert_equal(got["aabb"][:, 1], got["position"].max(axis=1), "aabb.max")


def test_mixed_short_and_long_tiles_in_one_rig(oracle32):
    """most tiles short, some long (a second vertex population bound loosely): one ring size serves both"""
    spec = _spec(3, 1, 8, seed=355)
    rig = synth.make_rig(spec)
    rng = np.random.default_rng(9)
    V = rig.vertex_count
    loose = slice(V // 2, V // 2 + 300)  # these vertices draw their four joints from a window of 30
    rig.joints[0][loose] = rng.integers(10, 40, size=(300, 4)).astype(np.uint16)
    crowd = synth.make_crowd(spec, rig, instances=97)
    got = run_gpu(rig, crowd, exact_arithmetic=1)
    assert 21 < got["info"].max_tile_joints <= 42
    ref = run_oracle(oracle32, rig, crowd)
    for k in ("position", "normal", "tangent", "aabb"):
        assert_equal(got[k], ref[k], k)


@pytest.mark.parametrize("targets,active", [(40, 1), (48, 2), (64, 2), (48, 5), (70, 2), (128, 2), (140, 2)])
def test_tiles_with_more_than_32_morph_pairs(oracle32, targets, active):
    """dense targets over the whole (small) mesh: every tile carries `targets` pairs.  Instances with one or two active targets are
    morphed by the self-staging kernel itself up to 128 pairs per tile (pair-encoded masks), the rest go through the work lists;
    either way the oracle's bits."""
    spec = synth.scaled(synth.tiny_spec(seed=500 + targets), nodes=30, joints=24, verts=1100, targets=targets, target_mode="dense",
                        target_frac=1.0, clips=1, keys=4)
    rig = synth.make_rig(spec)
    I = 211
    crowd = synth.make_crowd(spec, rig, instances=I)
    rng = np.random.default_rng(targets * 7 + active)
    w = np.zeros((I, targets), np.float32)
    for i in range(I):
        k = active if i % 5 else (0 if i % 10 else active + 2)  # mostly `active` targets, some idle instances, some with more
        if k:
            w[i, rng.choice(targets, size=k, replace=False)] = rng.uniform(0.1, 1.0, size=k).astype(np.float32)
    crowd.morph_weights[:] = w
    got = run_gpu(rig, crowd, exact_arithmetic=1)
    ref = run_oracle(oracle32, rig, crowd)
    for k in ("position", "normal", "tangent", "aabb"):
        assert_equal(got[k], ref[k], f"{targets} targets, {active} active: {k}")
    # a second frame with other weights on the same crowd shape (runs, spacers and lists change)
    crowd.morph_weights[:] = np.roll(w, 3, axis=0)
    got = run_gpu(rig, crowd, exact_arithmetic=1)
    ref = run_oracle(oracle32, rig, crowd)
    for k in ("position", "normal", "tangent"):
        assert_equal(got[k], ref[k], f"{targets} targets, {active} active, rolled: {k}")


@pytest.mark.parametrize("sets,targets", [(1, 0), (1, 9), (2, 6)])
def test_few_tiles_over_42_joints_are_handed_over_whole(oracle32, sets, targets):
    """one tile of a 60-tile mesh bound to a window of 60 bones (43 .. 64 joints: a ring sized for it would leave two blocks per CU): at most 2 % of
    the tiles, so it goes to k_skin_fix whole (384-piece staging) and the ring stays sized for the rest.  The oracle's bits."""
    spec = synth.scaled(synth.tiny_spec(seed=611 + sets + targets), nodes=90, joints=80, verts=60 * 256, sets=sets, targets=targets,
                        target_mode="dense", joint_spread=2, clips=2, keys=5, max_depth=10)
    rig = synth.make_rig(spec)
    rng = np.random.default_rng(23)
    v0 = 31 * 256
    for k in range(sets):
        rig.joints[k][v0:v0 + 256] = rng.integers(5, 65, size=(256, 4)).astype(np.uint16)
    crowd = synth.make_crowd(spec, rig, instances=41)
    got = run_gpu(rig, crowd, exact_arithmetic=1)
    info = got["info"]
    assert 42 < info.max_tile_joints <= 64, info.max_tile_joints
    assert info.handed_over_tiles >= 1
    ref = run_oracle(oracle32, rig, crowd)
    for k in ("position", "normal", "tangent", "aabb"):
        assert_equal(got[k], ref[k], f"{sets} sets, {targets} targets: {k}")
