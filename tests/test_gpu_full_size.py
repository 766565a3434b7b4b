"""BASELINE.json's configurations at FULL size on the GPU (C2 Fox x 4096, C3 avatar x 10,000,
C5 stress rig 1M vertices x 1000; C1 and C4's rig are covered in test_gpu_parity / by C3).
The oracle cannot evaluate a whole crowd in seconds, so parity is checked where it can:
  * a spread of sampled instances (first, last, batch/range boundaries, random) is evaluated by
    the fp32 oracle alone (a one-instance crowd) and compared with the same instances fetched from the
    full run — keyframe indices, world matrices and palettes bit-exact, skinned outputs within
    1e-5 rel / 1e-6 abs;
  * size-independent properties over ALL instances: bounds == min/max of the kernel's own
    positions on the samples, every bound finite and ordered, evaluation is idempotent (a second
    frame with the same inputs rewrites identical bytes for the sampled instances)."""
import numpy as np
import pytest

from renderer_amd import abi, api, synth
from tests.helpers import assert_close, assert_equal

pytestmark = pytest.mark.gpu


def run_full(oracle32, name, samples, conditioned=False, oracle64=None, exact=None, first=0, count=None, **cfg):
    """exact: None = what the library chooses for the rig (exact_arithmetic=0), True = forced exact
    (bit-equal skinned outputs expected), False = forced FMA.  first/count: a shard of the crowd."""
    spec = synth.CONFIGS[name]
    rig = synth.make_rig(spec)
    n = spec.instances if count is None else count
    crowd = synth.make_crowd(spec, rig, instances=n, first=first)
    if exact is not None:
        cfg["exact_arithmetic"] = 1 if exact else 2
    with api.Context(**cfg) as ctx:
        r = api.Rig(ctx, rig)
        c = api.Crowd(ctx, r, n)
        c.set_clocks(crowd.tau, crowd.clip_id)
        if crowd.morph_weights is not None:
            c.set_morph_weights(crowd.morph_weights)
        c.evaluate(abi.STAGE_ALL, sync=True)
        first_pass = {i: c.positions(i, 1).copy() for i in samples[:3]}
        c.evaluate(abi.STAGE_ALL, sync=True)
        for i, p in first_pass.items():
            assert_equal(c.positions(i, 1).view(np.uint32), p.view(np.uint32), f"{name}: idempotence, instance {i}")
        bounds = c.aabbs()
        assert np.isfinite(bounds).all(), f"{name}: non-finite bounds"
        assert (bounds[:, 0] <= bounds[:, 1]).all(), f"{name}: min > max"
        desc = rig.to_desc()
        for i in samples:
            one = (crowd.tau[i:i + 1], crowd.clip_id[i:i + 1],
                   None if crowd.morph_weights is None else crowd.morph_weights[i:i + 1])
            ref = oracle32.evaluate(rig, *one, desc=desc)
            tag = f"{name}[{i}]:"
            assert_equal(c.keyframe_index(i, 1)[0], ref.key_index[0], tag + "key_index")
            assert_equal(c.world(i, 1)[0], ref.world_rows()[0], tag + "world")
            assert_equal(c.palette(i, 1)[0], ref.palette_rows()[0], tag + "palette")
            pos = c.positions(i, 1)[0]
            extra = 0.0
            if conditioned:  # see test_gpu_parity.test_c5_shape_reduced: widen by the oracle's own fp32-vs-fp64 gap
                r64 = oracle64.evaluate(rig, *one, desc=desc, want=("position", "normal", "tangent"))
                extra = 4.0 * max(float(np.abs(getattr(ref, k)[0].astype(np.float64) - getattr(r64, k)[0]).max())
                                  for k in ("position", "normal", "tangent"))
            if c.exact:  # contraction-free arithmetic (chosen by the library or forced): the oracle's bits
                assert_equal(pos, ref.position[0], tag + "position")
                assert_equal(c.normals(i, 1)[0], ref.normal[0], tag + "normal")
                assert_equal(c.tangents(i, 1)[0], ref.tangent[0], tag + "tangent")
            elif c.exact_positions:  # chosen by the library for a far-from-origin rig: positions exact, normals / tangents fused
                assert not conditioned
                assert_equal(pos, ref.position[0], tag + "position")
                assert_close(c.normals(i, 1)[0], ref.normal[0], tag + "normal")
                assert_close(c.tangents(i, 1)[0], ref.tangent[0], tag + "tangent")
            else:
                assert_close(pos, ref.position[0], tag + "position", atol=1e-6 + extra)
                assert_close(c.normals(i, 1)[0], ref.normal[0], tag + "normal", atol=1e-6 + extra)
                assert_close(c.tangents(i, 1)[0], ref.tangent[0], tag + "tangent", atol=1e-6 + extra)
            assert_equal(bounds[i, 0], pos.min(axis=0), tag + "aabb.min")
            assert_equal(bounds[i, 1], pos.max(axis=0), tag + "aabb.max")
        c.close(); r.close()


def spread(n, seed, k=4):
    rng = np.random.default_rng(seed)
    fixed = [0, 1, 7, 8, 255, 256, n - 1]
    return sorted(set(int(x) for x in fixed if 0 <= x < n) | set(int(x) for x in rng.integers(0, n, size=k)))


def test_c2_fox_x4096(oracle32):
    run_full(oracle32, "c2", spread(4096, 2, k=8))


def test_c3_avatar_x10000(oracle32):
    run_full(oracle32, "c3", spread(10000, 3))


def test_c5_stress_x1000(oracle32):
    """C5 as the library runs it by default: the rig's joints sit hundreds of units from the origin, where
    fused and unfused multiply-adds differ by more than the 1e-6 absolute bar in the POSITION sums (they cancel),
    so skinrt computes positions contraction-free for it (skinrt_config.exact_arithmetic = 0 -> rig_info 2): they carry
    the fp32 oracle's bits; normals and tangents (3x3 sums, nothing cancels) stay fused and must sit inside
    1e-5 rel / 1e-6 abs with nothing widened."""
    run_full(oracle32, "c5", [0, 255, 999])


def test_c5_stress_x1000_exact(oracle32):
    run_full(oracle32, "c5", [1, 500, 998], exact=True)


def test_c5_stress_x1000_forced_fma(oracle32, oracle64):
    """exact_arithmetic = 2 forces FMA contraction on the same rig: closer to the fp64 oracle than the fp32
    oracle is, but not inside the bar measured against the fp32 oracle — the bar is widened here by the
    oracle's own fp32-vs-fp64 gap (see test_gpu_parity.test_c5_shape_reduced).  Kept as the measurement
    of why the default for this rig is the exact mode."""
    run_full(oracle32, "c5", [0, 999], conditioned=True, oracle64=oracle64, exact=False)


@pytest.mark.parametrize("ranks,rank", [(8, 3), (2, 1)])
def test_c4_shard_on_one_gpu(oracle32, ranks, rank):
    """BASELINE config 4 (100,000 avatars over 2 / 4 / 8 GPUs): the shard one rank evaluates, on this GPU —
    12,500 instances (27 GB of outputs) for 8 ranks, 50,000 (108 GB) for 2.  Instance ids are the crowd's
    (first = rank * shard), so sampled instances are the ones rank `rank` would hand to the all-gather."""
    from renderer_amd import shard
    lo, hi = shard.shard_range(synth.CONFIGS["c4"].instances, rank, ranks)
    n = hi - lo
    run_full(oracle32, "c4", [0, 79, 80, n // 2, n - 1], first=lo, count=n)


def test_instance_range_cap_with_batch_not_dividing_256(oracle32):
    """70,000 instances x 16 tiles: the per-block instance range reaches its cap of 256, and this
    rig's batch size (10: 16 joints per tile, no morph targets) does not divide 256 — the range
    must round DOWN (250), or blocks would index past the kernel's 256-entry instance-order
    tables (a fault seen once with a diagnostic build).  Sampled instances against the oracle,
    including the first / last of ranges under the wrong (260) and the right (250) cut."""
    spec = synth.scaled(synth.tiny_spec(seed=31), nodes=40, joints=32, verts=4096, targets=0, clips=1, keys=4,
                        joint_spread=6, max_depth=8)
    rig = synth.make_rig(spec)
    n = 70000
    crowd = synth.make_crowd(spec, rig, instances=n)
    samples = [0, 249, 250, 255, 256, 259, 260, 499, 500, 34999, 69749, 69750, n - 1]
    with api.Context() as ctx:
        r = api.Rig(ctx, rig)
        c = api.Crowd(ctx, r, n)
        c.set_clocks(crowd.tau, crowd.clip_id)
        c.evaluate(abi.STAGE_ALL, sync=True)
        bounds = c.aabbs()
        assert np.isfinite(bounds).all()
        desc = rig.to_desc()
        for i in samples:
            ref = oracle32.evaluate(rig, crowd.tau[i:i + 1], crowd.clip_id[i:i + 1], desc=desc)
            assert_equal(c.palette(i, 1)[0], ref.palette_rows()[0], f"[{i}] palette")
            pos = c.positions(i, 1)[0]
            assert_close(pos, ref.position[0], f"[{i}] position")
            assert_close(c.normals(i, 1)[0], ref.normal[0], f"[{i}] normal")
            assert_equal(bounds[i, 0], pos.min(axis=0), f"[{i}] aabb.min")
            assert_equal(bounds[i, 1], pos.max(axis=0), f"[{i}] aabb.max")
        c.close(); r.close()


def test_c3_schedule_independence_under_instance_permutation():
    """The skin kernels choose, per block of instances, an order (sorted by active morph pairs, idle
    instances as spacers) and what to hand over to k_skin_fix: none of that may show in the result.
    The full C3 crowd is evaluated twice, the second time with its instances shuffled — every
    block then sees another set of instances, other runs and other hand-overs.  Instance j of
    the shuffled run carries the inputs of instance perm[j]: bounds must be bit-equal for ALL 10,000
    instances, positions / normals / tangents for a sample."""
    spec = synth.CONFIGS["c3"]
    rig = synth.make_rig(spec)
    n = spec.instances
    crowd = synth.make_crowd(spec, rig, instances=n)
    rng = np.random.default_rng(77)
    perm = rng.permutation(n)
    sample = [0, 1, 79, 80, 81, n - 1] + [int(x) for x in rng.integers(0, n, size=10)]
    with api.Context() as ctx:
        r = api.Rig(ctx, rig)
        c = api.Crowd(ctx, r, n)
        c.set_clocks(crowd.tau, crowd.clip_id)
        c.set_morph_weights(crowd.morph_weights)
        c.evaluate(abi.STAGE_ALL, sync=True)
        bounds1 = c.aabbs().copy()
        first = {j: (c.positions(int(perm[j]), 1).copy(), c.normals(int(perm[j]), 1).copy(), c.tangents(int(perm[j]), 1).copy())
                 for j in sample}
        c.set_clocks(crowd.tau[perm], crowd.clip_id[perm])
        c.set_morph_weights(crowd.morph_weights[perm])
        c.evaluate(abi.STAGE_ALL, sync=True)
        bounds2 = c.aabbs()
        assert_equal(bounds2.view(np.uint32), bounds1[perm].view(np.uint32), "bounds of all instances")
        for j in sample:
            p, nr, tg = first[j]
            assert_equal(c.positions(j, 1).view(np.uint32), p.view(np.uint32), f"position, shuffled instance {j}")
            assert_equal(c.normals(j, 1).view(np.uint32), nr.view(np.uint32), f"normal, shuffled instance {j}")
            assert_equal(c.tangents(j, 1).view(np.uint32), tg.view(np.uint32), f"tangent, shuffled instance {j}")
        c.close(); r.close()


# ---- the WHOLE crowd against the oracle (VERDICT r2 "weak 2") ---------------------------------------------------------
# Every instance's bounds (min / max over ALL its vertices: every (tile, instance block, work list) combination takes
# part) and every 257th vertex of every instance (the phase walks with the instance: a block of 80 instances meets
# every 256-vertex tile about 80 times) are computed by the fp32 oracle on the host's cores and compared with the
# GPU's: bit for bit with exact_arithmetic = 1, at 1e-5 rel / 1e-6 abs in the default arithmetic.
import functools
import time

VERTEX_STEP = 257


@functools.lru_cache(maxsize=2)
def _whole_crowd_reference(name, first, count):
    from oracle.oracle import Oracle
    o = Oracle("f32")
    spec = synth.CONFIGS[name]
    rig = synth.make_rig(spec)
    n = spec.instances if count is None else count
    crowd = synth.make_crowd(spec, rig, instances=n, first=first)
    t0 = time.perf_counter()
    bounds = o.evaluate_mt(rig, crowd.tau, crowd.clip_id, crowd.morph_weights, want=("aabb",)).aabb
    t1 = time.perf_counter()
    part = o.evaluate_mt(rig, crowd.tau, crowd.clip_id, crowd.morph_weights, want=("position", "normal", "tangent"),
                         vertex_step=VERTEX_STEP, vertex_phase=0)
    print(f"[oracle] {name}: bounds of {n} instances x {rig.vertex_count} vertices in {t1 - t0:.1f} s, "
          f"every {VERTEX_STEP}th vertex in {time.perf_counter() - t1:.1f} s")
    return rig, crowd, bounds, part


def whole_crowd(name, exact, first=0, count=None):
    rig, crowd, ref_bounds, part = _whole_crowd_reference(name, first, count)
    n, V = crowd.tau.shape[0], rig.vertex_count
    with api.Context(exact_arithmetic=1 if exact else 0) as ctx:
        r = api.Rig(ctx, rig)
        c = api.Crowd(ctx, r, n)
        c.set_clocks(crowd.tau, crowd.clip_id)
        if crowd.morph_weights is not None:
            c.set_morph_weights(crowd.morph_weights)
        c.evaluate(abi.STAGE_ALL, sync=True)
        mode = r.info.exact_arithmetic   # 1: all outputs carry the oracle's bits, 2: positions do, 0: fused
        bounds = c.aabbs()
        if mode:
            assert_equal(bounds, ref_bounds, f"{name}: bounds of all {n} instances")
        else:
            assert_close(bounds, ref_bounds, f"{name}: bounds of all {n} instances")
        rows = part.position.shape[1]
        chunk = max(1, (256 << 20) // (V * 12))
        q = np.arange(rows, dtype=np.int64)[None, :] * VERTEX_STEP
        t0 = time.perf_counter()
        for a in range(0, n, chunk):
            b = min(n, a + chunk)
            idx = (np.arange(a, b, dtype=np.int64) % VERTEX_STEP)[:, None] + q
            valid = idx < V
            idx = np.minimum(idx, V - 1)[:, :, None]
            for k, fetch in (("position", c.positions), ("normal", c.normals), ("tangent", c.tangents)):
                got = np.take_along_axis(fetch(a, b - a), idx, axis=1)[valid]
                want = getattr(part, k)[a:b][valid]
                if mode == 1 or (mode == 2 and k == "position"):
                    assert_equal(got, want, f"{name}: {k}, instances {a}..{b}")
                else:
                    assert_close(got, want, f"{name}: {k}, instances {a}..{b}")
        print(f"[gpu] {name}: compared {n} x {rows} sampled vertices x 3 attributes in {time.perf_counter() - t0:.1f} s (arithmetic mode {mode})")
        c.close(); r.close()


@pytest.mark.parametrize("exact", [True, False])
def test_c2_whole_crowd_against_the_oracle(exact):
    whole_crowd("c2", exact)


@pytest.mark.parametrize("exact", [True, False])
def test_c3_whole_crowd_against_the_oracle(exact):
    whole_crowd("c3", exact)


@pytest.mark.parametrize("exact", [True, False])
def test_c5_whole_crowd_against_the_oracle(exact):
    whole_crowd("c5", exact)


def test_c4_shard_of_8_whole_crowd_against_the_oracle():
    from renderer_amd import shard
    lo, hi = shard.shard_range(synth.CONFIGS["c4"].instances, 3, 8)
    whole_crowd("c4", True, first=lo, count=hi - lo)
