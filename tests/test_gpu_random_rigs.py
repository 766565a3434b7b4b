"""Differential test over randomly drawn rigs: every case is a new combination of the things the kernels branch on —
tree or chain, depth, joint locality (LDS palette / L2 gather), 1-4 weight sets (k_skin2 takes 1-2, k_skin the rest),
dense / compacted morph targets and how many are active, interpolation mix, Matrix nodes, weights channels, ragged
vertex and instance counts, root transforms, both pose paths, both skin paths — evaluated through the C ABI with the
contraction-free kernels and compared with the fp32 oracle BIT FOR BIT.  Seeds are fixed: a failure names its case."""
import numpy as np
import pytest

from renderer_amd import abi, api, synth
from tests.helpers import assert_close, assert_equal, run_gpu, run_oracle

pytestmark = pytest.mark.gpu

KEYS = ("key_index", "world", "palette", "position", "normal", "tangent", "aabb")


def _draw(case: int):
    rng = np.random.default_rng(0xC0FFEE + case)
    chain = rng.random() < 0.3
    nodes = int(rng.integers(2, 90)) if not chain else int(rng.integers(20, 400))
    joints = int(rng.integers(1, nodes + 1))
    sets = int(rng.choice([1, 1, 2, 2, 3, 4]))
    targets = int(rng.choice([0, 1, 2, 3, 7, 20, 40]))
    clips = int(rng.integers(1, 4))
    spec = synth.SynthSpec(
        f"random{case}", seed=1000 + case, nodes=nodes, joints=joints, verts=int(rng.integers(1, 1400)), sets=sets,
        targets=targets, target_mode=str(rng.choice(["dense", "sparse"])), target_frac=float(rng.choice([0.02, 0.1, 0.6, 1.0])),
        clips=clips, keys=int(rng.integers(2, 9)), topology="chain" if chain else "tree", max_depth=int(rng.integers(1, 12)),
        frac_step=float(rng.choice([0.0, 0.3])), frac_cubic=float(rng.choice([0.0, 0.4])),
        scale_jitter=0.002 if chain else 0.1, matrix_nodes=int(rng.integers(0, 3)) if not chain else 0,
        joint_spread=int(rng.integers(0, 6)), scatter_joints=bool(rng.random() < 0.15),
        morph_zero_frac=float(rng.choice([0.0, 0.5, 0.9])),
        weight_channels=int(rng.integers(0, clips + 1)) if targets else 0)
    cfg = dict(pose_path=int(rng.choice([0, 1, 2])), skin_path=int(rng.choice([0, 0, 1])), exact_arithmetic=1,
               normalize_outputs=int(rng.random() < 0.2), cubic_quirk=int(rng.random() < 0.7))
    instances = int(rng.choice([1, 3, 64, 81, 257]))
    return spec, cfg, instances, bool(rng.random() < 0.5)


@pytest.mark.parametrize("case", range(150))
def test_random_rig_equals_the_oracle(oracle32, case):
    spec, cfg, instances, with_root = _draw(case)
    rig = synth.make_rig(spec)
    crowd = synth.make_crowd(spec, rig, instances=instances, with_root=with_root)
    if case % 5 == 0:
        crowd.clip_id[:: 3] = abi.NO_CLIP            # some instances at rest
    got = run_gpu(rig, crowd, **cfg)
    ref = run_oracle(oracle32, rig, crowd, cubic_quirk=cfg["cubic_quirk"], normalize_outputs=cfg["normalize_outputs"])
    tag = f"case {case} ({spec}, {cfg}, I={instances}): "
    for k in KEYS:
        assert_equal(got[k], ref[k], tag + k)


@pytest.mark.parametrize("case", range(150, 230))
def test_random_rig_fma_kernels_within_the_stated_tolerance(oracle32, case):
    """The same draw with the FMA kernels (exact_arithmetic = 2): integers and the pose stage still equal the oracle,
    skinned outputs within 1e-5 rel / 1e-6 abs.  Chains are left out: far from the origin the bar is not meaningful for
    fused arithmetic (tests/test_gpu_parity.py::test_c5_shape_reduced), which is why the library does not choose it
    there."""
    spec, cfg, instances, with_root = _draw(case)
    if spec.topology == "chain":
        pytest.skip("chain rig: the library itself runs the exact kernels on these")
    cfg = dict(cfg, exact_arithmetic=2)
    rig = synth.make_rig(spec)
    crowd = synth.make_crowd(spec, rig, instances=instances, with_root=with_root)
    got = run_gpu(rig, crowd, **cfg)
    ref = run_oracle(oracle32, rig, crowd, cubic_quirk=cfg["cubic_quirk"], normalize_outputs=cfg["normalize_outputs"])
    tag = f"case {case} ({spec}, {cfg}, I={instances}): "
    for k in ("key_index", "world", "palette"):
        assert_equal(got[k], ref[k], tag + k)
    for k in ("position", "normal", "tangent", "aabb"):
        assert_close(got[k], ref[k], tag + k)


@pytest.mark.parametrize("case", range(150, 230))
def test_random_rig_default_arithmetic_is_inside_the_stated_tolerance(oracle32, case):
    """exact_arithmetic = 0: the library chooses per rig, and whatever it chooses — fused, or for rigs far from the origin
    contraction-free positions with fused normals / tangents (skinrt_rig_info.exact_arithmetic = 2) — must be inside
    1e-5 rel / 1e-6 abs with nothing widened, chains included; where it reports contraction-free positions they and the
    bounds carry the oracle's bits."""
    spec, cfg, instances, with_root = _draw(case)
    cfg = dict(cfg, exact_arithmetic=0)
    rig = synth.make_rig(spec)
    crowd = synth.make_crowd(spec, rig, instances=instances, with_root=with_root)
    got = run_gpu(rig, crowd, **cfg)
    ref = run_oracle(oracle32, rig, crowd, cubic_quirk=cfg["cubic_quirk"], normalize_outputs=cfg["normalize_outputs"])
    chosen = got["info"].exact_arithmetic
    assert chosen in (0, 2)
    tag = f"case {case} ({spec}, {cfg}, I={instances}, chosen {chosen}): "
    for k in ("key_index", "world", "palette"):
        assert_equal(got[k], ref[k], tag + k)
    for k in ("position", "aabb"):
        if chosen == 2:
            assert_equal(got[k], ref[k], tag + k)
        else:
            assert_close(got[k], ref[k], tag + k)
    for k in ("normal", "tangent"):
        assert_close(got[k], ref[k], tag + k)
