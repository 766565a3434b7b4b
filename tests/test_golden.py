"""Golden fixtures (tests/golden/*.npz, written by tests/golden/generate.py from the fp32 oracle;
PARITY UNPINNED — see that script's header).

CPU: the oracle reproduces every fixture bit for bit (catches drift of oracle / generator /
rig file format), and the fp64 oracle stays within the float tolerance of them.
GPU: the HIP library, through the C ABI, matches the fixtures: integers bit-exact, floats within
1e-5 rel / 1e-6 abs (exact-arithmetic mode: bit-exact)."""
import glob
import os

import numpy as np
import pytest

from renderer_amd import rigfile, synth
from tests import helpers

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
NAMES = [os.path.splitext(os.path.basename(p))[0] for p in GOLDEN]
FLOAT_KEYS = ("world", "palette", "position", "normal", "tangent", "aabb")


def load(path):
    with np.load(path, allow_pickle=False) as z:
        d = {k: z[k] for k in z.files}
    rig = rigfile.rig_from_arrays(d)
    crowd = synth.CrowdState(d["tau"], d["clip_id"], d.get("morph_weights"), d.get("root"))
    return rig, crowd, d


def test_fixture_set_is_complete():
    assert len(GOLDEN) >= 11
    for need in ("tiny_all_features", "tiny_three_skins_two_props", "c1_shape", "c2_shape", "c3_shape", "c5_shape"):
        assert need in NAMES


@pytest.mark.parametrize("path", GOLDEN, ids=NAMES)
def test_oracle_reproduces_golden(path, oracle32):
    rig, crowd, d = load(path)
    quirk, norm = (int(x) for x in d["cfg"])
    out = helpers.run_oracle(oracle32, rig, crowd, cubic_quirk=quirk, normalize_outputs=norm)
    helpers.assert_equal(out["key_index"], d["key_index"], "key_index")
    for k in FLOAT_KEYS:
        helpers.assert_equal(out[k].view(np.uint32), d[k].view(np.uint32), k)


@pytest.mark.parametrize("path", GOLDEN, ids=NAMES)
def test_fp64_oracle_within_tolerance_of_golden(path, oracle64):
    rig, crowd, d = load(path)
    quirk, norm = (int(x) for x in d["cfg"])
    out = helpers.run_oracle(oracle64, rig, crowd, cubic_quirk=quirk, normalize_outputs=norm)
    helpers.assert_equal(out["key_index"], d["key_index"], "key_index")
    # long chains accumulate rounding: the c5-shaped chain gets the budget of its depth
    scale = 20.0 if "c5" in path else 1.0
    for k in FLOAT_KEYS:
        helpers.assert_close(d[k], out[k], k, rtol=helpers.RTOL * scale, atol=helpers.ATOL * scale)


@pytest.mark.parametrize("path", GOLDEN, ids=NAMES)
def test_rig_file_round_trip(path, tmp_path):
    rig, _, d = load(path)
    p = str(tmp_path / "again.npz")
    rigfile.save_rig(p, rig)
    again = rigfile.rig_to_arrays(rigfile.load_rig(p))
    first = rigfile.rig_to_arrays(rig)
    assert sorted(again) == sorted(first)
    for k in first:  # bytes, not values: the un-skinned sections of the multi-skin fixture carry NaN weights on purpose (never read)
        assert first[k].shape == again[k].shape == d[k].shape and first[k].dtype == again[k].dtype == d[k].dtype, k
        assert first[k].tobytes() == again[k].tobytes(), k
        assert first[k].tobytes() == d[k].tobytes(), k


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [2, 1])   # 2: FMA kernels, 1: contraction-free kernels
@pytest.mark.parametrize("path", GOLDEN, ids=NAMES)
def test_library_matches_golden(path, exact):
    rig, crowd, d = load(path)
    quirk, norm = (int(x) for x in d["cfg"])
    out = helpers.run_gpu(rig, crowd, cubic_quirk=quirk, normalize_outputs=norm, exact_arithmetic=exact)
    helpers.assert_equal(out["key_index"], d["key_index"], "key_index")
    # pose stage is contraction-free in both modes: bit-equal
    for k in ("world", "palette"):
        helpers.assert_equal(out[k].view(np.uint32), d[k].view(np.uint32), k)
    for k in ("position", "normal", "tangent", "aabb"):
        if exact == 1:
            helpers.assert_equal(out[k].view(np.uint32), d[k].view(np.uint32), k)
        else:
            helpers.assert_close(out[k], d[k], k)
