"""C-ABI boundary and host logic, CPU only: the shared library loads, exports every symbol the
header declares, the header is plain C and its struct layouts agree with the ctypes mirror, and
the host-side validation (`skinrt_validate_rig`, no device calls) reports the documented error
codes.  No compute entry point is called here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from renderer_amd import abi, api, synth
from renderer_amd.abi import Channel, MorphTarget

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "skinrt.h")


def header_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(skinrt_[a-z_0-9]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = api.load_library()
    declared = header_functions()
    assert len(declared) >= 27
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/skinrt.h but not exported"
    assert sorted(api.EXPORTS) == declared


def test_no_unexpected_exports():
    out = subprocess.run(["nm", "-D", "--defined-only", api.lib_path()], capture_output=True, text=True, check=True).stdout
    names = [l.split()[-1] for l in out.splitlines() if " T " in l]
    extra = [n for n in names if not n.startswith("skinrt_")]
    assert not extra, f"library exports non-API symbols: {extra[:5]}"


def test_abi_version_matches_header():
    m = re.search(r"#define\s+SKINRT_ABI_VERSION\s+(\d+)", open(HEADER).read())
    assert api.load_library().skinrt_abi_version() == int(m.group(1))


def test_header_is_plain_c_and_layouts_match_ctypes(tmp_path):
    src = tmp_path / "layout.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "skinrt.h"
#define S(T) printf(#T " %zu\n", sizeof(T))
#define O(T, f) printf(#T "." #f " %zu\n", offsetof(T, f))
int main(void) {
    S(skinrt_config); O(skinrt_config, device); O(skinrt_config, exact_arithmetic); O(skinrt_config, pose_path);
    S(skinrt_channel_desc); O(skinrt_channel_desc, times); O(skinrt_channel_desc, values);
    S(skinrt_clip_desc); O(skinrt_clip_desc, channels);
    S(skinrt_morph_target_desc); O(skinrt_morph_target_desc, tangent);
    S(skinrt_rig_desc); O(skinrt_rig_desc, joint_node); O(skinrt_rig_desc, clips); O(skinrt_rig_desc, position);
    O(skinrt_rig_desc, tangent_stride); O(skinrt_rig_desc, joints); O(skinrt_rig_desc, targets); O(skinrt_rig_desc, default_weights);
    S(skinrt_rig_info); O(skinrt_rig_info, morph_entries); O(skinrt_rig_info, device_bytes);
    return 0;
}
''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    got = dict(l.rsplit(" ", 1) for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    mirror = {"skinrt_config": abi.Config, "skinrt_channel_desc": abi.ChannelDesc, "skinrt_clip_desc": abi.ClipDesc,
              "skinrt_morph_target_desc": abi.MorphTargetDesc, "skinrt_rig_desc": abi.RigDesc, "skinrt_rig_info": abi.RigInfo}
    for key, val in got.items():
        if "." in key:
            t, f = key.split(".")
            assert getattr(mirror[t], f).offset == int(val), key
        else:
            assert C.sizeof(mirror[key]) == int(val), key


def test_error_codes_match_header():
    text = open(HEADER).read()
    for name in ("OK", "E_INVALID", "E_DEVICE", "E_NOMEM", "E_NONAFFINE", "E_RANGE", "E_LIMIT"):
        m = re.search(rf"#define\s+SKINRT_{name}\s+\(?(-?\d+)\)?", text)
        assert m, name
        assert getattr(abi, name) == int(m.group(1)), name


# ---- host-side validation (skinrt_validate_rig: pure host, runs without a GPU) ----------------------------------

def tiny():
    spec = synth.tiny_spec(seed=3)
    return spec, synth.make_rig(spec)


def expect(rig, code, needle, **cfg):
    with pytest.raises(api.SkinrtError) as e:
        api.validate_rig(rig, api.make_config(**cfg) if cfg else None)
    assert e.value.code == code, e.value
    assert needle in e.value.message, e.value.message


def test_validate_reports_rig_info():
    spec, rig = tiny()
    info = api.validate_rig(rig)
    assert (info.node_count, info.joint_count, info.vertex_count) == (spec.nodes, spec.joints, spec.verts)
    assert info.set_count == 1 and info.target_count == spec.targets and info.clip_count == spec.clips
    assert info.tile_vertices == 256 and info.tile_count == 1
    assert info.max_channels == rig.max_channels
    assert 1 <= info.level_count <= spec.nodes


def test_validate_tile_and_level_counts():
    spec = synth.scaled(synth.CONFIGS["c1"])
    info = api.validate_rig(synth.make_rig(spec))
    assert info.tile_count == (3273 + 255) // 256
    assert info.level_count <= spec.max_depth + 1


def test_parent_order_rejected():
    _, rig = tiny()
    rig.parent = rig.parent.copy(); rig.parent[2] = 5   # child before its parent: not a pre-order walk
    expect(rig, abi.E_INVALID, "DFS pre-order")


def test_joint_index_out_of_range():
    _, rig = tiny()
    rig.joints[0] = rig.joints[0].copy(); rig.joints[0][3, 1] = rig.joint_count
    rig.weights[0] = rig.weights[0].copy(); rig.weights[0][3, 1] = 0.5
    expect(rig, abi.E_RANGE, "joint index out of range")


def test_joint_node_out_of_range():
    _, rig = tiny()
    rig.joint_node = rig.joint_node.copy(); rig.joint_node[0] = rig.node_count
    expect(rig, abi.E_RANGE, "joint_node out of range")


def test_non_affine_inverse_bind():
    _, rig = tiny()
    rig.inverse_bind = rig.inverse_bind.copy(); rig.inverse_bind[1, 3] = 0.25   # column 0, row 3
    expect(rig, abi.E_NONAFFINE, "inverse bind matrix last row")


def test_unsorted_key_times():
    _, rig = tiny()
    ch = rig.clips[0][0]
    ch.times = ch.times[::-1].copy()
    expect(rig, abi.E_INVALID, "not sorted")


def test_duplicate_channel_target():
    _, rig = tiny()
    a = rig.clips[0][0]
    rig.clips[0].append(Channel(a.target_node, a.path, a.interpolation, a.times.copy(), a.values.copy()))
    expect(rig, abi.E_INVALID, "same node property")


def test_channel_path_out_of_range_rejected():
    _, rig = tiny()
    rig.clips[0][0].path = 4
    expect(rig, abi.E_INVALID, "channel path invalid")


def test_weights_channel_validation():
    """SKINRT_PATH_WEIGHTS (the reference: todo!(), loader/gltf/mod.rs:1312): width = target_count, one per clip,
    only on rigs with morph targets, finite keys."""
    spec = synth.tiny_spec(seed=5, weight_channels=1)
    rig = synth.make_rig(spec)
    info = api.validate_rig(rig)
    assert info.weight_channel_clips == 1
    wch = [ch for ch in rig.clips[0] if ch.path == abi.PATH_WEIGHTS]
    assert len(wch) == 1 and wch[0].values.shape[-1] == rig.target_count
    # a second one in the same clip
    rig.clips[0].append(abi.Channel(0, abi.PATH_WEIGHTS, abi.INTERP_STEP, wch[0].times.copy(), wch[0].values.reshape(len(wch[0].times), -1)[:, :rig.target_count].copy()))
    expect(rig, abi.E_INVALID, "two weights channels")
    # non-finite key
    rig = synth.make_rig(spec)
    w = [ch for ch in rig.clips[0] if ch.path == abi.PATH_WEIGHTS][0]
    w.values = w.values.copy(); w.values.reshape(-1)[0] = np.inf
    expect(rig, abi.E_INVALID, "non-finite morph weight key")
    # no morph targets at all
    rig = synth.make_rig(synth.tiny_spec(seed=5, targets=0))
    rig.clips[0].append(abi.Channel(0, abi.PATH_WEIGHTS, abi.INTERP_STEP, np.zeros(1, np.float32), np.zeros((1, 1), np.float32)))
    expect(rig, abi.E_INVALID, "without morph targets")


def test_channel_target_out_of_range():
    _, rig = tiny()
    rig.clips[0][0].target_node = 1000
    expect(rig, abi.E_RANGE, "target_node out of range")


def test_non_finite_inputs_rejected():
    _, rig = tiny()
    rig.translation = rig.translation.copy(); rig.translation[0, 0] = np.nan
    expect(rig, abi.E_INVALID, "non-finite node transform")
    _, rig = tiny()
    rig.weights[0] = rig.weights[0].copy(); rig.weights[0][0, 0] = np.inf
    expect(rig, abi.E_INVALID, "non-finite joint weight")


def test_dense_target_length_mismatch():
    _, rig = tiny()
    desc = rig.to_desc()
    desc.targets[0].count = rig.vertex_count - 1   # dense target (indices NULL) must cover every vertex
    lib = api.load_library()
    assert lib.skinrt_validate_rig(None, C.byref(desc), None) == abi.E_INVALID
    assert b"dense morph target count" in lib.skinrt_last_error()


def test_sparse_target_indices_checked():
    _, rig = tiny()
    rig.targets[0] = MorphTarget(position=np.zeros((2, 3), np.float32), indices=np.array([5, 5], np.uint32))
    expect(rig, abi.E_INVALID, "strictly increasing")
    rig.targets[0] = MorphTarget(position=np.zeros((1, 3), np.float32), indices=np.array([rig.vertex_count], np.uint32))
    expect(rig, abi.E_RANGE, "vertex index out of range")


def test_reference_limits_mode():
    # uniform/skin.rs:10 MAX_JOINTS = 128 and the shader's single weight set are opt-in limits
    spec = synth.tiny_spec(seed=4, nodes=140, joints=130, max_depth=10)
    rig = synth.make_rig(spec)
    api.validate_rig(rig)
    expect(rig, abi.E_LIMIT, "128 joints", reference_limits=1)
    spec = synth.tiny_spec(seed=4, sets=2)
    expect(synth.make_rig(spec), abi.E_LIMIT, "set 0 only", reference_limits=1)


def test_struct_size_guard():
    _, rig = tiny()
    desc = rig.to_desc()
    desc.struct_size = 8
    rc = api.load_library().skinrt_validate_rig(None, C.byref(desc), None)
    assert rc == abi.E_INVALID


def test_null_handles_are_errors_not_crashes():
    lib = api.load_library()
    assert lib.skinrt_evaluate(None, abi.STAGE_ALL) == abi.E_INVALID
    assert lib.skinrt_set_clocks(None, None, None) == abi.E_INVALID
    assert lib.skinrt_get_rig_info(None, None) == abi.E_INVALID
    assert lib.skinrt_destroy_crowd(None) in (abi.OK, abi.E_INVALID)
    assert b"NULL" in lib.skinrt_last_error() or lib.skinrt_last_error() != b""


def test_init_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.SkinrtError) as e:
        api.Context()
    assert e.value.code == abi.E_DEVICE


def test_node_entry_points_reject_bad_arguments_without_a_device():
    """skinrt_node_*: argument errors come back as codes before any device is touched."""
    lib = api.load_library()
    rig = synth.make_rig(synth.tiny_spec())
    desc = rig.to_desc()
    cfg = api.make_config()
    node = C.c_void_p()
    devs = (C.c_int32 * 2)(0, 1)
    assert lib.skinrt_node_create(None, devs, 2, C.byref(desc), 4, C.byref(node)) == abi.E_INVALID
    assert lib.skinrt_node_create(C.byref(cfg), None, 2, C.byref(desc), 4, C.byref(node)) == abi.E_INVALID
    assert lib.skinrt_node_create(C.byref(cfg), devs, 0, C.byref(desc), 4, C.byref(node)) == abi.E_INVALID
    assert lib.skinrt_node_create(C.byref(cfg), devs, 65, C.byref(desc), 4, C.byref(node)) == abi.E_INVALID
    assert lib.skinrt_node_create(C.byref(cfg), devs, 2, C.byref(desc), 4, None) == abi.E_INVALID
    for fn in ("skinrt_node_set_clocks", "skinrt_node_set_morph_weights", "skinrt_node_set_root_transforms"):
        assert getattr(lib, fn)(None, None, None) == abi.E_INVALID if fn.endswith("clocks") else getattr(lib, fn)(None, None) == abi.E_INVALID
    assert lib.skinrt_node_evaluate(None, abi.STAGE_ALL) == abi.E_INVALID
    assert lib.skinrt_node_synchronize(None) == abi.E_INVALID
    assert lib.skinrt_node_download_aabbs(None, None) == abi.E_INVALID
    assert lib.skinrt_node_shard_count(None, None) == abi.E_INVALID
    assert lib.skinrt_node_destroy(None) == abi.OK


def test_graft_entry_build_checks_the_headers_version():
    """build() is the driver's "does it build" step: it must compare the library with the header, not with a literal
    (a stale literal failed the step for a whole round without any test noticing)."""
    src = open(os.path.join(os.path.dirname(HEADER), "..", "__graft_entry__.py")).read()
    assert "SKINRT_ABI_VERSION" in src and not re.search(r"skinrt_abi_version\(\)\s*==\s*\d", src)


# ---- several skins / mesh sections (ABI 6) -------------------------------------------------------------------------

def multi():
    spec = synth.tiny_spec(seed=5, verts=700, joints=12, nodes=20, targets=6, weight_channels=1)
    base = synth.make_rig(spec)
    rig, tmap = synth.make_multi_rig(base)
    return base, rig, tmap


def test_multi_skin_rig_compiles_and_reports_its_layout():
    base, rig, tmap = multi()
    info = api.validate_rig(rig)
    assert (info.skin_count, info.section_count) == (3, 4)
    assert info.joint_count == rig.joint_count == sum(sk.joint_node.shape[0] for sk in rig.skins)
    assert info.palette_count == rig.joint_count + 1            # one entry for the un-skinned section's node
    assert info.target_count == len(tmap) and info.vertex_count == base.vertex_count
    assert info.weight_channel_clips == 1
    one = api.validate_rig(base)
    assert (one.skin_count, one.section_count, one.palette_count) == (1, 1, base.joint_count)


def test_sections_must_tile_vertices_and_targets():
    _, rig, _ = multi()
    rig.sections[1].first_vertex += 1
    expect(rig, abi.E_INVALID, "tile the vertices")
    _, rig, _ = multi()
    rig.sections[-1].vertex_count -= 1
    expect(rig, abi.E_INVALID, "cover every vertex")
    _, rig, _ = multi()
    rig.sections[0].target_count += 1
    expect(rig, abi.E_INVALID, "tile the morph targets")


def test_section_skin_and_node_are_range_checked():
    _, rig, _ = multi()
    rig.sections[0].skin = 3
    expect(rig, abi.E_RANGE, "section skin out of range")
    _, rig, _ = multi()
    rig.sections[2].node = rig.node_count
    expect(rig, abi.E_RANGE, "section node out of range")


def test_joint_indices_are_checked_against_the_vertex_own_skin():
    _, rig, _ = multi()
    sc = rig.sections[1]
    own = rig.skins[sc.skin].joint_node.shape[0]
    assert own < rig.joint_count
    rig.joints[0][sc.first_vertex, 0] = own   # a valid index of the whole palette, but not of this section's skin
    rig.weights[0][sc.first_vertex, 0] = 0.5
    expect(rig, abi.E_RANGE, "joint index out of range")


def test_unskinned_sections_do_not_read_joints_or_weights():
    _, rig, _ = multi()
    sc = rig.sections[-1]
    assert sc.skin == -1 and np.isnan(rig.weights[0][sc.first_vertex]).all() and (rig.joints[0][sc.first_vertex] == 0xFFFF).all()
    api.validate_rig(rig)  # the poisoned sets of the prop are never looked at


def test_weights_channel_must_sit_on_a_morphed_section_node():
    _, rig, _ = multi()
    used = {s.node for s in rig.sections}
    other = next(n for n in range(rig.node_count) if n not in used)
    w = next(ch for ch in rig.clips[0] if ch.path == abi.PATH_WEIGHTS)
    w.target_node = other
    expect(rig, abi.E_INVALID, "without a morphed mesh section")
    _, rig, _ = multi()
    w = next(ch for ch in rig.clips[0] if ch.path == abi.PATH_WEIGHTS)
    rig.clips[0].append(abi.Channel(w.target_node, abi.PATH_WEIGHTS, abi.INTERP_STEP, w.times.copy(), w.values.copy()))
    expect(rig, abi.E_INVALID, "two weights channels")


def test_dense_target_of_a_section_has_the_sections_length():
    _, rig, _ = multi()
    t = next(t for t in rig.targets if t.indices is None)
    t.position = np.concatenate([t.position, np.zeros((1, 3), np.float32)])
    d = rig.to_desc()
    # RigData sizes dense targets by their section; hand the library a wrong count directly
    k = [i for i, x in enumerate(rig.targets) if x is t][0]
    d.targets[k].count += 1
    info = abi.RigInfo()
    rc = api.load_library().skinrt_validate_rig(None, C.byref(d), C.byref(info))
    assert rc == abi.E_INVALID and b"dense morph target" in api.load_library().skinrt_last_error()


def test_shipped_library_is_not_a_measurement_build():
    """skinrt_build_flags() is 0 for the product: no ablation / experiment switch compiled in (they make results wrong by
    design), default tuning.  skinrt_init refuses anything else unless the config opts in."""
    assert api.load_library().skinrt_build_flags() == 0
    assert api.make_config().allow_experimental_build == 0
