"""ctypes access to libskinrt_host.so (renderer_amd/host): the native .glb loader + scene
flattener.  `load_glb(bytes)` returns the flat rig as a RigData (arrays copied out of the C++
object), plus the glTF node index of every flat node and the clip lengths."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Tuple

import numpy as np

from . import abi
from .abi import Channel, MorphTarget, RigData, RigDesc, Section, Skin

_HOST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")
_PATH = os.path.join(_HOST, "build", "libskinrt_host.so")
_lib = None


class HostError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"skinrt_host error {code}: {message}")
        self.code, self.message = code, message


def load_library() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.run(["make", "-C", _HOST], check=True, capture_output=True)
        lib = C.CDLL(_PATH)
        lib.skinrt_host_load_glb.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
        lib.skinrt_host_load_glb.restype = C.c_int
        lib.skinrt_host_load_pmx.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
        lib.skinrt_host_load_pmx.restype = C.c_int
        lib.skinrt_host_rig_pmx_indices.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        lib.skinrt_host_rig_pmx_indices.restype = C.POINTER(C.c_uint32)
        lib.skinrt_host_rig_pmx_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint32 * 10)]
        lib.skinrt_host_rig_pmx_counts.restype = None
        lib.skinrt_host_rig_desc.argtypes = [C.c_void_p]
        lib.skinrt_host_rig_desc.restype = C.POINTER(RigDesc)
        lib.skinrt_host_rig_node_gltf_index.argtypes = [C.c_void_p, C.c_uint32]
        lib.skinrt_host_rig_node_gltf_index.restype = C.c_uint32
        lib.skinrt_host_rig_clip_length.argtypes = [C.c_void_p, C.c_uint32]
        lib.skinrt_host_rig_clip_length.restype = C.c_float
        lib.skinrt_host_rig_sanitized_tangent_deltas.argtypes = [C.c_void_p]
        lib.skinrt_host_rig_sanitized_tangent_deltas.restype = C.c_uint64
        lib.skinrt_host_free_rig.argtypes = [C.c_void_p]
        lib.skinrt_host_free_rig.restype = None
        lib.skinrt_host_last_error.restype = C.c_char_p
        lib.skinrt_host_calculate_tangent.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        lib.skinrt_host_calculate_tangent.restype = C.c_int
        _lib = lib
    return _lib


def _np(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return np.zeros(shape, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).reshape(shape).copy()


def _rig_from_desc(d: RigDesc) -> RigData:
    n, j, v = d.node_count, d.joint_count, d.vertex_count
    ts = d.tangent_stride
    skins = sections = None
    if d.skin_count or d.section_count:  # the flattener's form: every skin and every mesh node of the scenes
        skins = []
        for k in range(d.skin_count):
            sk = d.skins[k]
            skins.append(Skin(_np(sk.joint_node, (sk.joint_count,), np.uint32),
                              _np(sk.inverse_bind, (sk.joint_count, 16), np.float32) if sk.inverse_bind else None))
        sections = [Section(sc.node, sc.skin, sc.first_vertex, sc.vertex_count, sc.first_target, sc.target_count)
                    for sc in (d.sections[k] for k in range(d.section_count))]

    def weights_width(node):
        if sections is None:
            return d.target_count
        return next(sc.target_count for sc in sections if sc.node == node and sc.target_count)
    clips: List[List[Channel]] = []
    for c in range(d.clip_count):
        cl = d.clips[c]
        chans = []
        for k in range(cl.channel_count):
            ch = cl.channels[k]
            w = weights_width(ch.target_node) if ch.path == abi.PATH_WEIGHTS else (4 if ch.path == abi.PATH_ROTATION else 3)
            w *= 3 if ch.interpolation == abi.INTERP_CUBICSPLINE else 1
            chans.append(Channel(ch.target_node, ch.path, ch.interpolation, _np(ch.times, (ch.key_count,), np.float32),
                                 _np(ch.values, (ch.key_count, w), np.float32)))
        clips.append(chans)
    targets = []
    for m in range(d.target_count):
        t = d.targets[m]
        targets.append(MorphTarget(position=_np(t.position, (t.count, 3), np.float32),
                                   normal=_np(t.normal, (t.count, 3), np.float32) if t.normal else None,
                                   tangent=_np(t.tangent, (t.count, ts), np.float32) if t.tangent else None,
                                   indices=_np(t.indices, (t.count,), np.uint32) if t.indices else None))
    return RigData(parent=_np(d.parent, (n,), np.int32), translation=_np(d.translation, (n, 3), np.float32),
                   rotation=_np(d.rotation, (n, 4), np.float32), scale=_np(d.scale, (n, 3), np.float32),
                   matrix=_np(d.matrix, (n, 16), np.float32) if d.matrix else None,
                   has_matrix=_np(d.has_matrix, (n,), np.uint8) if d.has_matrix else None,
                   # with skins[] the short-form fields are unused; they mirror skin 0 for callers that look at one skin
                   joint_node=_np(d.joint_node, (j,), np.uint32) if skins is None else (skins[0].joint_node if skins else np.zeros(0, np.uint32)),
                   inverse_bind=(_np(d.inverse_bind, (j, 16), np.float32) if d.inverse_bind else None) if skins is None
                   else (skins[0].inverse_bind if skins else None),
                   skins=skins, sections=sections,
                   position=_np(d.position, (v, 3), np.float32), normal=_np(d.normal, (v, 3), np.float32),
                   tangent=_np(d.tangent, (v, ts), np.float32),
                   joints=[_np(d.joints[k], (v, 4), np.uint16) for k in range(d.set_count)],
                   weights=[_np(d.weights[k], (v, 4), np.float32) for k in range(d.set_count)],
                   clips=clips, targets=targets,
                   default_weights=_np(d.default_weights, (d.target_count,), np.float32) if d.default_weights else None,
                   name="glb")


def load_glb(data: bytes, scene: int = -1) -> Tuple[RigData, np.ndarray, np.ndarray]:
    """(rig, gltf node index per flat node, clip lengths)"""
    lib = load_library()
    h = C.c_void_p()
    rc = lib.skinrt_host_load_glb(data, len(data), scene, C.byref(h))
    if rc != 0:
        raise HostError(rc, lib.skinrt_host_last_error().decode("utf-8", "replace"))
    try:
        d = lib.skinrt_host_rig_desc(h).contents
        rig = _rig_from_desc(d)
        order = np.array([lib.skinrt_host_rig_node_gltf_index(h, k) for k in range(d.node_count)], np.uint32)
        lengths = np.array([lib.skinrt_host_rig_clip_length(h, c) for c in range(d.clip_count)], np.float32)
        rig.sanitized_tangent_deltas = int(lib.skinrt_host_rig_sanitized_tangent_deltas(h))
        return rig, order, lengths
    finally:
        lib.skinrt_host_free_rig(h)


def calculate_tangent(mode: int, positions, indices=None) -> np.ndarray:
    lib = load_library()
    p = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    out = np.zeros_like(p)
    idx = None if indices is None else np.ascontiguousarray(indices, np.uint32)
    rc = lib.skinrt_host_calculate_tangent(mode, p.ctypes.data, p.shape[0], None if idx is None else idx.ctypes.data,
                                           0 if idx is None else idx.size, out.ctypes.data)
    if rc != 0:
        raise HostError(rc, lib.skinrt_host_last_error().decode("utf-8", "replace"))
    return out


def load_pmx(data: bytes):
    """PMX 2.0 model -> (rig, PMX bone index per flat node, surface indices, counts dict).  See
    renderer_amd/host/pmx_loader.hpp for the (build-defined) mapping."""
    lib = load_library()
    h = C.c_void_p()
    rc = lib.skinrt_host_load_pmx(data, len(data), C.byref(h))
    if rc != 0:
        raise HostError(rc, lib.skinrt_host_last_error().decode("utf-8", "replace"))
    try:
        d = lib.skinrt_host_rig_desc(h).contents
        rig = _rig_from_desc(d)
        rig.name = "pmx"
        order = np.array([lib.skinrt_host_rig_node_gltf_index(h, k) for k in range(d.node_count)], np.uint32)
        n = C.c_uint32(0)
        ip = lib.skinrt_host_rig_pmx_indices(h, C.byref(n))
        indices = _np(ip, (n.value,), np.uint32)
        cnt = (C.c_uint32 * 10)()
        lib.skinrt_host_rig_pmx_counts(h, C.byref(cnt))
        names = ("vertices", "triangles", "materials", "bones", "morphs", "vertex_morphs", "bdef1", "bdef2", "bdef4", "approximated")
        return rig, order, indices, dict(zip(names, [int(x) for x in cnt]))
    finally:
        lib.skinrt_host_free_rig(h)
