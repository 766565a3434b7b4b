"""ctypes mirror of include/skinrt.h plus `RigData`, the numpy form of the flat loader output.

`RigData` carries exactly the fields of `skinrt_rig_desc` (SURVEY.md §8a rows a1, a5-a8, a11,
a13: reference `renderer-asset/src/{animation,node,skin,primitive,mesh}.rs`) as numpy arrays;
`RigData.to_desc()` produces the C struct (and keeps every buffer it points at alive).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

PATH_TRANSLATION, PATH_ROTATION, PATH_SCALE, PATH_WEIGHTS = 0, 1, 2, 3
INTERP_LINEAR, INTERP_STEP, INTERP_CUBICSPLINE = 0, 1, 2
STAGE_POSE, STAGE_SKIN, STAGE_AABB, STAGE_ALL = 1, 2, 4, 7
LAYOUT_COLOR_SKIN, LAYOUT_TEXTURE_SKIN = 0, 1
NO_CLIP = 0xFFFFFFFF

OK, E_INVALID, E_DEVICE, E_NOMEM, E_NONAFFINE, E_RANGE, E_LIMIT = 0, -1, -2, -3, -4, -5, -6

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_u16p = C.POINTER(C.c_uint16)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("cubic_quirk", C.c_uint32),
        ("normalize_outputs", C.c_uint32),
        ("reference_limits", C.c_uint32),
        ("exact_arithmetic", C.c_uint32),
        ("pose_path", C.c_uint32),
        ("skin_path", C.c_uint32),
        ("output_placement", C.c_uint32),
        ("allow_experimental_build", C.c_uint32),
        ("placement_max_filler_gib", C.c_uint32),
    ]


class PlacementInfo(C.Structure):
    _fields_ = [
        ("strategy", C.c_uint32),
        ("reserved", C.c_uint32),
        ("filler_bytes", C.c_uint64),
        ("probe_together_ms", C.c_float),
        ("probe_alone_ms", C.c_float),
        ("spread_gain", C.c_float),
        ("probe_GBps", C.c_float),
    ]


class ChannelDesc(C.Structure):
    _fields_ = [
        ("target_node", C.c_uint32),
        ("path", C.c_uint8),
        ("interpolation", C.c_uint8),
        ("reserved", C.c_uint16),
        ("key_count", C.c_uint32),
        ("times", _f32p),
        ("values", _f32p),
    ]


class ClipDesc(C.Structure):
    _fields_ = [("channel_count", C.c_uint32), ("channels", C.POINTER(ChannelDesc))]


class MorphTargetDesc(C.Structure):
    _fields_ = [
        ("count", C.c_uint32),
        ("indices", _u32p),
        ("position", _f32p),
        ("normal", _f32p),
        ("tangent", _f32p),
    ]


class SkinDesc(C.Structure):
    _fields_ = [("joint_count", C.c_uint32), ("joint_node", _u32p), ("inverse_bind", _f32p)]


class SectionDesc(C.Structure):
    _fields_ = [("node", C.c_uint32), ("skin", C.c_int32), ("first_vertex", C.c_uint32), ("vertex_count", C.c_uint32),
                ("first_target", C.c_uint32), ("target_count", C.c_uint32)]


class SkinLayout(C.Structure):
    _fields_ = [("first_palette_entry", C.c_uint32), ("joint_count", C.c_uint32)]


class SectionLayout(C.Structure):
    _fields_ = [("node", C.c_uint32), ("skin", C.c_int32), ("first_vertex", C.c_uint32), ("vertex_count", C.c_uint32),
                ("first_target", C.c_uint32), ("target_count", C.c_uint32), ("first_palette_entry", C.c_uint32),
                ("reserved", C.c_uint32)]


class RigDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("node_count", C.c_uint32),
        ("parent", _i32p),
        ("translation", _f32p),
        ("rotation", _f32p),
        ("scale", _f32p),
        ("matrix", _f32p),
        ("has_matrix", _u8p),
        ("joint_count", C.c_uint32),
        ("joint_node", _u32p),
        ("inverse_bind", _f32p),
        ("clip_count", C.c_uint32),
        ("clips", C.POINTER(ClipDesc)),
        ("vertex_count", C.c_uint32),
        ("position", _f32p),
        ("normal", _f32p),
        ("tangent", _f32p),
        ("tangent_stride", C.c_uint32),
        ("set_count", C.c_uint32),
        ("joints", C.POINTER(_u16p)),
        ("weights", C.POINTER(_f32p)),
        ("target_count", C.c_uint32),
        ("targets", C.POINTER(MorphTargetDesc)),
        ("default_weights", _f32p),
        ("skin_count", C.c_uint32),
        ("skins", C.POINTER(SkinDesc)),
        ("section_count", C.c_uint32),
        ("sections", C.POINTER(SectionDesc)),
    ]


class RigInfo(C.Structure):
    _fields_ = [
        ("node_count", C.c_uint32), ("joint_count", C.c_uint32), ("vertex_count", C.c_uint32),
        ("set_count", C.c_uint32), ("target_count", C.c_uint32), ("clip_count", C.c_uint32),
        ("level_count", C.c_uint32), ("tile_count", C.c_uint32), ("tile_vertices", C.c_uint32),
        ("max_tile_joints", C.c_uint32), ("lds_tiles", C.c_uint32), ("max_channels", C.c_uint32),
        ("morph_entries", C.c_uint64), ("device_bytes", C.c_uint64),
        ("exact_arithmetic", C.c_uint32), ("bind_extent", C.c_float),
        ("weight_channel_clips", C.c_uint32), ("skip_weight_sets", C.c_uint32),
        ("skin_count", C.c_uint32), ("section_count", C.c_uint32), ("palette_count", C.c_uint32), ("handed_over_tiles", C.c_uint32),
    ]


def _ptr(a: Optional[np.ndarray], typ):
    if a is None:
        return C.cast(None, typ)
    return a.ctypes.data_as(typ)


def _arr(a, dtype, shape=None) -> Optional[np.ndarray]:
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None:
        a = a.reshape(shape)
    return a


@dataclass
class Channel:
    """AnimationChannelAsset (renderer-asset/src/animation.rs:29-34), SoA keyframes."""
    target_node: int
    path: int
    interpolation: int
    times: np.ndarray   # [Kf] f32
    values: np.ndarray  # [Kf][w] or [Kf][3][w] f32


@dataclass
class MorphTarget:
    """PrimitiveAssetMorphTarget (renderer-asset/src/primitive.rs:33-38); dense or compacted."""
    position: Optional[np.ndarray] = None  # [count][3]
    normal: Optional[np.ndarray] = None    # [count][3]
    tangent: Optional[np.ndarray] = None   # [count][tangent_stride]
    indices: Optional[np.ndarray] = None   # [count] u32, None = dense


@dataclass
class Skin:
    """SkinAsset (renderer-asset/src/skin.rs:5-11): joints as flat node indices, inverse bind matrices or None."""
    joint_node: np.ndarray                      # [J] u32
    inverse_bind: Optional[np.ndarray] = None   # [J][16]


@dataclass
class Section:
    """A NodeAsset with a mesh (renderer-asset/src/node.rs:89-99): its vertices and morph targets in the rig's arrays,
    its skin (index into RigData.skins) or -1 for an un-skinned mesh placed by the node's world matrix."""
    node: int
    skin: int
    first_vertex: int
    vertex_count: int
    first_target: int = 0
    target_count: int = 0


@dataclass
class RigData:
    parent: np.ndarray                     # [N] i32
    translation: np.ndarray                # [N][3]
    rotation: np.ndarray                   # [N][4]
    scale: np.ndarray                      # [N][3]
    joint_node: np.ndarray                 # [J] u32
    position: np.ndarray                   # [V][3]
    normal: np.ndarray                     # [V][3]
    tangent: np.ndarray                    # [V][ts]
    joints: List[np.ndarray]               # K x [V][4] u16
    weights: List[np.ndarray]              # K x [V][4] f32
    inverse_bind: Optional[np.ndarray] = None   # [J][16]
    matrix: Optional[np.ndarray] = None         # [N][16]
    has_matrix: Optional[np.ndarray] = None     # [N] u8
    clips: List[List[Channel]] = field(default_factory=list)
    targets: List[MorphTarget] = field(default_factory=list)
    default_weights: Optional[np.ndarray] = None
    name: str = "rig"
    skins: Optional[List[Skin]] = None          # several skins on one node tree (then joint_node / inverse_bind above are unused)
    sections: Optional[List[Section]] = None    # mesh instances; None = one section: all vertices, all targets, skin 0

    def __post_init__(self):
        self.parent = _arr(self.parent, np.int32)
        n = self.parent.shape[0]
        self.translation = _arr(self.translation, np.float32, (n, 3))
        self.rotation = _arr(self.rotation, np.float32, (n, 4))
        self.scale = _arr(self.scale, np.float32, (n, 3))
        self.matrix = _arr(self.matrix, np.float32, (n, 16)) if self.matrix is not None else None
        self.has_matrix = _arr(self.has_matrix, np.uint8)
        self.joint_node = _arr(self.joint_node, np.uint32)
        j = self.joint_node.shape[0]
        self.inverse_bind = _arr(self.inverse_bind, np.float32, (j, 16)) if self.inverse_bind is not None else None
        self.position = _arr(self.position, np.float32)
        v = self.position.shape[0] if self.position.ndim == 2 else self.position.size // 3
        self.position = self.position.reshape(v, 3)
        self.normal = _arr(self.normal, np.float32, (v, 3))
        self.tangent = _arr(self.tangent, np.float32)
        self.tangent = self.tangent.reshape(v, -1) if v else np.zeros((0, 3), np.float32)  # a skeleton without a mesh (pose only)
        self.joints = [_arr(a, np.uint16, (v, 4)) for a in self.joints]
        self.weights = [_arr(a, np.float32, (v, 4)) for a in self.weights]
        self.default_weights = _arr(self.default_weights, np.float32)
        for t in self.targets:
            t.position = _arr(t.position, np.float32)
            t.normal = _arr(t.normal, np.float32)
            t.tangent = _arr(t.tangent, np.float32)
            t.indices = _arr(t.indices, np.uint32)
        for clip in self.clips:
            for ch in clip:
                ch.times = _arr(ch.times, np.float32)
                ch.values = _arr(ch.values, np.float32)
        for sk in self.skins or []:
            sk.joint_node = _arr(sk.joint_node, np.uint32)
            sk.inverse_bind = _arr(sk.inverse_bind, np.float32, (sk.joint_node.shape[0], 16)) if sk.inverse_bind is not None else None

    # sizes ------------------------------------------------------------------------------
    @property
    def node_count(self): return int(self.parent.shape[0])
    @property
    def joint_count(self):
        """joints of all skins together"""
        if self.skins is not None:
            return sum(int(sk.joint_node.shape[0]) for sk in self.skins)
        return int(self.joint_node.shape[0])

    def skin_list(self) -> List[Skin]:
        """the skins, whichever form the rig uses (short form: one skin, or none without joints)"""
        if self.skins is not None:
            return self.skins
        return [Skin(self.joint_node, self.inverse_bind)] if self.joint_node.shape[0] else []

    def section_list(self) -> List[Section]:
        if self.sections is not None:
            return self.sections
        if not self.vertex_count and not self.target_count:
            return []
        return [Section(0, 0 if self.skin_list() else -1, 0, self.vertex_count, 0, self.target_count)]

    def rigid_nodes(self) -> List[int]:
        """distinct nodes of un-skinned sections with vertices, in section order: one palette entry each after the skins' joints"""
        out: List[int] = []
        for sc in self.section_list():
            if sc.skin < 0 and sc.vertex_count and sc.node not in out:
                out.append(sc.node)
        return out

    @property
    def palette_count(self): return self.joint_count + len(self.rigid_nodes())

    def skin_first(self, skin: int) -> int:
        return sum(int(sk.joint_node.shape[0]) for sk in self.skin_list()[:skin])
    @property
    def vertex_count(self): return int(self.position.shape[0])
    @property
    def set_count(self): return len(self.joints)
    @property
    def target_count(self): return len(self.targets)
    @property
    def tangent_stride(self): return int(self.tangent.shape[1]) if self.vertex_count else 3
    @property
    def max_channels(self): return max([len(c) for c in self.clips], default=0)

    def clip_length(self, clip: int) -> float:
        """max over channels of AnimationChannelAsset.length (loader/gltf/mod.rs:1243-1247)."""
        return max([float(ch.times.max()) for ch in self.clips[clip]], default=0.0)

    def _target_section(self, m: int) -> Section:
        for sc in self.section_list():
            if sc.first_target <= m < sc.first_target + sc.target_count:
                return sc
        raise IndexError(f"morph target {m} belongs to no section")

    def to_desc(self) -> "RigDesc":
        d = RigDesc()
        keep = []
        d.struct_size = C.sizeof(RigDesc)
        d.node_count = self.node_count
        d.parent = _ptr(self.parent, _i32p)
        d.translation = _ptr(self.translation, _f32p)
        d.rotation = _ptr(self.rotation, _f32p)
        d.scale = _ptr(self.scale, _f32p)
        d.matrix = _ptr(self.matrix, _f32p)
        d.has_matrix = _ptr(self.has_matrix, _u8p)
        d.joint_count = self.joint_count
        d.joint_node = _ptr(self.joint_node, _u32p)
        d.inverse_bind = _ptr(self.inverse_bind, _f32p)
        d.clip_count = len(self.clips)
        clip_arr = (ClipDesc * max(1, len(self.clips)))()
        for ci, clip in enumerate(self.clips):
            ch_arr = (ChannelDesc * max(1, len(clip)))()
            for k, ch in enumerate(clip):
                ch_arr[k].target_node = ch.target_node
                ch_arr[k].path = ch.path
                ch_arr[k].interpolation = ch.interpolation
                ch_arr[k].key_count = ch.times.shape[0]
                ch_arr[k].times = _ptr(ch.times, _f32p)
                ch_arr[k].values = _ptr(ch.values, _f32p)
            keep.append(ch_arr)
            clip_arr[ci].channel_count = len(clip)
            clip_arr[ci].channels = C.cast(ch_arr, C.POINTER(ChannelDesc))
        keep.append(clip_arr)
        d.clips = C.cast(clip_arr, C.POINTER(ClipDesc))
        d.vertex_count = self.vertex_count
        d.position = _ptr(self.position, _f32p)
        d.normal = _ptr(self.normal, _f32p)
        d.tangent = _ptr(self.tangent, _f32p)
        d.tangent_stride = self.tangent_stride
        d.set_count = self.set_count
        jp = (_u16p * max(1, self.set_count))(*[_ptr(a, _u16p) for a in self.joints])
        wp = (_f32p * max(1, self.set_count))(*[_ptr(a, _f32p) for a in self.weights])
        keep += [jp, wp]
        d.joints = C.cast(jp, C.POINTER(_u16p))
        d.weights = C.cast(wp, C.POINTER(_f32p))
        d.target_count = len(self.targets)
        t_arr = (MorphTargetDesc * max(1, len(self.targets)))()
        for k, t in enumerate(self.targets):
            cnt = self._target_section(k).vertex_count if t.indices is None else int(t.indices.shape[0])
            t_arr[k].count = cnt
            t_arr[k].indices = _ptr(t.indices, _u32p)
            t_arr[k].position = _ptr(t.position, _f32p)
            t_arr[k].normal = _ptr(t.normal, _f32p)
            t_arr[k].tangent = _ptr(t.tangent, _f32p)
        keep.append(t_arr)
        d.targets = C.cast(t_arr, C.POINTER(MorphTargetDesc))
        d.default_weights = _ptr(self.default_weights, _f32p)
        if self.skins is not None:
            sk_arr = (SkinDesc * max(1, len(self.skins)))()
            for k, sk in enumerate(self.skins):
                sk_arr[k].joint_count = sk.joint_node.shape[0]
                sk_arr[k].joint_node = _ptr(sk.joint_node, _u32p)
                sk_arr[k].inverse_bind = _ptr(sk.inverse_bind, _f32p)
            keep.append(sk_arr)
            d.skin_count = len(self.skins)
            d.skins = C.cast(sk_arr, C.POINTER(SkinDesc))
            d.joint_count = 0
            d.joint_node = C.cast(None, _u32p)
            d.inverse_bind = C.cast(None, _f32p)
        if self.sections is not None:
            sc_arr = (SectionDesc * max(1, len(self.sections)))()
            for k, sc in enumerate(self.sections):
                sc_arr[k].node, sc_arr[k].skin = sc.node, sc.skin
                sc_arr[k].first_vertex, sc_arr[k].vertex_count = sc.first_vertex, sc.vertex_count
                sc_arr[k].first_target, sc_arr[k].target_count = sc.first_target, sc.target_count
            keep.append(sc_arr)
            d.section_count = len(self.sections)
            d.sections = C.cast(sc_arr, C.POINTER(SectionDesc))
        d._keep = (keep, self)  # keep python owners alive as long as the struct lives
        return d
