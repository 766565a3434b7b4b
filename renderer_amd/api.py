"""Python host side over the C ABI (include/skinrt.h), via ctypes.

There is no Rust toolchain in this image, so the caller above the boundary is Python here;
the Rust-side binding a maintainer would add is in INTEGRATION.md.  These classes mirror the
reference's objects on this path: `Rig` ~ the uploaded (Vec<SceneAsset>, Vec<AnimationAsset>)
(renderer-asset/src/loader/gltf/mod.rs:199), `Crowd.evaluate` ~ the body of `Entity::render`
(renderer/src/client/entity/mod.rs:34), `Crowd.skin_uniform` ~ `SkinUniformBuffer::update`
(renderer/src/renderer/uniform/skin.rs:59-71).

The HIP library is mandatory: there is no CPU fallback.  If libskinrt.so is missing or no GPU is
present, construction fails loudly (SkinrtError).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

from . import abi
from .abi import Config, PlacementInfo, RigData, RigDesc, RigInfo, SectionLayout, SkinLayout

_LIB_PATH = os.environ.get("SKINRT_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libskinrt.so")
_lib = None

# every symbol include/skinrt.h declares
EXPORTS = [
    "skinrt_abi_version", "skinrt_last_error", "skinrt_init", "skinrt_destroy_ctx", "skinrt_upload_rig",
    "skinrt_validate_rig", "skinrt_destroy_rig", "skinrt_get_rig_info", "skinrt_create_crowd", "skinrt_get_output_placement",
    "skinrt_destroy_crowd", "skinrt_set_clocks", "skinrt_set_morph_weights", "skinrt_set_root_transforms",
    "skinrt_evaluate", "skinrt_synchronize", "skinrt_last_stage_ms", "skinrt_get_stream",
    "skinrt_get_positions", "skinrt_get_normals", "skinrt_get_tangents", "skinrt_get_aabbs",
    "skinrt_get_palette", "skinrt_get_world", "skinrt_get_keyframe_index", "skinrt_download",
    "skinrt_export_skin_uniform", "skinrt_export_interleaved", "skinrt_calculate_tangent",
    "skinrt_calculate_tangent_noalloc", "skinrt_export_skin_uniform_device", "skinrt_export_interleaved_device",
    "skinrt_shard_range", "skinrt_build_flags", "skinrt_rig_skin_layout", "skinrt_rig_section_layout",
    "skinrt_set_clocks_async", "skinrt_advance_clocks", "skinrt_get_clocks",
    "skinrt_export_skin_uniform_of", "skinrt_export_interleaved_section", "skinrt_export_instance_uniform",
    "skinrt_export_skin_uniform_of_device", "skinrt_export_interleaved_section_device", "skinrt_export_instance_uniform_device",
    "skinrt_export_skin_storage_device",
    "skinrt_create_topology", "skinrt_destroy_topology", "skinrt_recompute_normals",
    "skinrt_node_create", "skinrt_node_destroy", "skinrt_node_shard_count", "skinrt_node_get_shard", "skinrt_node_set_clocks",
    "skinrt_node_set_morph_weights", "skinrt_node_set_root_transforms", "skinrt_node_evaluate", "skinrt_node_synchronize",
    "skinrt_node_download_aabbs",
]


class SkinrtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"skinrt error {code}: {message}")
        self.code = code
        self.message = message


def lib_path() -> str:
    return _LIB_PATH


def load_library() -> C.CDLL:
    """Load libskinrt.so (built in-tree by __graft_entry__.build()).  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise SkinrtError(abi.E_DEVICE, f"{_LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(_LIB_PATH)
    vp = C.c_void_p
    lib.skinrt_abi_version.restype = C.c_int
    lib.skinrt_last_error.restype = C.c_char_p
    lib.skinrt_build_flags.restype = C.c_uint32
    lib.skinrt_build_flags.argtypes = []
    sigs = {
        "skinrt_init": [C.POINTER(Config), C.POINTER(vp)],
        "skinrt_destroy_ctx": [vp],
        "skinrt_upload_rig": [vp, C.POINTER(RigDesc), C.POINTER(vp)],
        "skinrt_validate_rig": [C.POINTER(Config), C.POINTER(RigDesc), C.POINTER(RigInfo)],
        "skinrt_destroy_rig": [vp],
        "skinrt_get_rig_info": [vp, C.POINTER(RigInfo)],
        "skinrt_create_crowd": [vp, vp, C.c_uint32, C.POINTER(vp)],
        "skinrt_destroy_crowd": [vp],
        "skinrt_get_output_placement": [vp, C.POINTER(PlacementInfo)],
        "skinrt_set_clocks": [vp, vp, vp],
        "skinrt_set_morph_weights": [vp, vp],
        "skinrt_set_root_transforms": [vp, vp],
        "skinrt_evaluate": [vp, C.c_uint32],
        "skinrt_synchronize": [vp],
        "skinrt_last_stage_ms": [vp, C.POINTER(C.c_float * 4)],
        "skinrt_get_stream": [vp, C.POINTER(vp)],
        "skinrt_download": [vp, vp, vp, C.c_uint64],
        "skinrt_export_skin_uniform": [vp, C.c_uint32, vp],
        "skinrt_export_interleaved": [vp, C.c_uint32, C.c_uint32, vp, vp],
        "skinrt_export_skin_uniform_device": [vp, C.c_uint32, C.c_uint32, vp],
        "skinrt_export_interleaved_device": [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp],
        "skinrt_rig_skin_layout": [vp, C.c_uint32, C.POINTER(SkinLayout)],
        "skinrt_rig_section_layout": [vp, C.c_uint32, C.POINTER(SectionLayout)],
        "skinrt_set_clocks_async": [vp, vp, vp],
        "skinrt_advance_clocks": [vp, C.c_float, vp],
        "skinrt_export_skin_uniform_of": [vp, C.c_uint32, C.c_uint32, vp],
        "skinrt_export_interleaved_section": [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp],
        "skinrt_export_instance_uniform": [vp, C.c_uint32, C.c_uint32, vp],
        "skinrt_export_skin_uniform_of_device": [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp],
        "skinrt_export_skin_storage_device": [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp],
        "skinrt_export_interleaved_section_device": [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp],
        "skinrt_export_instance_uniform_device": [vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp],
        "skinrt_shard_range": [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)],
        "skinrt_calculate_tangent": [vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, vp, C.POINTER(C.c_float)],
        "skinrt_calculate_tangent_noalloc": [vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, vp],
        "skinrt_create_topology": [vp, C.c_uint32, vp, C.c_uint32, C.POINTER(vp)],
        "skinrt_destroy_topology": [vp],
        "skinrt_recompute_normals": [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32],
        "skinrt_node_create": [C.POINTER(Config), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(RigDesc), C.c_uint32, C.POINTER(vp)],
        "skinrt_node_destroy": [vp],
        "skinrt_node_shard_count": [vp, C.POINTER(C.c_uint32)],
        "skinrt_node_get_shard": [vp, C.c_uint32, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)],
        "skinrt_node_set_clocks": [vp, vp, vp],
        "skinrt_node_set_morph_weights": [vp, vp],
        "skinrt_node_set_root_transforms": [vp, vp],
        "skinrt_node_evaluate": [vp, C.c_uint32],
        "skinrt_node_synchronize": [vp],
        "skinrt_node_download_aabbs": [vp, vp],
    }
    for g in ("positions", "normals", "tangents", "aabbs", "palette", "world", "keyframe_index", "clocks"):
        sigs[f"skinrt_get_{g}"] = [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    _lib = lib
    return lib


def _check(rc: int):
    if rc != 0:
        raise SkinrtError(rc, load_library().skinrt_last_error().decode("utf-8", "replace"))


def make_config(device: int = 0, cubic_quirk: int = 1, normalize_outputs: int = 0, reference_limits: int = 0,
                exact_arithmetic: int = 0, pose_path: int = 0, skin_path: int = 0, output_placement: int = 0,
                allow_experimental_build: Optional[int] = None, placement_max_filler_gib: int = 0) -> Config:
    cfg = Config()
    cfg.struct_size = C.sizeof(Config)
    cfg.device = device
    cfg.cubic_quirk = cubic_quirk
    cfg.normalize_outputs = normalize_outputs
    cfg.reference_limits = reference_limits
    cfg.exact_arithmetic = exact_arithmetic
    cfg.pose_path = pose_path
    cfg.skin_path = skin_path
    cfg.output_placement = output_placement
    # measurement builds (SKINRT_LIB pointing at a variant library) are refused by skinrt_init unless asked for
    if allow_experimental_build is None:
        allow_experimental_build = 1 if os.environ.get("SKINRT_ALLOW_EXPERIMENTAL") == "1" else 0
    cfg.allow_experimental_build = allow_experimental_build
    cfg.placement_max_filler_gib = placement_max_filler_gib
    return cfg


def shard_range(instances: int, rank: int, world: int) -> Tuple[int, int]:
    """(first, count) of rank's instance range — the C ABI's rule (skinrt_shard_range)."""
    lo, n = C.c_uint32(), C.c_uint32()
    _check(load_library().skinrt_shard_range(instances, rank, world, C.byref(lo), C.byref(n)))
    return int(lo.value), int(n.value)


def validate_rig(rig: RigData, config: Optional[Config] = None) -> RigInfo:
    """Host-only validation + compilation (no GPU needed)."""
    lib = load_library()
    info = RigInfo()
    desc = rig.to_desc()
    _check(lib.skinrt_validate_rig(C.byref(config) if config is not None else None, C.byref(desc), C.byref(info)))
    return info


class Context:
    def __init__(self, device: int = 0, **cfg):
        self.lib = load_library()
        self.config = make_config(device=device, **cfg)
        self.handle = C.c_void_p()
        _check(self.lib.skinrt_init(C.byref(self.config), C.byref(self.handle)))

    def synchronize(self):
        _check(self.lib.skinrt_synchronize(self.handle))

    @property
    def stream(self) -> int:
        s = C.c_void_p()
        _check(self.lib.skinrt_get_stream(self.handle, C.byref(s)))
        return s.value or 0

    def download(self, dev_ptr: int, count: int, dtype) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        _check(self.lib.skinrt_download(self.handle, C.c_void_p(dev_ptr), out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def calculate_tangent(self, mode: int, positions, indices=None, with_ms: bool = False):
        """calculate_tangent (renderer-asset/src/tangent.rs:15-58) on the device; returns [V][3] float32
        (and the kernels' time in ms when with_ms)."""
        p = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        idx = None if indices is None else np.ascontiguousarray(indices, np.uint32).reshape(-1)
        out = np.zeros_like(p)
        ms = C.c_float(0.0)
        _check(self.lib.skinrt_calculate_tangent(self.handle, mode, p.ctypes.data_as(C.c_void_p), p.shape[0],
                                                 None if idx is None else idx.ctypes.data_as(C.c_void_p),
                                                 0 if idx is None else idx.shape[0], out.ctypes.data_as(C.c_void_p), C.byref(ms)))
        return (out, float(ms.value)) if with_ms else out

    def calculate_tangent_noalloc(self, mode: int, dev_positions: int, vertex_count: int, dev_indices: int, index_count: int,
                                  dev_out: int):
        """skinrt_calculate_tangent_noalloc: device pointers in, device pointer out, asynchronous on the ctx stream."""
        _check(self.lib.skinrt_calculate_tangent_noalloc(self.handle, mode, C.c_void_p(dev_positions), vertex_count,
                                                         C.c_void_p(dev_indices) if dev_indices else None, index_count,
                                                         C.c_void_p(dev_out)))

    def close(self):
        if self.handle:
            self.lib.skinrt_destroy_ctx(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Rig:
    def __init__(self, ctx: Context, data: RigData):
        self.ctx = ctx
        self.data = data
        self.handle = C.c_void_p()
        desc = data.to_desc()
        _check(ctx.lib.skinrt_upload_rig(ctx.handle, C.byref(desc), C.byref(self.handle)))

    @property
    def info(self) -> RigInfo:
        info = RigInfo()
        _check(self.ctx.lib.skinrt_get_rig_info(self.handle, C.byref(info)))
        return info

    def skin_layout(self, skin: int) -> SkinLayout:
        out = SkinLayout()
        _check(self.ctx.lib.skinrt_rig_skin_layout(self.handle, skin, C.byref(out)))
        return out

    def section_layout(self, section: int) -> SectionLayout:
        out = SectionLayout()
        _check(self.ctx.lib.skinrt_rig_section_layout(self.handle, section, C.byref(out)))
        return out

    def close(self):
        if self.handle:
            self.ctx.lib.skinrt_destroy_rig(self.handle)
            self.handle = C.c_void_p()


class Topology:
    """A mesh's triangle list prepared once for per-frame normals (skinrt_create_topology)."""

    def __init__(self, ctx: Context, vertex_count: int, indices=None):
        self.ctx = ctx
        self.handle = C.c_void_p()
        idx = None if indices is None else np.ascontiguousarray(indices, np.uint32).reshape(-1)
        _check(ctx.lib.skinrt_create_topology(ctx.handle, int(vertex_count), None if idx is None else idx.ctypes.data_as(C.c_void_p),
                                              0 if idx is None else idx.shape[0], C.byref(self.handle)))

    def close(self):
        if self.handle:
            self.ctx.lib.skinrt_destroy_topology(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Crowd:
    def __init__(self, ctx: Context, rig: Rig, instances: int):
        self.ctx, self.rig, self.instances = ctx, rig, int(instances)
        self.handle = C.c_void_p()
        _check(ctx.lib.skinrt_create_crowd(ctx.handle, rig.handle, self.instances, C.byref(self.handle)))

    @property
    def exact(self) -> bool:
        """True when the skin stage runs contraction-free on this crowd's rig (bit-equal to the fp32 oracle)."""
        return self.rig.info.exact_arithmetic == 1

    @property
    def exact_positions(self) -> bool:
        """positions carry the fp32 oracle's bits (exact_arithmetic 1, or 2: normals / tangents fused)"""
        return self.rig.info.exact_arithmetic != 0

    @property
    def placement(self) -> dict:
        """What skinrt_create_crowd did with the output arrays and what it measured."""
        info = PlacementInfo()
        _check(self.ctx.lib.skinrt_get_output_placement(self.handle, C.byref(info)))
        names = {0: "plain", 1: "spread", 2: "searched-none-found", 3: "search-out-of-room"}
        return {"strategy": names.get(info.strategy, str(info.strategy)), "filler_GiB": info.filler_bytes / 2 ** 30,
                "probe_together_ms": info.probe_together_ms, "probe_alone_ms": info.probe_alone_ms,
                "spread_gain": info.spread_gain, "probe_GBps": info.probe_GBps}

    # inputs ------------------------------------------------------------------------------
    def set_clocks(self, tau, clip_id=None):
        tau = np.ascontiguousarray(tau, dtype=np.float32)
        assert tau.shape == (self.instances,)
        cp = None
        if clip_id is not None:
            clip_id = np.ascontiguousarray(clip_id, dtype=np.uint32)
            assert clip_id.shape == (self.instances,)
            cp = clip_id.ctypes.data_as(C.c_void_p)
        _check(self.ctx.lib.skinrt_set_clocks(self.handle, tau.ctypes.data_as(C.c_void_p), cp))

    def set_clocks_async(self, tau_ptr: int, clip_ptr: int = 0):
        """skinrt_set_clocks_async: device (or GPU-visible host) addresses; stream-ordered, does not block."""
        _check(self.ctx.lib.skinrt_set_clocks_async(self.handle, C.c_void_p(tau_ptr), C.c_void_p(clip_ptr) if clip_ptr else None))

    def advance_clocks(self, dt: float, rate_ptr: int = 0):
        """skinrt_advance_clocks: tau = fmodf(tau + dt * rate, clip_length) on the device; rate_ptr: device [I] f32 or 0."""
        _check(self.ctx.lib.skinrt_advance_clocks(self.handle, C.c_float(dt), C.c_void_p(rate_ptr) if rate_ptr else None))

    def clocks(self) -> np.ndarray:
        return self._fetch("clocks", np.float32, (self.instances,))

    def set_morph_weights(self, weights):
        wp = None
        if weights is not None:
            weights = np.ascontiguousarray(weights, dtype=np.float32)
            assert weights.shape == (self.instances, self.rig.data.target_count)
            wp = weights.ctypes.data_as(C.c_void_p)
        _check(self.ctx.lib.skinrt_set_morph_weights(self.handle, wp))

    def set_root_transforms(self, matrices):
        mp = None
        if matrices is not None:
            matrices = np.ascontiguousarray(matrices, dtype=np.float32)
            assert matrices.shape == (self.instances, 16)
            mp = matrices.ctypes.data_as(C.c_void_p)
        _check(self.ctx.lib.skinrt_set_root_transforms(self.handle, mp))

    # evaluation ----------------------------------------------------------------------------
    def evaluate(self, stages: int = abi.STAGE_ALL, sync: bool = False):
        _check(self.ctx.lib.skinrt_evaluate(self.handle, stages))
        if sync:
            self.ctx.synchronize()

    def stage_ms(self) -> Tuple[float, float, float, float]:
        ms = (C.c_float * 4)()
        _check(self.ctx.lib.skinrt_last_stage_ms(self.handle, C.byref(ms)))
        return tuple(float(x) for x in ms)

    def recompute_normals(self, topology: "Topology", first: int = 0, count: Optional[int] = None, output: int = 0, sync: bool = False):
        """skinrt_recompute_normals: calculate_tangent (tangent.rs:15-58) of every instance's skinned positions, written
        over its normals (output 0) or tangents (1); stream-ordered behind the evaluate."""
        n = self.instances - first if count is None else count
        _check(self.ctx.lib.skinrt_recompute_normals(self.handle, topology.handle, first, n, output))
        if sync:
            self.ctx.synchronize()

    # outputs ---------------------------------------------------------------------------------
    def _dev(self, name: str) -> Tuple[int, int]:
        p = C.c_void_p()
        n = C.c_uint64()
        _check(getattr(self.ctx.lib, f"skinrt_get_{name}")(self.handle, C.byref(p), C.byref(n)))
        return (p.value or 0), int(n.value)

    def device_pointer(self, name: str) -> Tuple[int, int]:
        """(device address, element count) of 'positions' | 'normals' | 'tangents' | 'aabbs' |
        'palette' | 'world' | 'keyframe_index'."""
        return self._dev(name)

    def _fetch(self, name, dtype, shape, first=0, count=None):
        ptr, n = self._dev(name)
        per = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        count = shape[0] - first if count is None else count
        item = np.dtype(dtype).itemsize
        out = self.ctx.download(ptr + first * per * item, count * per, dtype)
        return out.reshape((count,) + tuple(shape[1:]))

    def positions(self, first=0, count=None):
        return self._fetch("positions", np.float32, (self.instances, self.rig.data.vertex_count, 3), first, count)

    def normals(self, first=0, count=None):
        return self._fetch("normals", np.float32, (self.instances, self.rig.data.vertex_count, 3), first, count)

    def tangents(self, first=0, count=None):
        return self._fetch("tangents", np.float32, (self.instances, self.rig.data.vertex_count, 3), first, count)

    def aabbs(self, first=0, count=None):
        return self._fetch("aabbs", np.float32, (self.instances, 2, 3), first, count)

    def palette(self, first=0, count=None):
        """[count][J][24]: 3 rows x 4 of the joint matrix, then 3 rows x 4 of the normal matrix (4th of each row 0).  The
        device buffer (skinrt_get_palette) keeps the nine normal entries packed; they are spread to padded rows here."""
        dev = self._fetch("palette", np.float32, (self.instances, self.rig.data.palette_count, 24), first, count)
        out = np.zeros_like(dev)
        out[..., :12] = dev[..., :12]
        for r in range(3):
            out[..., 12 + 4 * r:15 + 4 * r] = dev[..., 12 + 3 * r:15 + 3 * r]
        assert not dev[..., 21:].any(), "palette padding must be zero"
        return out

    def world(self, first=0, count=None):
        return self._fetch("world", np.float32, (self.instances, self.rig.data.node_count, 12), first, count)

    def keyframe_index(self, first=0, count=None):
        return self._fetch("keyframe_index", np.uint32, (self.instances, self.rig.data.max_channels), first, count)

    def skin_uniform(self, instance: int) -> np.ndarray:
        """128 x SkinMatrix (28 f32 each) exactly as the reference uploads it (uniform/skin.rs:41-50)."""
        out = np.empty((128, 28), dtype=np.float32)
        _check(self.ctx.lib.skinrt_export_skin_uniform(self.handle, instance, out.ctypes.data_as(C.c_void_p)))
        return out

    def interleaved(self, instance: int, layout: int, attr=None) -> np.ndarray:
        stride = 76 if layout == abi.LAYOUT_COLOR_SKIN else 68
        out = np.empty(self.rig.data.vertex_count * stride, dtype=np.uint8)
        ap = None
        if attr is not None:
            attr = np.ascontiguousarray(attr, dtype=np.float32)
            ap = attr.ctypes.data_as(C.c_void_p)
        _check(self.ctx.lib.skinrt_export_interleaved(self.handle, instance, layout, ap, out.ctypes.data_as(C.c_void_p)))
        return out

    def skin_uniform_of(self, skin: int, instance: int) -> np.ndarray:
        """128 x SkinMatrix of one skin of a rig with several."""
        out = np.empty((128, 28), dtype=np.float32)
        _check(self.ctx.lib.skinrt_export_skin_uniform_of(self.handle, skin, instance, out.ctypes.data_as(C.c_void_p)))
        return out

    def interleaved_section(self, section: int, instance: int, layout: int, attr=None) -> np.ndarray:
        stride = 76 if layout == abi.LAYOUT_COLOR_SKIN else 68
        out = np.empty(self.rig.section_layout(section).vertex_count * stride, dtype=np.uint8)
        ap = None
        if attr is not None:
            attr = np.ascontiguousarray(attr, dtype=np.float32)
            ap = attr.ctypes.data_as(C.c_void_p)
        _check(self.ctx.lib.skinrt_export_interleaved_section(self.handle, section, instance, layout, ap, out.ctypes.data_as(C.c_void_p)))
        return out

    def instance_uniform(self, instance: int, node: int) -> np.ndarray:
        """InstanceUniform (124 bytes = 31 f32: mat4, 3 padded normal columns, 12 bytes of padding) of one node."""
        out = np.empty(31, dtype=np.float32)
        _check(self.ctx.lib.skinrt_export_instance_uniform(self.handle, instance, node, out.ctypes.data_as(C.c_void_p)))
        return out

    def skin_uniform_of_device(self, skin: int, first: int, count: int, dev_dst: int):
        _check(self.ctx.lib.skinrt_export_skin_uniform_of_device(self.handle, skin, first, count, C.c_void_p(dev_dst)))

    def skin_storage_device(self, skin: int, first: int, count: int, slots: int, dev_dst: int):
        """`slots` SkinMatrix records per instance (storage-buffer palette: no 128-joint limit)."""
        _check(self.ctx.lib.skinrt_export_skin_storage_device(self.handle, skin, first, count, slots, C.c_void_p(dev_dst)))

    def interleaved_section_device(self, section: int, first: int, count: int, layout: int, dev_attr: int, dev_dst: int):
        _check(self.ctx.lib.skinrt_export_interleaved_section_device(self.handle, section, first, count, layout,
                                                                     C.c_void_p(dev_attr) if dev_attr else None, C.c_void_p(dev_dst)))

    def instance_uniform_device(self, first: int, count: int, dev_nodes: int, node_count: int, stride: int, dev_dst: int):
        _check(self.ctx.lib.skinrt_export_instance_uniform_device(self.handle, first, count, C.c_void_p(dev_nodes), node_count, stride,
                                                                  C.c_void_p(dev_dst)))

    def skin_uniform_device(self, first: int, count: int, dev_dst: int):
        """[count][128] SkinMatrix written by one kernel into device memory at `dev_dst` (asynchronous)."""
        _check(self.ctx.lib.skinrt_export_skin_uniform_device(self.handle, first, count, C.c_void_p(dev_dst)))

    def interleaved_device(self, first: int, count: int, layout: int, dev_attr: int, dev_dst: int):
        """[count][V] ColorSkinVertex / TextureSkinVertex records written by one kernel into device memory (asynchronous)."""
        _check(self.ctx.lib.skinrt_export_interleaved_device(self.handle, first, count, layout,
                                                             C.c_void_p(dev_attr) if dev_attr else None, C.c_void_p(dev_dst)))

    def close(self):
        if self.handle:
            self.ctx.lib.skinrt_destroy_crowd(self.handle)
            self.handle = C.c_void_p()


class Node:
    """One crowd sharded over several devices of this process (skinrt_node_*): whole-crowd inputs in, per-shard
    outputs through `shard(g)`, which lends a Crowd view of the shard's handles (owned by the node)."""

    def __init__(self, devices, data: RigData, instances: int, **cfg):
        self.lib = load_library()
        self.data, self.instances = data, int(instances)
        self.config = make_config(device=0, **cfg)
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        desc = data.to_desc()
        self.handle = C.c_void_p()
        _check(self.lib.skinrt_node_create(C.byref(self.config), devs, len(devices), C.byref(desc), self.instances, C.byref(self.handle)))

    @property
    def shard_count(self) -> int:
        n = C.c_uint32()
        _check(self.lib.skinrt_node_shard_count(self.handle, C.byref(n)))
        return int(n.value)

    def shard(self, g: int):
        """(crowd view or None for an empty shard, first instance, count)."""
        ctx_h, crowd_h, first, count = C.c_void_p(), C.c_void_p(), C.c_uint32(), C.c_uint32()
        _check(self.lib.skinrt_node_get_shard(self.handle, g, C.byref(ctx_h), C.byref(crowd_h), C.byref(first), C.byref(count)))
        if not crowd_h.value:
            return None, int(first.value), int(count.value)
        ctx = Context.__new__(Context)
        ctx.lib, ctx.config, ctx.handle = self.lib, self.config, ctx_h
        rig = Rig.__new__(Rig)
        rig.ctx, rig.data, rig.handle = ctx, self.data, C.c_void_p()
        crowd = Crowd.__new__(Crowd)
        crowd.ctx, crowd.rig, crowd.instances, crowd.handle = ctx, rig, int(count.value), crowd_h
        return crowd, int(first.value), int(count.value)

    def set_clocks(self, tau, clip_id=None):
        tau = np.ascontiguousarray(tau, dtype=np.float32)
        assert tau.shape == (self.instances,)
        cp = None
        if clip_id is not None:
            clip_id = np.ascontiguousarray(clip_id, dtype=np.uint32)
            assert clip_id.shape == (self.instances,)
            cp = clip_id.ctypes.data_as(C.c_void_p)
        _check(self.lib.skinrt_node_set_clocks(self.handle, tau.ctypes.data_as(C.c_void_p), cp))

    def set_morph_weights(self, weights):
        wp = None
        if weights is not None:
            weights = np.ascontiguousarray(weights, dtype=np.float32)
            assert weights.shape == (self.instances, self.data.target_count)
            wp = weights.ctypes.data_as(C.c_void_p)
        _check(self.lib.skinrt_node_set_morph_weights(self.handle, wp))

    def set_root_transforms(self, matrices):
        mp = None
        if matrices is not None:
            matrices = np.ascontiguousarray(matrices, dtype=np.float32)
            assert matrices.shape == (self.instances, 16)
            mp = matrices.ctypes.data_as(C.c_void_p)
        _check(self.lib.skinrt_node_set_root_transforms(self.handle, mp))

    def evaluate(self, stages: int = abi.STAGE_ALL, sync: bool = False):
        _check(self.lib.skinrt_node_evaluate(self.handle, stages))
        if sync:
            self.synchronize()

    def synchronize(self):
        _check(self.lib.skinrt_node_synchronize(self.handle))

    def aabbs(self) -> np.ndarray:
        out = np.empty((self.instances, 2, 3), np.float32)
        _check(self.lib.skinrt_node_download_aabbs(self.handle, out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if self.handle:
            self.lib.skinrt_node_destroy(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
