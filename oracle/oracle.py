"""ctypes binding of the CPU oracle (oracle/skin_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by
renderer_amd/.  Parity unpinned (see skin_oracle.c header)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

from renderer_amd.abi import RigData, RigDesc

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force: bool = False) -> None:
    """Compile liboracle_f32.so / liboracle_f64.so with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "skin_oracle.c")
    libs = [os.path.join(_HERE, n) for n in ("liboracle_f32.so", "liboracle_f64.so")]
    stale = force or any((not os.path.exists(l)) or os.path.getmtime(l) < os.path.getmtime(src) for l in libs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, stdout=subprocess.DEVNULL)


class CrowdIn(C.Structure):
    _fields_ = [
        ("instances", C.c_uint32),
        ("tau", C.POINTER(C.c_float)),
        ("clip_id", C.POINTER(C.c_uint32)),
        ("morph_weights", C.POINTER(C.c_float)),
        ("root", C.POINTER(C.c_float)),
        ("cubic_quirk", C.c_uint32),
        ("normalize_outputs", C.c_uint32),
    ]


class Out(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("world", "palette", "key_index", "position", "normal", "tangent", "aabb")]


@dataclass
class OracleResult:
    world: Optional[np.ndarray] = None      # [I][N][16] column-major Mat4
    palette: Optional[np.ndarray] = None    # [I][J][28] SkinMatrix layout
    key_index: Optional[np.ndarray] = None  # [I][Cmax] u32
    position: Optional[np.ndarray] = None   # [I][V][3]
    normal: Optional[np.ndarray] = None
    tangent: Optional[np.ndarray] = None
    aabb: Optional[np.ndarray] = None       # [I][2][3]

    # conversions to the library's device layouts ----------------------------------------
    def world_rows(self) -> np.ndarray:
        """[I][N][12]: first three rows of each Mat4, row-major (skinrt_get_world layout)."""
        w = self.world.reshape(self.world.shape[0], -1, 4, 4)  # [..][col][row]
        return np.ascontiguousarray(w.transpose(0, 1, 3, 2)[:, :, :3, :]).reshape(w.shape[0], -1, 12)

    def palette_rows(self) -> np.ndarray:
        """[I][J][24]: 3 rows x 4 of the joint matrix then 3 rows x 4 of the normal matrix
        (4th = 0) — skinrt_get_palette layout."""
        p = self.palette
        i, j = p.shape[0], p.shape[1]
        m = p[:, :, :16].reshape(i, j, 4, 4).transpose(0, 1, 3, 2)[:, :, :3, :]      # rows of mat4
        nm = p[:, :, 16:].reshape(i, j, 3, 4)[:, :, :, :3].transpose(0, 1, 3, 2)     # [row][col]
        nm4 = np.zeros((i, j, 3, 4), dtype=p.dtype)
        nm4[:, :, :, :3] = nm
        return np.concatenate([m.reshape(i, j, 12), nm4.reshape(i, j, 12)], axis=2)


class Oracle:
    def __init__(self, precision: str = "f32"):
        build()
        self.precision = precision
        self.lib = C.CDLL(os.path.join(_HERE, f"liboracle_{precision}.so"))
        self.real = np.float32 if precision == "f32" else np.float64
        self.creal = C.c_float if precision == "f32" else C.c_double
        assert self.lib.orc_real_size() == np.dtype(self.real).itemsize
        self.lib.orc_evaluate.restype = C.c_int
        self.lib.orc_evaluate.argtypes = [C.POINTER(RigDesc), C.POINTER(CrowdIn), C.POINTER(Out), C.c_uint32, C.c_uint32]
        self.lib.orc_find_key.restype = C.c_uint32
        self.lib.orc_evaluate_sampled.restype = C.c_int
        self.lib.orc_evaluate_sampled.argtypes = [C.POINTER(RigDesc), C.POINTER(CrowdIn), C.POINTER(Out), C.c_uint32, C.c_uint32,
                                                  C.c_uint32, C.c_uint32]
        self.lib.orc_palette_count.restype = C.c_uint32
        self.lib.orc_palette_count.argtypes = [C.POINTER(RigDesc)]

    # --- unit functions (known-answer tests) ----------------------------------------------
    def _vec(self, a):
        return np.ascontiguousarray(a, dtype=self.real)

    def _p(self, a):
        return a.ctypes.data_as(C.POINTER(self.creal))

    def linear(self, a, b, t):
        a, b = self._vec(a), self._vec(b)
        out = np.zeros_like(a)
        self.lib.orc_linear(self._p(a), self._p(b), self.creal(t), C.c_int(a.size), self._p(out))
        return out

    def cubic_spline(self, vk, bk, vk1, ak1, t, td, quirk):
        vk, bk, vk1, ak1 = map(self._vec, (vk, bk, vk1, ak1))
        out = np.zeros_like(vk)
        self.lib.orc_cubic_spline(self._p(vk), self._p(bk), self._p(vk1), self._p(ak1), self.creal(t),
                                  self.creal(td), C.c_int(vk.size), C.c_int(int(quirk)), self._p(out))
        return out

    def find_key(self, times, tau):
        times = np.ascontiguousarray(times, dtype=np.float32)
        u = self.creal(0)
        cl = C.c_int(0)
        k = self.lib.orc_find_key(times.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(times.size),
                                  self.creal(tau), C.byref(u), C.byref(cl))
        return int(k), float(u.value), bool(cl.value)

    def trs_to_mat4(self, t, q, s):
        t, q, s = map(self._vec, (t, q, s))
        out = np.zeros(16, dtype=self.real)
        self.lib.orc_trs_to_mat4(self._p(t), self._p(q), self._p(s), self._p(out))
        return out

    def mat4_mul(self, a, b):
        a, b = self._vec(a).reshape(16), self._vec(b).reshape(16)
        out = np.zeros(16, dtype=self.real)
        self.lib.orc_mat4_mul(self._p(a), self._p(b), self._p(out))
        return out

    def mat4_inverse(self, m):
        m = self._vec(m).reshape(16)
        out = np.zeros(16, dtype=self.real)
        self.lib.orc_mat4_inverse(self._p(m), self._p(out))
        return out

    def normal_transform(self, m):
        m = self._vec(m).reshape(16)
        out = np.zeros(12, dtype=self.real)
        self.lib.orc_normal_transform(self._p(m), self._p(out))
        return out

    def decompose(self, m):
        m = self._vec(m).reshape(16)
        t, q, s = np.zeros(3, self.real), np.zeros(4, self.real), np.zeros(3, self.real)
        self.lib.orc_decompose(self._p(m), self._p(t), self._p(q), self._p(s))
        return t, q, s

    def calculate_tangent(self, mode, positions, indices=None):
        """calculate_tangent (renderer-asset/src/tangent.rs:15-58); [V][3]."""
        p = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        idx = None if indices is None else np.ascontiguousarray(indices, np.uint32).reshape(-1)
        out = np.zeros(p.shape, dtype=self.real)
        rc = self.lib.orc_calculate_tangent(C.c_uint32(mode), p.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(p.shape[0]),
                                            None if idx is None else idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                                            C.c_uint32(0 if idx is None else idx.shape[0]), self._p(out))
        if rc == -1:
            raise NotImplementedError("TriangleStrip: todo!() in the reference (tangent.rs:26)")
        if rc == -2:
            raise IndexError("vertex index out of range (reference: index panic)")
        return out

    def instance_uniform(self, world16):
        """InstanceUniform (uniform/transform.rs:8-14) of one world matrix: 31 values (mat4, 3 padded normal columns, padding)."""
        m = self._vec(world16).reshape(16)
        out = np.zeros(31, dtype=self.real)
        self.lib.orc_instance_uniform(self._p(m), self._p(out))
        return out

    def rigid_vertex(self, iu31, p, n, t, normalize=0):
        iu31, p, n, t = self._vec(iu31).reshape(31), self._vec(p), self._vec(n), self._vec(t)
        op, on, ot = (np.zeros(3, dtype=self.real) for _ in range(3))
        self.lib.orc_rigid_vertex(self._p(iu31), self._p(p), self._p(n), self._p(t), C.c_int(int(normalize)), self._p(op), self._p(on), self._p(ot))
        return op, on, ot

    def evaluate_mt(self, rig: RigData, tau, clip_id=None, morph_weights=None, root=None, *, threads=None, **kw) -> OracleResult:
        """evaluate() with the instances split over host threads (ctypes releases the GIL inside orc_evaluate)."""
        import os as _os
        from concurrent.futures import ThreadPoolExecutor
        n = int(np.asarray(tau).shape[0])
        # a GPU box shows the whole machine's CPUs: its share is in OMP_NUM_THREADS
        threads = threads or min(int(_os.environ.get("OMP_NUM_THREADS", "0")) or (_os.cpu_count() or 1), _os.cpu_count() or 1, 64)
        threads = max(1, min(threads, n))
        desc = kw.pop("desc", None) or rig.to_desc()
        res = self.evaluate(rig, tau, clip_id, morph_weights, root, begin=0, end=0, desc=desc, **kw)  # allocates the outputs
        cuts = [n * k // threads for k in range(threads + 1)]
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda k: self.evaluate(rig, tau, clip_id, morph_weights, root, begin=cuts[k], end=cuts[k + 1], desc=desc,
                                                into=res, **kw), range(threads)))
        return res

    # --- whole path --------------------------------------------------------------------------
    def evaluate(self, rig: RigData, tau, clip_id=None, morph_weights=None, root=None, *,
                 cubic_quirk=1, normalize_outputs=0, want=("world", "palette", "key_index", "position",
                                                           "normal", "tangent", "aabb"),
                 begin=0, end=None, desc=None, into: Optional[OracleResult] = None,
                 vertex_step: int = 1, vertex_phase: int = 0) -> OracleResult:
        """vertex_step > 1: position / normal / tangent hold only the vertices v of instance i with
        v % vertex_step == (vertex_phase + i) % vertex_step, compacted to [I][ceil(V / vertex_step)][3] (row v // vertex_step);
        no bounds (orc_evaluate_sampled)."""
        tau = np.ascontiguousarray(tau, dtype=np.float32)
        n_inst = tau.shape[0]
        end = n_inst if end is None else end
        desc = desc if desc is not None else rig.to_desc()
        cin = CrowdIn()
        cin.instances = n_inst
        cin.tau = tau.ctypes.data_as(C.POINTER(C.c_float))
        keep = [tau]
        if clip_id is not None:
            clip_id = np.ascontiguousarray(clip_id, dtype=np.uint32); keep.append(clip_id)
            cin.clip_id = clip_id.ctypes.data_as(C.POINTER(C.c_uint32))
        if morph_weights is not None:
            morph_weights = np.ascontiguousarray(morph_weights, dtype=np.float32); keep.append(morph_weights)
            cin.morph_weights = morph_weights.ctypes.data_as(C.POINTER(C.c_float))
        if root is not None:
            root = np.ascontiguousarray(root, dtype=np.float32); keep.append(root)
            cin.root = root.ctypes.data_as(C.POINTER(C.c_float))
        cin.cubic_quirk = int(cubic_quirk)
        cin.normalize_outputs = int(normalize_outputs)
        res = into if into is not None else OracleResult()
        vrows = rig.vertex_count if vertex_step <= 1 else (rig.vertex_count + vertex_step - 1) // vertex_step
        assert self.lib.orc_palette_count(C.byref(desc)) == rig.palette_count
        shapes = {
            "world": ((n_inst, rig.node_count, 16), self.real),
            "palette": ((n_inst, rig.palette_count, 28), self.real),
            "key_index": ((n_inst, rig.max_channels), np.uint32),
            "position": ((n_inst, vrows, 3), self.real),
            "normal": ((n_inst, vrows, 3), self.real),
            "tangent": ((n_inst, vrows, 3), self.real),
            "aabb": ((n_inst, 2, 3), self.real),
        }
        out = Out()
        for name in want:
            if getattr(res, name) is None:
                shp, dt = shapes[name]
                setattr(res, name, np.zeros(shp, dtype=dt))
            setattr(out, name, getattr(res, name).ctypes.data)
        if vertex_step > 1:
            assert "aabb" not in want, "sampled evaluation computes no bounds"
            rc = self.lib.orc_evaluate_sampled(C.byref(desc), C.byref(cin), C.byref(out), C.c_uint32(begin), C.c_uint32(end),
                                               C.c_uint32(vertex_step), C.c_uint32(vertex_phase))
        else:
            rc = self.lib.orc_evaluate(C.byref(desc), C.byref(cin), C.byref(out), C.c_uint32(begin), C.c_uint32(end))
        if rc != 0:
            raise RuntimeError(f"orc_evaluate failed: {rc}")
        del keep
        return res
