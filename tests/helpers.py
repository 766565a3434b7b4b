"""Shared helpers for the parity tests: run the HIP path through the C ABI, run the oracle on
the same inputs, compare at the north-star bar (bit-exact integers; 1e-5 rel / 1e-6 abs fp32)."""
import numpy as np

from renderer_amd import abi, api, synth

RTOL, ATOL = 1e-5, 1e-6


def run_gpu(rig, crowd, stages=abi.STAGE_ALL, want=("world", "palette", "key_index", "position", "normal", "tangent", "aabb"), **cfg):
    ctx = api.Context(**cfg)
    try:
        r = api.Rig(ctx, rig)
        c = api.Crowd(ctx, r, crowd.tau.shape[0])
        c.set_clocks(crowd.tau, crowd.clip_id)
        if crowd.morph_weights is not None:
            c.set_morph_weights(crowd.morph_weights)
        if crowd.root is not None:
            c.set_root_transforms(crowd.root)
        c.evaluate(stages, sync=True)
        out = {}
        get = {"world": c.world, "palette": c.palette, "key_index": c.keyframe_index, "position": c.positions,
               "normal": c.normals, "tangent": c.tangents, "aabb": c.aabbs}
        for k in want:
            out[k] = get[k]()
        out["info"] = r.info
        out["ms"] = c.stage_ms()
        c.close(); r.close()
        return out
    finally:
        ctx.close()


def run_oracle(oracle, rig, crowd, cubic_quirk=1, normalize_outputs=0, want=("world", "palette", "key_index", "position", "normal", "tangent", "aabb")):
    res = oracle.evaluate(rig, crowd.tau, crowd.clip_id, crowd.morph_weights, crowd.root,
                          cubic_quirk=cubic_quirk, normalize_outputs=normalize_outputs, want=want)
    out = {}
    if "world" in want: out["world"] = res.world_rows()
    if "palette" in want: out["palette"] = res.palette_rows()
    if "key_index" in want: out["key_index"] = res.key_index
    for k in ("position", "normal", "tangent", "aabb"):
        if k in want: out[k] = getattr(res, k)
    return out


def assert_close(got, ref, what, rtol=RTOL, atol=ATOL):
    got = np.asarray(got, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    err = np.abs(got - ref)
    tol = atol + rtol * np.abs(ref)
    bad = ~(err <= tol)
    if bad.any():
        idx = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.size} outside 1e-5 rel / 1e-6 abs; worst at {idx}: "
                             f"got {got[idx]!r} ref {ref[idx]!r} err {err[idx]:.3e}")


def assert_equal(got, ref, what):
    got = np.asarray(got); ref = np.asarray(ref)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    neq = ~(got == ref)
    if neq.any():
        idx = np.unravel_index(np.argmax(neq), neq.shape)
        raise AssertionError(f"{what}: {int(neq.sum())}/{neq.size} differ; first at {idx}: got {got[idx]!r} ref {ref[idx]!r}")


import ctypes as C


class DeviceBuffers:
    """Caller-owned device memory (plain HIP through ctypes: the caller's business, not the library's)."""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.ptrs = []

    def _ok(self, rc):
        assert rc == 0, f"HIP error {rc}"

    def upload(self, a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        self._ok(self.hip.hipMalloc(C.byref(p), C.c_size_t(max(a.nbytes, 4))))
        self._ok(self.hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), C.c_int(1)))
        self.ptrs.append(p)
        return p.value

    def empty(self, nbytes, fill=0xAB):
        p = C.c_void_p()
        self._ok(self.hip.hipMalloc(C.byref(p), C.c_size_t(max(nbytes, 4))))
        self._ok(self.hip.hipMemset(p, C.c_int(fill), C.c_size_t(max(nbytes, 4))))
        self._ok(self.hip.hipDeviceSynchronize())  # a memset of device memory returns before it has run
        self.ptrs.append(p)
        return p.value

    def download(self, ptr, shape, dtype):
        out = np.empty(shape, dtype)
        self._ok(self.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(out.nbytes), C.c_int(2)))
        return out

    def close(self):
        for p in self.ptrs:
            self.hip.hipFree(p)
        self.ptrs = []
