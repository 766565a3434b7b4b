// kernels_geom.hip — load-time geometry next to the path: the reference's
// calculate_tangent (renderer-asset/src/tangent.rs:15-58) on the device.
//
// The reference walks the triangles in order and ADDS each unit face normal to its three vertices,
// then normalises per vertex.  Floating-point addition is not associative, so a scatter-add with
// atomics would give run-to-run different bits.  Here the sum is turned into a gather with the
// reference's order: corners (3 per triangle, numbered 3f + k) are bucketed per vertex
// (count -> exclusive scan -> fill), each vertex sorts its few corner ids ascending — which is the
// order in which the sequential loop reached it — and adds the face normals in that order.  The
// result is bit-equal to the sequential loop.  Compiled with -ffp-contract=off (glam's scalar Vec3
// does separate multiplies and adds).
//
// The same sum PER FRAME on a crowd's deformed positions (the second half of that row: "per-frame normal recompute
// after morphing"): a topology object keeps the mesh's corner buckets, sorted once; k_crowd_normals then evaluates
// calculate_tangent for every (instance, vertex) from that instance's skinned positions, face normals recomputed where
// they are used (no [instances][faces] scratch), in the same order with the same operations.
#include <hip/hip_runtime.h>

#include <mutex>

#include "device_types.hpp"

namespace skinrt {

namespace {

__device__ __forceinline__ uint32_t corner_vertex(const uint32_t* __restrict__ indices, uint32_t c) { return indices ? indices[c] : c; }

// A face with a corner index >= vertices touches nothing (the host entry point rejects such input before
// any launch — the reference panics on it; the device-pointer entry point cannot look at the indices,
// so the kernels must be safe for them).
__device__ __forceinline__ bool face_valid(const uint32_t* __restrict__ indices, uint32_t f, uint32_t vertices) {
    return corner_vertex(indices, 3 * f) < vertices && corner_vertex(indices, 3 * f + 1) < vertices && corner_vertex(indices, 3 * f + 2) < vertices;
}

// unit normal of face f: normalize((p2 - p1) x (p0 - p1)), glam operation order (tangent.rs:32-41)
__device__ __forceinline__ void face_unit_normal(const float* __restrict__ pos, const uint32_t* __restrict__ indices, uint32_t f, float (&n)[3]) {
    const uint32_t i0 = corner_vertex(indices, 3 * f), i1 = corner_vertex(indices, 3 * f + 1), i2 = corner_vertex(indices, 3 * f + 2);
    const float* p0 = pos + (size_t)i0 * 3; const float* p1 = pos + (size_t)i1 * 3; const float* p2 = pos + (size_t)i2 * 3;
    const float a[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    const float b[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
    const float c[3] = {a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]};
    const float r = 1.0f / sqrtf((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]);
    n[0] = c[0] * r; n[1] = c[1] * r; n[2] = c[2] * r;
}
__global__ __launch_bounds__(256) void k_face_normals(const float* __restrict__ pos, const uint32_t* __restrict__ indices,
                                                      uint32_t faces, uint32_t vertices, float* __restrict__ fn) {
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    if (f >= faces || !face_valid(indices, f, vertices)) return;
    float n[3];
    face_unit_normal(pos, indices, f, n);
    fn[(size_t)f * 3 + 0] = n[0]; fn[(size_t)f * 3 + 1] = n[1]; fn[(size_t)f * 3 + 2] = n[2];
}

__global__ __launch_bounds__(256) void k_corner_count(const uint32_t* __restrict__ indices, uint32_t corners, uint32_t vertices,
                                                      uint32_t* __restrict__ cnt) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c < corners && face_valid(indices, c / 3, vertices)) atomicAdd(&cnt[corner_vertex(indices, c)], 1u);
}

__global__ __launch_bounds__(256) void k_corner_fill(const uint32_t* __restrict__ indices, uint32_t corners, uint32_t vertices,
                                                     const uint32_t* __restrict__ start, uint32_t* __restrict__ cursor, uint32_t* __restrict__ adj) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= corners || !face_valid(indices, c / 3, vertices)) return;
    const uint32_t v = corner_vertex(indices, c);
    adj[start[v] + atomicAdd(&cursor[v], 1u)] = c;
}

// one thread per vertex: its corners ascending, face normals added in that order, then v * (1 / sqrt(dot(v, v)))
// (tangent.rs:52-56).  A vertex has a handful of corners: insertion sort in place.  Poles and fans have hundreds to
// thousands: above kInsertionMax an in-place heapsort, O(n log n) — the quadratic sort was a hang in waiting for a
// caller's mesh (the device-pointer entry point cannot look at the indices first).  Above
// SKINRT_TANGENT_MAX_VALENCE corners on ONE vertex (no mesh has that: 16,384 triangles round a point) the vertex is not
// summed at all and gets NaN, the value the reference gives a vertex it cannot normalise.
constexpr uint32_t kInsertionMax = 32;
__device__ void sift_down(uint32_t* a, uint32_t root, uint32_t n) {
    const uint32_t key = a[root];
    for (;;) {
        uint32_t child = 2 * root + 1;
        if (child >= n) break;
        uint32_t cv = a[child];
        if (child + 1 < n) { const uint32_t c2 = a[child + 1]; if (c2 > cv) { cv = c2; ++child; } }
        if (cv <= key) break;
        a[root] = cv;
        root = child;
    }
    a[root] = key;
}
// corner ids adj[b .. e) ascending, in place
__device__ __forceinline__ void sort_corners(uint32_t* adj, uint32_t b, uint32_t e) {
    if (e - b <= kInsertionMax) {
        for (uint32_t i = b + 1; i < e; ++i) {
            const uint32_t key = adj[i];
            uint32_t j = i;
            for (; j > b && adj[j - 1] > key; --j) adj[j] = adj[j - 1];
            adj[j] = key;
        }
    } else {
        uint32_t* a = adj + b;
        const uint32_t n = e - b;
        for (uint32_t i = n / 2; i-- > 0;) sift_down(a, i, n);
        for (uint32_t m = n - 1; m > 0; --m) {
            const uint32_t top = a[0];
            a[0] = a[m];
            a[m] = top;
            sift_down(a, 0, m);
        }
    }
}
__global__ __launch_bounds__(256) void k_vertex_sum(uint32_t vertices, const uint32_t* __restrict__ start, uint32_t* __restrict__ adj,
                                                    const float* __restrict__ fn, float* __restrict__ out) {
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (v >= vertices) return;
    const uint32_t b = start[v], e = start[v + 1];
    if (e - b > SKINRT_TANGENT_MAX_VALENCE) {
        const float nan = __builtin_nanf("");
        out[(size_t)v * 3 + 0] = nan; out[(size_t)v * 3 + 1] = nan; out[(size_t)v * 3 + 2] = nan;
        return;
    }
    sort_corners(adj, b, e);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    for (uint32_t i = b; i < e; ++i) {
        const float* n = fn + (size_t)(adj[i] / 3) * 3;
        s0 += n[0]; s1 += n[1]; s2 += n[2];
    }
    const float r = 1.0f / sqrtf((s0 * s0 + s1 * s1) + s2 * s2);
    out[(size_t)v * 3 + 0] = s0 * r; out[(size_t)v * 3 + 1] = s1 * r; out[(size_t)v * 3 + 2] = s2 * r;
}

// topology: every vertex's corner list sorted once (k_vertex_sum sorts on every call because its work area is per call)
__global__ __launch_bounds__(256) void k_sort_corners(uint32_t vertices, const uint32_t* __restrict__ start, uint32_t* __restrict__ adj) {
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (v >= vertices) return;
    const uint32_t b = start[v], e = start[v + 1];
    if (e - b <= SKINRT_TANGENT_MAX_VALENCE) sort_corners(adj, b, e);
}

// calculate_tangent per (instance, vertex) of a crowd: thread = vertex v of NI consecutive instances (first + blockIdx.y * NI ...).
// pos_all / out_all: [instances][vertices][3].  The face normals a vertex needs are recomputed from each instance's
// positions (each is computed once per corner that uses it: three times per triangle, the same bits every time) — an
// instance's positions are L2-resident for the block row that walks it, and no [instances][faces] table has to exist.
// A corner's face and its three vertex indices are fetched once for the NI instances (they are the mesh's, not the
// instance's): a third of the gathers of the one-instance form.  Per instance the additions keep the sequential order.
constexpr uint32_t kNormalsInstances = 8;
__global__ __launch_bounds__(256) void k_crowd_normals(const float* __restrict__ pos_all, const uint32_t* __restrict__ indices,
                                                       const uint32_t* __restrict__ start, const uint32_t* __restrict__ adj,
                                                       uint32_t vertices, uint32_t first, uint32_t count, float* __restrict__ out_all) {
    constexpr uint32_t NI = kNormalsInstances;
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (v >= vertices) return;
    const uint32_t k0 = blockIdx.y * NI;                       // first of this thread's instances, relative to `first`
    const uint32_t ni = count - k0 < NI ? count - k0 : NI;      // >= 1 by the grid's size
    const size_t stride = (size_t)vertices * 3;
    const float* pos0 = pos_all + (size_t)(first + k0) * stride;
    float* out0 = out_all + (size_t)(first + k0) * stride + (size_t)v * 3;
    const uint32_t b = start[v], e = start[v + 1];
    if (e - b > SKINRT_TANGENT_MAX_VALENCE) {
        const float nan = __builtin_nanf("");
        for (uint32_t k = 0; k < ni; ++k) { float* o = out0 + k * stride; o[0] = nan; o[1] = nan; o[2] = nan; }
        return;
    }
    float s[NI][3];
#pragma unroll
    for (uint32_t k = 0; k < NI; ++k) { s[k][0] = 0.0f; s[k][1] = 0.0f; s[k][2] = 0.0f; }
    for (uint32_t i = b; i < e; ++i) {
        const uint32_t f = adj[i] / 3;
        const uint32_t i0 = corner_vertex(indices, 3 * f), i1 = corner_vertex(indices, 3 * f + 1), i2 = corner_vertex(indices, 3 * f + 2);
#pragma unroll
        for (uint32_t k = 0; k < NI; ++k) {
            if (k >= ni) break;  // the crowd's last group may be short
            const float* pos = pos0 + k * stride;
            const float* p0 = pos + (size_t)i0 * 3; const float* p1 = pos + (size_t)i1 * 3; const float* p2 = pos + (size_t)i2 * 3;
            // normalize((p2 - p1) x (p0 - p1)): face_unit_normal's operations, on indices already fetched
            const float a[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
            const float d[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
            const float c[3] = {a[1] * d[2] - d[1] * a[2], a[2] * d[0] - d[2] * a[0], a[0] * d[1] - d[0] * a[1]};
            const float r = 1.0f / sqrtf((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]);
            s[k][0] += c[0] * r; s[k][1] += c[1] * r; s[k][2] += c[2] * r;
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < NI; ++k) {
        if (k >= ni) break;
        const float r = 1.0f / sqrtf((s[k][0] * s[k][0] + s[k][1] * s[k][1]) + s[k][2] * s[k][2]);
        float* o = out0 + k * stride;
        o[0] = s[k][0] * r; o[1] = s[k][1] * r; o[2] = s[k][2] * r;
    }
}

// Exclusive prefix sum of n counters (integers: exact in any order).  Three small kernels instead of a
// library scan: no temporary-storage query, no alignment rules, nothing that could allocate.
constexpr uint32_t kScanChunk = 1024;  // elements per block of k_scan_chunks
__global__ __launch_bounds__(256) void k_scan_chunks(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out,
                                                     uint32_t* __restrict__ chunk_sum) {
    __shared__ uint32_t part[256];
    const uint32_t t = threadIdx.x, base = blockIdx.x * kScanChunk + t * 4;
    uint32_t v[4], run = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = base + k < n ? in[base + k] : 0u; run += v[k]; }
    part[t] = run;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {  // inclusive scan of the 256 thread totals
        const uint32_t x = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    uint32_t excl = part[t] - run;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (base + k < n) out[base + k] = excl; excl += v[k]; }
    if (t == 255) chunk_sum[blockIdx.x] = part[255];
}
// one block: chunk_sum[] -> its own exclusive scan, in place
__global__ __launch_bounds__(256) void k_scan_sums(uint32_t* __restrict__ chunk_sum, uint32_t chunks) {
    __shared__ uint32_t part[256];
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < chunks; b0 += 256) {
        const uint32_t t = threadIdx.x, i = b0 + t;
        const uint32_t v = i < chunks ? chunk_sum[i] : 0u;
        part[t] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 256; d <<= 1) {
            const uint32_t x = t >= d ? part[t - d] : 0u;
            __syncthreads();
            part[t] += x;
            __syncthreads();
        }
        if (i < chunks) chunk_sum[i] = carry + part[t] - v;
        carry += part[255];
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_scan_add(uint32_t* __restrict__ out, uint32_t n, const uint32_t* __restrict__ chunk_sum) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] += chunk_sum[i / kScanChunk];
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Work area of launch_tangents for the largest input skinrt_calculate_tangent_noalloc accepts.  One copy per
// GPU the library runs on, for the life of the process; every call on a GPU shares it.
constexpr size_t kStaticTangentBytes = (size_t)16 << 20;
static __device__ __align__(256) unsigned char scratch[kStaticTangentBytes];

std::mutex g_scratch_mutex;
constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices];       // device address of `scratch` on each device (nullptr: not resolved)
hipError_t g_scratch_err[kMaxDevices];  // why not
hipEvent_t g_scratch_done[kMaxDevices]; // recorded behind the last user of the device's work area (see tangent_static_use)

}  // namespace

// Resolve the device address of the static work area on `device` (which must be the current device).
// `scratch` as written in host code is a host-side handle, not a device address: every use below goes
// through the pointer hipGetSymbolAddress returns, and each device has its own.
hipError_t tangent_static_setup(int device) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return hipSuccess;
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    g_scratch_err[device] = e;
    if (e != hipSuccess) { (void)hipGetLastError(); return e; }
    // one event per device, made here so that the entry point itself creates nothing
    if (hipEventCreateWithFlags(&g_scratch_done[device], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); g_scratch_done[device] = nullptr; }
    g_scratch_ptr[device] = ptr;
    return hipSuccess;
}

// Users of one device's work area, one after the other WITHOUT allocating and without blocking the host: several ctx
// (streams) may live on one GPU — skinrt_node_create allows a device to be listed more than once — and their calls would
// race on the shared tables.  `launch` enqueues the user's kernels on `s`; they are ordered behind the previous user's
// by an event made at set-up.  Under stream capture an event recorded outside the capture cannot be waited for (and
// one recorded inside it belongs to the graph), so a capturing stream is not chained: overlap is then the caller's
// business, as skinrt.h says.
hipError_t tangent_static_use(int device, hipStream_t s, hipError_t (*launch)(void*), void* arg) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    hipEvent_t ev = cap == hipStreamCaptureStatusNone ? g_scratch_done[device] : nullptr;
    hipError_t e = hipSuccess;
    if (ev) e = hipStreamWaitEvent(s, ev, 0);  // (an event never recorded counts as complete)
    if (e == hipSuccess) e = launch(arg);
    if (e == hipSuccess && ev) e = hipEventRecord(ev, s);
    return e;
}

hipError_t tangent_static_area(int device, void** ptr, size_t* bytes) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) return g_scratch_err[device] != hipSuccess ? g_scratch_err[device] : hipErrorNotInitialized;
    *ptr = g_scratch_ptr[device];
    *bytes = kStaticTangentBytes;
    return hipSuccess;
}

// pos: [V][3] device; indices: [corners] device or nullptr (then corner c is vertex c); corners = 3 * faces;
// work: device scratch of tangent_work_bytes(V, corners); out: [V][3] device
// layout of the work area: every table starts on a 256-byte boundary
size_t tangent_work_bytes(uint32_t vertices, uint32_t corners) {
    const size_t n = (size_t)vertices + 1, chunks = (n + kScanChunk - 1) / kScanChunk;
    return 3 * align256(n * 4) + 2 * align256((size_t)corners * 4) + align256(chunks * 4);
}

hipError_t launch_tangents(const float* pos, const uint32_t* indices, uint32_t vertices, uint32_t corners, void* work, size_t work_bytes,
                           float* out, hipStream_t s) {
    if (vertices == 0) return hipSuccess;
    if (tangent_work_bytes(vertices, corners) > work_bytes || (reinterpret_cast<uintptr_t>(work) & 255u)) return hipErrorInvalidValue;
    const uint32_t n = vertices + 1, chunks = (n + kScanChunk - 1) / kScanChunk;
    char* w = static_cast<char*>(work);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(w);              w += align256((size_t)n * 4);        // [V+1] corner counts
    uint32_t* start = reinterpret_cast<uint32_t*>(w);            w += align256((size_t)n * 4);        // [V+1] their exclusive scan
    uint32_t* cursor = reinterpret_cast<uint32_t*>(w);           w += align256((size_t)n * 4);        // [V+1]
    uint32_t* adj = reinterpret_cast<uint32_t*>(w);              w += align256((size_t)corners * 4);  // [corners]
    float* fn = reinterpret_cast<float*>(w);                     w += align256((size_t)corners * 4);  // [faces][3]
    uint32_t* chunk_sum = reinterpret_cast<uint32_t*>(w);                                             // [chunks]
    const uint32_t faces = corners / 3;
    // the work area keeps the last call's contents: everything read before it is written starts at zero
    hipError_t e = hipMemsetAsync(cnt, 0, 3 * align256((size_t)n * 4), s);  // cnt, start, cursor
    if (e != hipSuccess) return e;
    if (corners) {
        hipLaunchKernelGGL(k_face_normals, dim3((faces + 255) / 256), dim3(256), 0, s, pos, indices, faces, vertices, fn);
        hipLaunchKernelGGL(k_corner_count, dim3((corners + 255) / 256), dim3(256), 0, s, indices, corners, vertices, cnt);
    }
    hipLaunchKernelGGL(k_scan_chunks, dim3(chunks), dim3(256), 0, s, cnt, n, start, chunk_sum);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, s, chunk_sum, chunks);
    hipLaunchKernelGGL(k_scan_add, dim3((n + 255) / 256), dim3(256), 0, s, start, n, chunk_sum);
    if (corners) hipLaunchKernelGGL(k_corner_fill, dim3((corners + 255) / 256), dim3(256), 0, s, indices, corners, vertices, start, cursor, adj);
    hipLaunchKernelGGL(k_vertex_sum, dim3((vertices + 255) / 256), dim3(256), 0, s, vertices, start, adj, fn, out);
    return hipGetLastError();
}

// The corner buckets of a mesh, built once: start[V + 1] and adj[corners] (each vertex's corner ids ascending) inside
// `work` (tangent_work_bytes(V, corners) bytes, 256-byte aligned, kept by the caller for as long as the tables are used).
hipError_t launch_topology(const uint32_t* indices, uint32_t vertices, uint32_t corners, void* work, size_t work_bytes,
                           const uint32_t** out_start, const uint32_t** out_adj, hipStream_t s) {
    if (vertices == 0) return hipErrorInvalidValue;
    if (tangent_work_bytes(vertices, corners) > work_bytes || (reinterpret_cast<uintptr_t>(work) & 255u)) return hipErrorInvalidValue;
    const uint32_t n = vertices + 1, chunks = (n + kScanChunk - 1) / kScanChunk;
    char* w = static_cast<char*>(work);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(w);              w += align256((size_t)n * 4);
    uint32_t* start = reinterpret_cast<uint32_t*>(w);            w += align256((size_t)n * 4);
    uint32_t* cursor = reinterpret_cast<uint32_t*>(w);           w += align256((size_t)n * 4);
    uint32_t* adj = reinterpret_cast<uint32_t*>(w);              w += align256((size_t)corners * 4);
    w += align256((size_t)corners * 4);  // (the face-normal table of launch_tangents: unused here)
    uint32_t* chunk_sum = reinterpret_cast<uint32_t*>(w);
    hipError_t e = hipMemsetAsync(cnt, 0, 3 * align256((size_t)n * 4), s);
    if (e != hipSuccess) return e;
    if (corners) hipLaunchKernelGGL(k_corner_count, dim3((corners + 255) / 256), dim3(256), 0, s, indices, corners, vertices, cnt);
    hipLaunchKernelGGL(k_scan_chunks, dim3(chunks), dim3(256), 0, s, cnt, n, start, chunk_sum);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, s, chunk_sum, chunks);
    hipLaunchKernelGGL(k_scan_add, dim3((n + 255) / 256), dim3(256), 0, s, start, n, chunk_sum);
    if (corners) hipLaunchKernelGGL(k_corner_fill, dim3((corners + 255) / 256), dim3(256), 0, s, indices, corners, vertices, start, cursor, adj);
    hipLaunchKernelGGL(k_sort_corners, dim3((vertices + 255) / 256), dim3(256), 0, s, vertices, start, adj);
    *out_start = start;
    *out_adj = adj;
    return hipGetLastError();
}

// calculate_tangent of instances [first, first + count) of a crowd from their positions; pos_all / out_all [I][V][3]
hipError_t launch_crowd_normals(const float* pos_all, const uint32_t* indices, const uint32_t* start, const uint32_t* adj,
                                uint32_t vertices, uint32_t first, uint32_t count, float* out_all, hipStream_t s) {
    if (vertices == 0) return hipSuccess;
    for (uint32_t done = 0; done < count;) {  // gridDim.y holds at most 65,535 groups of kNormalsInstances instances
        const uint32_t cap = 65535u * kNormalsInstances;
        const uint32_t n = count - done < cap ? count - done : cap;
        hipLaunchKernelGGL(k_crowd_normals, dim3((vertices + 255) / 256, (n + kNormalsInstances - 1) / kNormalsInstances), dim3(256), 0, s, pos_all,
                           indices, start, adj, vertices, first + done, n, out_all);
        done += n;
    }
    return hipGetLastError();
}

}  // namespace skinrt
