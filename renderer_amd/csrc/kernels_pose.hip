// kernels_pose.hip — animation sampling, local TRS -> matrix, hierarchy propagation and the
// joint palette for gfx950.
//
// Built with -ffp-contract=off (and the pragma below): every product and sum rounds once, in
// the reference's order, so this stage is bit-equal to the fp32 oracle — which matters because
// deep chains (config C5) and world*inverse_bind cancellation amplify any re-association.
//
// Matrices are affine 3x4, stored as 3 rows of 4 floats (row r = m[r][0..3], m[r][3] =
// translation).  glam's full 4x4 products (node.rs:49-51) reduce to these forms exactly when
// the last row is 0,0,0,1: the dropped terms are x*0 and +0.
//
// Two launch shapes:
//   k_pose_wave   one 64-lane wavefront per instance, node matrices in LDS, nodes processed
//                 level by level (topologically-levelled propagation), palette from LDS.
//   split path    for rigs whose node table does not fit LDS or whose hierarchy is too deep to
//                 level-step: k_sample_local (thread per instance x node), k_propagate_quad
//                 (16 lanes per instance walking the DFS order, the 3x4 product spread over lanes),
//                 k_palette (thread per instance x joint).
#include "device_types.hpp"

#pragma clang fp contract(off)
// the LDS-DMA statements below write m0 and say so in their clobber lists; clang warns that m0 is a reserved register
#pragma clang diagnostic ignored "-Winline-asm"

namespace skinrt {

namespace {

struct Rows {
    float4 r0, r1, r2;
};

// Interpolate::linear — renderer-asset/src/animation.rs:51-53
__device__ __forceinline__ float lerp1(float a, float b, float omt, float t) { return a * omt + b * t; }

// Sample one channel (a1-a4).  Returns the selected key.
__device__ uint32_t sample_channel(const ChanRec ch, const float* __restrict__ times,
                                   const float* __restrict__ values, float tau, uint32_t quirk, float out[4]) {
    const float* tm = times + ch.time_off;
    const float* v = values + ch.value_off;
    const uint32_t n = ch.key_count;
    const uint32_t w = ch.path == SKINRT_PATH_ROTATION ? 4u : 3u;
    const bool cubic = ch.interp == SKINRT_INTERP_CUBICSPLINE;
    uint32_t k;
    bool clamped = true;
    float u = 0.0f;
    if (n == 1 || tau <= tm[0]) k = 0;
    else if (tau >= tm[n - 1]) k = n - 1;
    else {
        uint32_t lo = 0, hi = n - 1;  // times[lo] <= tau < times[hi]
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (tm[mid] <= tau) lo = mid; else hi = mid;
        }
        k = lo;
        clamped = false;
    }
    if (cubic) {
        if (clamped) {
            for (uint32_t i = 0; i < w; ++i) out[i] = v[(k * 3 + 1) * w + i];
            return k;
        }
        const float t0 = tm[k], t1 = tm[k + 1];
        u = (tau - t0) / (t1 - t0);
        const float td = t1 - t0;
        // Interpolate::cubic_spline — animation.rs:55-63 (t2 = t.powi(3) when quirk)
        const float t3 = (u * u) * u;
        const float t2 = quirk ? t3 : u * u;
        const float c1 = 2.0f * t3 - 3.0f * t2 + 1.0f;
        const float c2 = t3 - 2.0f * t2 + u;
        const float c3 = -2.0f * t3 + 3.0f * t2;
        const float c4 = t3 - t2;
        for (uint32_t i = 0; i < w; ++i) {
            const float vk = v[(k * 3 + 1) * w + i];
            const float bk = v[(k * 3 + 2) * w + i];
            const float vk1 = v[((k + 1) * 3 + 1) * w + i];
            const float ak1 = v[((k + 1) * 3 + 0) * w + i];
            const float first = vk * c1;
            const float second = bk * td * c2;
            const float third = vk1 * c3;
            const float forth = ak1 * td * c4;
            out[i] = first + second + third + forth;
        }
        return k;
    }
    if (clamped || ch.interp == SKINRT_INTERP_STEP) {
        for (uint32_t i = 0; i < w; ++i) out[i] = v[k * w + i];
        return k;
    }
    const float t0 = tm[k], t1 = tm[k + 1];
    u = (tau - t0) / (t1 - t0);
    const float omt = 1.0f - u;
    for (uint32_t i = 0; i < w; ++i) out[i] = lerp1(v[k * w + i], v[(k + 1) * w + i], omt, u);
    return k;
}

// impl From<DecomposedTransform> for Mat4 — renderer-asset/src/node.rs:47-53 with glam's
// from_quat; T*R*S reduces to (R columns scaled by s, translation column = t).
__device__ __forceinline__ Rows trs_to_rows(const float trs[10]) {
    const float x = trs[3], y = trs[4], z = trs[5], w = trs[6];
    const float x2 = x + x, y2 = y + y, z2 = z + z;
    const float xx = x * x2, xy = x * y2, xz = x * z2;
    const float yy = y * y2, yz = y * z2, zz = z * z2;
    const float wx = w * x2, wy = w * y2, wz = w * z2;
    const float sx = trs[7], sy = trs[8], sz = trs[9];
    Rows m;
    m.r0 = make_float4((1.0f - (yy + zz)) * sx, (xy - wz) * sy, (xz + wy) * sz, trs[0]);
    m.r1 = make_float4((xy + wz) * sx, (1.0f - (xx + zz)) * sy, (yz - wx) * sz, trs[1]);
    m.r2 = make_float4((xz - wy) * sx, (yz + wx) * sy, (1.0f - (xx + yy)) * sz, trs[2]);
    return m;
}

// glam Mat4 * Mat4 for affine operands: out = a * b,
// element = ((a.c0*b0 + a.c1*b1) + a.c2*b2) [+ a.c3 for the translation column].
__device__ __forceinline__ float4 mul_row(const float4 a, const Rows& b) {
    float4 o;
    o.x = (a.x * b.r0.x + a.y * b.r1.x) + a.z * b.r2.x;
    o.y = (a.x * b.r0.y + a.y * b.r1.y) + a.z * b.r2.y;
    o.z = (a.x * b.r0.z + a.y * b.r1.z) + a.z * b.r2.z;
    o.w = ((a.x * b.r0.w + a.y * b.r1.w) + a.z * b.r2.w) + a.w;
    return o;
}
__device__ __forceinline__ Rows mul_affine(const Rows& a, const Rows& b) {
    Rows o;
    o.r0 = mul_row(a.r0, b);
    o.r1 = mul_row(a.r1, b);
    o.r2 = mul_row(a.r2, b);
    return o;
}

__device__ __forceinline__ Rows load_rows(const float* p) {
    const float4* q = reinterpret_cast<const float4*>(p);
    Rows m;
    m.r0 = q[0]; m.r1 = q[1]; m.r2 = q[2];
    return m;
}
__device__ __forceinline__ void store_rows(float* p, const Rows& m) {
    float4* q = reinterpret_cast<float4*>(p);
    q[0] = m.r0; q[1] = m.r1; q[2] = m.r2;
}

// Local matrix of node n for one instance (a5, a6) + key indices of the channels it sampled.
__device__ Rows node_local(const RigDev& rig, uint32_t clip, bool has_clip, uint32_t n, float tau,
                           uint32_t quirk, uint32_t* __restrict__ keyrow) {
    float trs[10];
    const float* rest = rig.rest_trs + (size_t)n * 10;
#pragma unroll
    for (int e = 0; e < 10; ++e) trs[e] = rest[e];
    bool animated = false;
    if (has_clip) {
        const int32_t* nc = rig.node_chan + ((size_t)clip * rig.N + n) * 3;
#pragma unroll
        for (int path = 0; path < 3; ++path) {
            const int32_t c = nc[path];
            if (c >= 0) {
                animated = true;
                const ChanRec ch = rig.chans[c];
                float v[4];
                const uint32_t k = sample_channel(ch, rig.times, rig.values, tau, quirk, v);
                keyrow[ch.local] = k;
                if (path == SKINRT_PATH_TRANSLATION) { trs[0] = v[0]; trs[1] = v[1]; trs[2] = v[2]; }
                else if (path == SKINRT_PATH_ROTATION) { trs[3] = v[0]; trs[4] = v[1]; trs[5] = v[2]; trs[6] = v[3]; }
                else { trs[7] = v[0]; trs[8] = v[1]; trs[9] = v[2]; }
            }
        }
    }
    // NodeTransform::Matrix stays a matrix unless a channel forces the TRS form (node.rs:55-62, :73-87)
    if (rig.node_matrix[n] && !animated) return load_rows(rig.rest_mat + (size_t)n * 12);
    return trs_to_rows(trs);
}

// a8 + a9: joint = world * inverse_bind; normal = Mat3::from_mat4(joint.inverse().transpose())
// (renderer/src/renderer/uniform/skin.rs:30-37).  glam's cofactor inverse reduces, for an affine
// matrix, to the 2x2 minors below (the dropped factors are *1 and -0), in the same order.
struct Normal9 {
    float n00, n01, n02, n10, n11, n12, n20, n21, n22;  // nRC = row R, column C of Mat3(inverse(m)^T)
};
__device__ __forceinline__ Normal9 normal_rows(const Rows& m) {
    // glam naming: mCR = column C, row R
    const float m00 = m.r0.x, m10 = m.r0.y, m20 = m.r0.z;
    const float m01 = m.r1.x, m11 = m.r1.y, m21 = m.r1.z;
    const float m02 = m.r2.x, m12 = m.r2.y, m22 = m.r2.z;
    const float i00 = m11 * m22 - m12 * m21, i01 = m01 * m22 - m02 * m21, i02 = m01 * m12 - m02 * m11;
    const float i10 = m10 * m22 - m12 * m20, i11 = m00 * m22 - m02 * m20, i12 = m00 * m12 - m02 * m10;
    const float i20 = m10 * m21 - m11 * m20, i21 = m00 * m21 - m01 * m20, i22 = m00 * m11 - m01 * m10;
    const float dot1 = (m00 * i00 + m01 * (-i10)) + m02 * i20;
    const float rcp = 1.0f / dot1;
    return Normal9{i00 * rcp, (-i01) * rcp, i02 * rcp, (-i10) * rcp, i11 * rcp, (-i12) * rcp, i20 * rcp, (-i21) * rcp, i22 * rcp};
}

__device__ __forceinline__ void palette_entry(const Rows& world, const Rows& ibm, float* __restrict__ out24) {
    const Rows m = mul_affine(world, ibm);
    const Normal9 n = normal_rows(m);
    float4* o = reinterpret_cast<float4*>(out24);
    o[0] = m.r0; o[1] = m.r1; o[2] = m.r2;
    // normal rows packed: 9 floats, so that the skin kernels fetch them as 2 x 16 bytes + 4 bytes per influence
    o[3] = make_float4(n.n00, n.n01, n.n02, n.n10);
    o[4] = make_float4(n.n11, n.n12, n.n20, n.n21);
    o[5] = make_float4(n.n22, 0.0f, 0.0f, 0.0f);
}

__device__ __forceinline__ void clear_unused_keys(const RigDev& rig, uint32_t clip, bool has_clip,
                                                  uint32_t* __restrict__ keyrow, uint32_t lane, uint32_t stride) {
    const uint32_t used = has_clip ? rig.clip_chan_count[clip] : 0u;
    for (uint32_t c = used + lane; c < rig.Cmax; c += stride) keyrow[c] = 0xFFFFFFFFu;
}

// ---- wavefront-per-instance ------------------------------------------------------------
// local_t != nullptr: the local matrices were sampled by k_sample_nodes (lane = instance; see there) and are
// read from local_t[instance][node][12] — one contiguous run per wave; with [node][instance] a wave's 64 lanes touched
// 64 rows 480 KB apart per load (round 3: this kernel 0.145 -> 0.114 ms on C3, the writer 0.114 -> 0.124); else they
// are sampled here, lane = node.
__global__ __launch_bounds__(64) void k_pose_wave(const RigDev rig, const CrowdDev cr, const uint32_t quirk, const float* __restrict__ local_t) {
    extern __shared__ float4 node_s[];  // [N][3] rows
    const uint32_t i = blockIdx.x, lane = threadIdx.x;
    if (local_t) {
        for (uint32_t n = lane; n < rig.N; n += 64) {
            const Rows l = load_rows(local_t + ((size_t)i * rig.N + n) * 12);  // [instance][node]: this kernel's layout (launch_pose)
            node_s[n * 3 + 0] = l.r0; node_s[n * 3 + 1] = l.r1; node_s[n * 3 + 2] = l.r2;
        }
    } else {
        const uint32_t clip = cr.clip[i];
        const bool has_clip = clip < rig.clips;
        const float tau = cr.tau[i];
        uint32_t* keyrow = cr.keyidx + (size_t)i * rig.Cmax;
        clear_unused_keys(rig, clip, has_clip, keyrow, lane, 64);
        for (uint32_t n = lane; n < rig.N; n += 64) {
            const Rows l = node_local(rig, clip, has_clip, n, tau, quirk, keyrow);
            node_s[n * 3 + 0] = l.r0; node_s[n * 3 + 1] = l.r1; node_s[n * 3 + 2] = l.r2;
        }
    }
    __syncthreads();
    Rows root;
    if (cr.root) root = load_rows(cr.root + (size_t)i * 12);
    for (uint32_t lv = cr.root ? 0u : 1u; lv < rig.levels; ++lv) {
        const uint32_t e = rig.level_start[lv + 1];
        for (uint32_t idx = rig.level_start[lv] + lane; idx < e; idx += 64) {
            const uint32_t n = rig.level_nodes[idx];
            const int32_t p = rig.parent[n];
            Rows l, pw;
            l.r0 = node_s[n * 3 + 0]; l.r1 = node_s[n * 3 + 1]; l.r2 = node_s[n * 3 + 2];
            if (p >= 0) { pw.r0 = node_s[p * 3 + 0]; pw.r1 = node_s[p * 3 + 1]; pw.r2 = node_s[p * 3 + 2]; }
            else pw = root;
            const Rows w = mul_affine(pw, l);
            node_s[n * 3 + 0] = w.r0; node_s[n * 3 + 1] = w.r1; node_s[n * 3 + 2] = w.r2;
        }
        __syncthreads();
    }
    float4* wout = reinterpret_cast<float4*>(cr.world + (size_t)i * rig.N * 12);
    for (uint32_t e = lane; e < rig.N * 3; e += 64) wout[e] = node_s[e];
    for (uint32_t j = lane; j < rig.J; j += 64) {
        const uint32_t n = rig.joint_node[j];
        Rows w;
        w.r0 = node_s[n * 3 + 0]; w.r1 = node_s[n * 3 + 1]; w.r2 = node_s[n * 3 + 2];
        palette_entry(w, load_rows(rig.ibm + (size_t)j * 12), cr.palette + ((size_t)i * rig.J + j) * kPalFloats);
    }
}

// ---- channel sampling with lane = instance ----------------------------------------------------
// Thread = (node n, instance i), i fastest: the 64 lanes of a wave sample the SAME node's channels for 64
// instances.  When they play the same clip (the common case) every table load of the wave — channel record, key
// times of the binary search, key values — falls into the same one or two cache lines, where the lane = node
// mapping touched 64 different lines per load (C3: 619 such loads per wave, the pose stage's largest cost).  The
// arithmetic is node_local()'s, so key indices and matrices are bit-identical.  Output, in the layout its consumer
// wants: local_t[n][i][12] for the chain kernels (k_propagate_ring / _quad: lane = instance there too; 48 contiguous
// bytes per lane = fully coalesced), local_t[i][n][12] for k_pose_wave (inst_major: 48-byte pieces 12.5 KB apart —
// the writer pays 10 us for what saves the reader 30).
__global__ __launch_bounds__(256) void k_sample_nodes(const RigDev rig, const CrowdDev cr, const uint32_t quirk, float* __restrict__ local_t,
                                                      const uint32_t iblocks, const uint32_t inst_major) {
    const uint32_t n = blockIdx.x / iblocks, i = (blockIdx.x - n * iblocks) * 256 + threadIdx.x;
    if (n >= rig.N || i >= cr.I) return;
    const uint32_t clip = cr.clip[i];
    const bool has_clip = clip < rig.clips;
    uint32_t* keyrow = cr.keyidx + (size_t)i * rig.Cmax;
    if (n == 0) clear_unused_keys(rig, clip, has_clip, keyrow, 0, 1);
    const Rows l = node_local(rig, clip, has_clip, n, cr.tau[i], quirk, keyrow);
    store_rows(local_t + (inst_major ? (size_t)i * rig.N + n : (size_t)n * cr.I + i) * 12, l);
}

// ---- split path -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sample_local(const RigDev rig, const CrowdDev cr, const uint32_t quirk) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (uint64_t)cr.I * rig.N) return;
    const uint32_t i = (uint32_t)(g / rig.N), n = (uint32_t)(g % rig.N);
    const uint32_t clip = cr.clip[i];
    const bool has_clip = clip < rig.clips;
    uint32_t* keyrow = cr.keyidx + (size_t)i * rig.Cmax;
    if (n == 0) clear_unused_keys(rig, clip, has_clip, keyrow, 0, 1);
    const Rows l = node_local(rig, clip, has_clip, n, cr.tau[i], quirk, keyrow);
    store_rows(cr.world + g * 12, l);
}

// Deep rigs (C5: a 4096-node chain) leave no parallelism across nodes, so the parallelism is taken
// INSIDE the 3x4 product: 16 lanes per instance (12 used), lane e = row e/4, column e%4 of the
// current world matrix.  A lane needs its own ROW of the parent's world matrix — the 4 lanes of its
// DPP quad, fetched with quad broadcasts, no memory — and one COLUMN of the node's local matrix,
// which does not depend on the chain and is prefetched a chunk of nodes ahead.  The dependent
// chain per node is 4 DPP moves + 6 float ops; the element order is mul_row()'s, so the result is
// bit-equal to the level-parallel kernel and to the oracle.
template <int CTRL>
__device__ __forceinline__ float quad_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

constexpr int kPropChunk = 8;  // nodes whose local columns are in registers ahead of the chain

__global__ __launch_bounds__(64) void k_propagate_quad(const RigDev rig, const CrowdDev cr, const float* __restrict__ local_t) {
    const uint32_t lane = threadIdx.x, e = lane & 15;
    const uint32_t i = blockIdx.x * 4 + (lane >> 4);
    const bool act = i < cr.I && e < 12;
    const uint32_t r = e >> 2, c = e & 3;
    float* w = cr.world + (size_t)(act ? i : 0) * rig.N * 12;  // locals in, worlds out (in place)
    const bool has_root = cr.root != nullptr;
    const float root_e = (has_root && act) ? cr.root[(size_t)i * 12 + e] : 0.0f;
    const uint32_t N = rig.N;

    float l0[kPropChunk], l1[kPropChunk], l2[kPropChunk];  // column c of the chunk's local matrices
    auto load_chunk = [&](uint32_t base, float (&a0)[kPropChunk], float (&a1)[kPropChunk], float (&a2)[kPropChunk]) {
#pragma unroll
        for (int d = 0; d < kPropChunk; ++d) {
            const uint32_t n = base + d;
            const bool ok = act && n < N;
            // locals: written in place by k_sample_local, or local_t[n][i][12] from k_sample_nodes
            const float* q = local_t ? local_t + ((size_t)n * cr.I + (act ? i : 0)) * 12 + c : w + (size_t)n * 12 + c;
            a0[d] = ok ? q[0] : 0.0f;
            a1[d] = ok ? q[4] : 0.0f;
            a2[d] = ok ? q[8] : 0.0f;
        }
    };
    load_chunk(0, l0, l1, l2);
    float prev = 0.0f;  // this lane's element of the previous node's world matrix
    for (uint32_t base = 0; base < N; base += kPropChunk) {
        float n0[kPropChunk], n1[kPropChunk], n2[kPropChunk];
        load_chunk(base + kPropChunk, n0, n1, n2);  // past the end: zeros, never used
#pragma unroll
        for (int d = 0; d < kPropChunk; ++d) {
            const uint32_t n = base + d;
            if (n < N) {  // wave-uniform
                const int32_t p = rig.parent[n];
                float pe = prev;
                bool mul = true;
                if (p < 0) {
                    pe = root_e;
                    mul = has_root;
                } else if (p != (int32_t)n - 1) {
                    pe = act ? w[(size_t)p * 12 + e] : 0.0f;  // written earlier by this same lane
                }
                const float ax = quad_bcast<0x00>(pe), ay = quad_bcast<0x55>(pe), az = quad_bcast<0xAA>(pe), aw = quad_bcast<0xFF>(pe);
                float out = (ax * l0[d] + ay * l1[d]) + az * l2[d];
                if (c == 3) out = out + aw;
                if (!mul) out = r == 0 ? l0[d] : (r == 1 ? l1[d] : l2[d]);  // root without a root transform: world = local
                if (act) w[(size_t)n * 12 + e] = out;
                prev = out;
            }
        }
#pragma unroll
        for (int d = 0; d < kPropChunk; ++d) { l0[d] = n0[d]; l1[d] = n1[d]; l2[d] = n2[d]; }
    }
}

// ---- the same chain with the locals prefetched through an LDS ring ------------------------------------------------
// k_propagate_quad needs ~10 dependent instructions per node, yet took 264 ns per node (C5: 1.08 ms for the 4096-node
// chain, with the GPU nearly empty — 16 lanes per instance is about one wave per CU): it waited for memory.  The next
// chunk's local columns were requested one chunk of arithmetic (~0.3 us) ahead and come from HBM, the parents' scalar
// loads were waited for in every chunk, and deepening the register prefetch does not survive the compiler (a ring of
// register arrays carried round the loop is copied at the back-edge, with a wait for every load in flight).
// Here a second wave does nothing but fetch: block = 2 waves x 4 instances.  The LOADER wave moves whole chunks —
// kPropChunk nodes x 4 instances x 48 B, contiguous per node in local_t[n][i][12], plus the chunk's parents — from
// global memory straight into an LDS ring with LDS-DMA (no registers, nothing for a compiler to copy), kPropDepth
// chunks ahead; its vmcnt counts nothing but those transfers, so "chunk j+1 has landed" is one counted wait.  The
// CHAIN wave reads its column and the parents from LDS and writes the world matrices; it never waits for memory.  One
// LDS-only barrier per chunk hands a slot over.  Element order is k_propagate_quad's: bit-equal.
constexpr int kRingChunk = 16;                             // nodes per chunk: 16 x 12 16-byte pieces = 3 DMA instructions
constexpr int kRingDepth = 8;                              // chunks in flight
constexpr int kRingSlots = kRingDepth + 1;
constexpr int kRingSlotBytes = kRingChunk * 192 + 64;      // locals + parents

__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(128) void k_propagate_ring(const RigDev rig, const CrowdDev cr, const float* __restrict__ local_t) {
    __shared__ __align__(16) char ring[kRingSlots * kRingSlotBytes];
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t N = rig.N, I = cr.I, i0 = blockIdx.x * 4;
    const uint32_t nchunks = (N + kRingChunk - 1) / kRingChunk;
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring;

    if (wave == 1) {
        // ================= loader wave =================
        // piece p = k * 64 + lane (k = 0..2) of a chunk: node p / 12, then (instance, row) = p % 12
        auto issue = [&](uint32_t j) {
            const uint32_t c0 = min(j, nchunks - 1) * kRingChunk;  // past the end: the last chunk again, into a slot nobody reads
            const float* gbase = local_t + (size_t)c0 * I * 12;     // wave-uniform
            const uint32_t slot = ring_base + (j % kRingSlots) * kRingSlotBytes;
#pragma unroll
            for (uint32_t k = 0; k < 3; ++k) {
                const uint32_t pc = k * 64 + lane, nd = min(pc / 12, N - 1 - c0), q = pc % 12;
                const uint32_t inst = min(i0 + q / 3, I - 1);  // a block's idle instances shadow the last real one
                const uint32_t voff = (nd * I + inst) * 48 + (q % 3) * 16;
                const uint32_t ldsb = slot + k * 1024;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(gbase), "s"(ldsb) : "memory", "m0");
            }
            {
                const int32_t* pbase = rig.parent + c0;
                const uint32_t voff = min(lane, N - 1 - c0) * 4;
                const uint32_t ldsb = slot + kRingChunk * 192;
                if (lane < 16) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(pbase), "s"(ldsb) : "memory", "m0");
            }
        };
        for (uint32_t j = 0; j < (uint32_t)kRingDepth; ++j) issue(j);
        for (uint32_t j = 0; j <= nchunks; ++j) {
            // every chunk is exactly 4 transfers: all but the newest kRingDepth - 1 chunks have landed
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(4 * (kRingDepth - 1)) : "memory");
            ring_barrier();  // chunk j is the chain wave's; the slot of chunk j - 1 is free again
            if (j < nchunks) issue(j + kRingDepth);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may still be writing LDS when the block ends
        return;
    }

    // ================= chain wave =================
    const uint32_t e = lane & 15, q = lane >> 4;
    const uint32_t i = i0 + q;
    const bool act = i < I && e < 12;
    const uint32_t r = e >> 2, c = e & 3;
    float* w = cr.world + (size_t)min(i, I - 1) * N * 12;
    const bool has_root = cr.root != nullptr;
    float root_e = has_root ? cr.root[(size_t)min(i, I - 1) * 12 + min(e, 11u)] : 0.0f;
    // waited for HERE, once: left pending, the compiler answers every later read of root_e — one per node — with
    // s_waitcnt vmcnt(0), and in the loop that means waiting for the previous node's store
    asm volatile("" : "+v"(root_e));
    float prev = 0.0f;  // this lane's element of the previous node's world matrix
    for (uint32_t j = 0; j < nchunks; ++j) {
        ring_barrier();  // chunk j has landed
        const char* slot = ring + (j % kRingSlots) * kRingSlotBytes;
        const float* loc = reinterpret_cast<const float*>(slot) + q * 12 + c;  // + node * 48 floats: column c of my instance's local
        const int32_t* par = reinterpret_cast<const int32_t*>(slot + kRingChunk * 192);
        // the whole chunk's LDS reads are issued before the chain starts (a ragged last chunk reads the clamped
        // duplicates the loader put there), so their latency is paid once per chunk, not once per node
        float L0[kRingChunk], L1[kRingChunk], L2[kRingChunk];
        int32_t P[kRingChunk];
#pragma unroll
        for (int d = 0; d < kRingChunk; ++d) { L0[d] = loc[d * 48]; L1[d] = loc[d * 48 + 4]; L2[d] = loc[d * 48 + 8]; P[d] = par[d]; }
        // A whole chunk of plain chain links (parent = the node before, the common case in a deep rig) needs no per-node
        // tests: with one wave per SIMD every instruction costs its full issue latency, and the tests, selects and
        // per-store EXEC masks were two thirds of the instructions per node.  Lane d checks node d's parent.
        const uint32_t c0 = j * kRingChunk;
        const bool link = lane >= (uint32_t)kRingChunk || par[lane] == (int32_t)(c0 + lane) - 1;
        if (c0 + kRingChunk <= N && c0 > 0 && __ballot(!link) == 0ull) {  // wave-uniform
            if (act) {  // one EXEC mask for the chunk; quads 0-2 of a live instance are complete, which is all DPP needs
                float* wn = w + (size_t)c0 * 12 + e;
#pragma unroll
                for (int d = 0; d < kRingChunk; ++d) {
                    const float ax = quad_bcast<0x00>(prev), ay = quad_bcast<0x55>(prev), az = quad_bcast<0xAA>(prev), aw = quad_bcast<0xFF>(prev);
                    float out = (ax * L0[d] + ay * L1[d]) + az * L2[d];
                    if (c == 3) out = out + aw;
                    wn[d * 12] = out;
                    prev = out;
                }
            }
            continue;
        }
#pragma unroll
        for (int d = 0; d < kRingChunk; ++d) {
            const uint32_t n = c0 + d;
            if (n < N) {  // wave-uniform
                const float l0 = L0[d], l1 = L1[d], l2 = L2[d];
                const int32_t p = __builtin_amdgcn_readfirstlane(P[d]);
                float pe = prev;
                bool mul = true;
                if (p < 0) {
                    pe = root_e;
                    mul = has_root;
                } else if (p != (int32_t)n - 1) {
                    pe = w[(size_t)p * 12 + min(e, 11u)];  // written earlier by this same lane (rare: chains are the common case)
                    // consumed HERE, so that the wait for this load stays inside the branch: placed at the join it would
                    // make every node wait for the previous node's store
                    asm volatile("" : "+v"(pe));
                }
                const float ax = quad_bcast<0x00>(pe), ay = quad_bcast<0x55>(pe), az = quad_bcast<0xAA>(pe), aw = quad_bcast<0xFF>(pe);
                float out = (ax * l0 + ay * l1) + az * l2;
                if (c == 3) out = out + aw;
                if (!mul) out = r == 0 ? l0 : (r == 1 ? l1 : l2);  // root without a root transform: world = local
                if (act) w[(size_t)n * 12 + e] = out;
                prev = out;
            }
        }
    }
    ring_barrier();  // the loader's last one
}

__global__ __launch_bounds__(256) void k_palette(const RigDev rig, const CrowdDev cr) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (uint64_t)cr.I * rig.J) return;
    const uint32_t i = (uint32_t)(g / rig.J), j = (uint32_t)(g % rig.J);
    const Rows w = load_rows(cr.world + ((size_t)i * rig.N + rig.joint_node[j]) * 12);
    palette_entry(w, load_rows(rig.ibm + (size_t)j * 12), cr.palette + g * kPalFloats);
}

// glTF "weights" channels (SKINRT_PATH_WEIGHTS): thread = (instance, morph target m).  An instance whose clip has
// such a channel takes weight m from component m of the channel's sample — the rules of sample_channel (a2-a4),
// one component; every other instance copies the weight the host set.  The reference stops at todo!() here
// (loader/gltf/mod.rs:1312); the oracle restates this rule (orc_sample_weights).
__global__ __launch_bounds__(256) void k_sample_weights(const RigDev rig, const CrowdDev cr, float* __restrict__ mw_out, const uint32_t quirk) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (uint64_t)cr.I * rig.M) return;
    const uint32_t i = (uint32_t)(g / rig.M), mt = (uint32_t)(g - (uint64_t)i * rig.M);
    const uint32_t clip = cr.clip[i];
    WeightSrc ws{-1, 0, 0};
    if (clip < rig.clips) ws = rig.weight_src[(size_t)clip * rig.M + mt];
    const int32_t c = ws.chan;
    if (c < 0) { mw_out[g] = cr.mw_in[g]; return; }
    // the channel's keys are M floats wide, M = the targets of ITS mesh section; m = this target's component
    const uint32_t M = ws.width, m = mt - ws.first;
    const ChanRec ch = rig.chans[c];
    const float* tm = rig.times + ch.time_off;
    const float* v = rig.values + ch.value_off;
    const uint32_t n = ch.key_count;
    const float tau = cr.tau[i];
    uint32_t k;
    bool clamped = true;
    if (n == 1 || tau <= tm[0]) k = 0;
    else if (tau >= tm[n - 1]) k = n - 1;
    else {
        uint32_t lo = 0, hi = n - 1;  // times[lo] <= tau < times[hi]
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (tm[mid] <= tau) lo = mid; else hi = mid;
        }
        k = lo;
        clamped = false;
    }
    if (m == 0) cr.keyidx[(size_t)i * rig.Cmax + ch.local] = k;
    float out;
    if (ch.interp == SKINRT_INTERP_CUBICSPLINE) {
        if (clamped) out = v[(size_t)(k * 3 + 1) * M + m];
        else {
            const float t0 = tm[k], t1 = tm[k + 1];
            const float u = (tau - t0) / (t1 - t0);
            const float td = t1 - t0;
            const float t3 = (u * u) * u;
            const float t2 = quirk ? t3 : u * u;
            const float c1 = 2.0f * t3 - 3.0f * t2 + 1.0f;
            const float c2 = t3 - 2.0f * t2 + u;
            const float c3 = -2.0f * t3 + 3.0f * t2;
            const float c4 = t3 - t2;
            const float vk = v[(size_t)(k * 3 + 1) * M + m], bk = v[(size_t)(k * 3 + 2) * M + m];
            const float vk1 = v[(size_t)((k + 1) * 3 + 1) * M + m], ak1 = v[(size_t)((k + 1) * 3 + 0) * M + m];
            const float first = vk * c1;
            const float second = bk * td * c2;
            const float third = vk1 * c3;
            const float forth = ak1 * td * c4;
            out = first + second + third + forth;
        }
    } else if (clamped || ch.interp == SKINRT_INTERP_STEP) {
        out = v[(size_t)k * M + m];
    } else {
        const float t0 = tm[k], t1 = tm[k + 1];
        const float u = (tau - t0) / (t1 - t0);
        out = lerp1(v[(size_t)k * M + m], v[(size_t)(k + 1) * M + m], 1.0f - u, u);
    }
    mw_out[g] = out;
}

// skinrt_advance_clocks: tau = fmodf(tau + dt * rate, clip_length) — the product and the sum round once each
// (contraction is off in this file), fmodf is exact.  Instances at rest and clips of length 0 keep their clock.
__global__ __launch_bounds__(256) void k_advance_clocks(float* __restrict__ tau, const uint32_t* __restrict__ clip, const float* __restrict__ clip_length,
                                                        const uint32_t clips, const uint32_t I, const float dt, const float* __restrict__ rate) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= I) return;
    const uint32_t c = clip[i];
    if (c >= clips) return;
    const float len = clip_length[c];
    if (!(len > 0.0f)) return;
    const float step = rate ? dt * rate[i] : dt;
    tau[i] = fmodf(tau[i] + step, len);
}

// InstanceUniform{transform, normal, padding} (renderer/src/renderer/uniform/transform.rs:8-14) of node nodes[n] of
// instance first + k: transform = the world matrix (column-major, last row 0,0,0,1), normal = the columns of
// Mat3(inverse(transform)^T) padded with 0 (:23-34) — the palette's arithmetic (normal_rows) on the world matrix.
__global__ __launch_bounds__(256) void k_export_instance_uniform(const float* __restrict__ world, const uint32_t N, const uint32_t first,
                                                                 const uint32_t count, const uint32_t* __restrict__ nodes, const uint32_t node_count,
                                                                 const uint32_t stride_dwords, float* __restrict__ dst) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (uint64_t)count * node_count) return;
    const uint32_t k = (uint32_t)(g / node_count), q = (uint32_t)(g - (uint64_t)k * node_count);
    const uint32_t n = nodes[q];
    float* o = dst + g * stride_dwords;
    if (n >= N) {
        for (int e = 0; e < 31; ++e) o[e] = 0.0f;
        return;
    }
    const Rows m = load_rows(world + ((size_t)(first + k) * N + n) * 12);
    const Normal9 nm = normal_rows(m);
    o[0] = m.r0.x; o[1] = m.r1.x; o[2] = m.r2.x; o[3] = 0.0f;
    o[4] = m.r0.y; o[5] = m.r1.y; o[6] = m.r2.y; o[7] = 0.0f;
    o[8] = m.r0.z; o[9] = m.r1.z; o[10] = m.r2.z; o[11] = 0.0f;
    o[12] = m.r0.w; o[13] = m.r1.w; o[14] = m.r2.w; o[15] = 1.0f;
    o[16] = nm.n00; o[17] = nm.n10; o[18] = nm.n20; o[19] = 0.0f;  // column 0 of the normal matrix, pad_vec3
    o[20] = nm.n01; o[21] = nm.n11; o[22] = nm.n21; o[23] = 0.0f;
    o[24] = nm.n02; o[25] = nm.n12; o[26] = nm.n22; o[27] = 0.0f;
    o[28] = 0.0f; o[29] = 0.0f; o[30] = 0.0f;                       // padding: [u8; 12]
}

}  // namespace

hipError_t launch_advance_clocks(const RigDev& rig, float* tau, const uint32_t* clip, uint32_t I, float dt, const float* rate, hipStream_t s) {
    if (I == 0 || rig.clips == 0) return hipSuccess;
    if (!tau || !clip || !rig.clip_length) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_advance_clocks, dim3((I + 255) / 256), dim3(256), 0, s, tau, clip, rig.clip_length, rig.clips, I, dt, rate);
    return hipGetLastError();
}

hipError_t launch_export_instance_uniform(const CrowdDev& cr, uint32_t N, uint32_t first, uint32_t count, const uint32_t* nodes,
                                          uint32_t node_count, uint32_t stride, void* dst, hipStream_t s) {
    const uint64_t n = (uint64_t)count * node_count;
    if (n == 0) return hipSuccess;
    if ((uint64_t)first + count > cr.I || !nodes || !dst || stride < SKINRT_INSTANCE_UNIFORM_BYTES || (stride & 3u) ||
        (reinterpret_cast<uintptr_t>(dst) & 3u) || (n + 255) / 256 > 0x7fffffffull)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_export_instance_uniform, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, cr.world, N, first, count, nodes, node_count,
                       stride / 4, static_cast<float*>(dst));
    return hipGetLastError();
}

hipError_t launch_sample_weights(const RigDev& rig, const CrowdDev& cr, float* mw_out, uint32_t quirk, hipStream_t s) {
    const uint64_t n = (uint64_t)cr.I * rig.M;
    if (n == 0) return hipSuccess;
    if (!mw_out || !cr.mw_in || !rig.weight_src || (n + 255) / 256 > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_sample_weights, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, rig, cr, mw_out, quirk);
    return hipGetLastError();
}

hipError_t launch_pose(const RigDev& rig, const CrowdDev& cr, uint32_t quirk, PosePath path, float* local_t, hipStream_t s) {
    if (cr.I == 0) return hipSuccess;
    if (local_t) {
        const uint32_t iblocks = (cr.I + 255) / 256;
        if ((uint64_t)iblocks * rig.N > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL(k_sample_nodes, dim3(iblocks * rig.N), dim3(256), 0, s, rig, cr, quirk, local_t, iblocks, path == kPoseWave ? 1u : 0u);
    }
    if (path == kPoseWave) {
        const size_t lds = (size_t)rig.N * 48;
        hipLaunchKernelGGL(k_pose_wave, dim3(cr.I), dim3(64), lds, s, rig, cr, quirk, local_t);
        return hipGetLastError();
    }
    const uint64_t in = (uint64_t)cr.I * rig.N;
    if (!local_t) hipLaunchKernelGGL(k_sample_local, dim3((uint32_t)((in + 255) / 256)), dim3(256), 0, s, rig, cr, quirk);
    // (the ring kernel addresses a chunk of 16 nodes with 32-bit byte offsets)
    if (local_t && (uint64_t)kRingChunk * cr.I * 48 < 0xffffffffull) hipLaunchKernelGGL(k_propagate_ring, dim3((cr.I + 3) / 4), dim3(128), 0, s, rig, cr, local_t);
    else hipLaunchKernelGGL(k_propagate_quad, dim3((cr.I + 3) / 4), dim3(64), 0, s, rig, cr, local_t);
    if (rig.J) {
        const uint64_t ij = (uint64_t)cr.I * rig.J;
        hipLaunchKernelGGL(k_palette, dim3((uint32_t)((ij + 255) / 256)), dim3(256), 0, s, rig, cr);
    }
    return hipGetLastError();
}

}  // namespace skinrt
