// kernels_skin.hip — morph-target blending + linear-blend skinning (+ per-tile bounds) for
// gfx950.
//
// Shape: one 256-thread block = one tile of 256 vertices x one range of instances.  A thread
// keeps ITS vertex (base position/normal/tangent, 4K weights, 4K tile-local joint indices) in
// registers for the whole block and loops over the instances, so the shared mesh is read once
// per range and every instance costs 36 B of stores per vertex.  The joint palette of a batch of
// instances is staged into LDS (only the joints the tile references: rig_compile builds per-tile
// joint lists and rewrites the vertex indices to list positions); each influence is then six
// ds_read_b128 whose register pairs feed v_pk_fma_f32 directly.  Morph targets are applied by a
// wave-uniform loop over the tile's (tile, target) pairs that have a non-zero weight for the
// instance; pair data streams from L2 (the block -> XCD mapping keeps a tile's blocks on one XCD).
//
// Arithmetic order is the WGSL's (renderer/src/shader/shader.wgsl:127-141, :159-163):
//   S  = ((w0*M[j0] + w1*M[j1]) + w2*M[j2]) + w3*M[j3]   (further weight sets continue the chain)
//   p' = ((S.c0*x + S.c1*y) + S.c2*z) + S.c3             n' = (N.c0*x + N.c1*y) + N.c2*z
// EXACT=true: no FMA contraction -> bit-equal to the fp32 oracle.  EXACT=false (default): the
// same association with fused multiply-adds; within 1e-5 rel / 1e-6 abs.
#include "device_types.hpp"

namespace skinrt {

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool EXACT>
__device__ __forceinline__ float madd(float a, float b, float c) {  // a*b + c
    if constexpr (EXACT) {
#pragma clang fp contract(off)
        const float p = a * b;
        return p + c;
    } else {
        return __builtin_fmaf(a, b, c);
    }
}
__device__ __forceinline__ float mul_nc(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_nc(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
template <bool EXACT>
__device__ __forceinline__ f2 madd2(float a, f2 b, f2 c) {  // a*b + c on a register pair
    const f2 av = {a, a};
    if constexpr (EXACT) {
#pragma clang fp contract(off)
        const f2 p = av * b;
        return p + c;
    } else {
        return __builtin_elementwise_fma(av, b, c);
    }
}
__device__ __forceinline__ f2 mul2(float a, f2 b) {
#pragma clang fp contract(off)
    const f2 av = {a, a};
    return av * b;
}

// min / max over the 16 lanes of each DPP row for six values at once (x,y,z mins then maxs).
// Hand-written because fminf/fmaxf + update_dpp compile to mov_dpp + canonicalize + min (4x the
// instructions).  Six independent chains interleave, which also covers the VALU->DPP hazard.
__device__ __forceinline__ void row_minmax6(float& a0, float& a1, float& a2, float& b0, float& b1, float& b2) {
#define SKINRT_DPP_STEP(CTRL)                                                        \
    asm volatile("s_nop 1\n\t"                                                       \
                 "v_min_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_min_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_min_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_max_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_max_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_max_f32_dpp %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2))
    SKINRT_DPP_STEP("quad_perm:[1,0,3,2]");
    SKINRT_DPP_STEP("quad_perm:[2,3,0,1]");
    SKINRT_DPP_STEP("row_half_mirror");
    SKINRT_DPP_STEP("row_mirror");
#undef SKINRT_DPP_STEP
}

struct LdsPal {
    const f4* base;  // this instance's staged palette: [list position][6]
    __device__ __forceinline__ const f4* entry(uint32_t idx) const { return base + idx * 6; }
};
struct GlobalPal {
    const f4* base;  // this instance's full palette in HBM/L2: [joint][6]
    __device__ __forceinline__ const f4* entry(uint32_t idx) const { return base + idx * 6; }
};

// Position pass then normal/tangent pass: the two accumulators are never live together and each
// influence's rows are consumed as they arrive, which keeps the kernel at 4 waves per SIMD.
template <bool EXACT, int K, class Pal>
__device__ __forceinline__ void skin_position(const Pal pal, const float (&w)[4 * K], const uint32_t (&li)[4 * K],
                                              const float (&p)[3], float (&op)[3]) {
    f2 S[6];  // rows of the blended 3x4: S[2r] = (m_r0, m_r1), S[2r+1] = (m_r2, m_r3)
#pragma unroll
    for (int a = 0; a < 4 * K; ++a) {
        const f4* e = pal.entry(li[a]);
        const f4 r0 = e[0], r1 = e[1], r2 = e[2];
        const f2 m[6] = {r0.xy, r0.zw, r1.xy, r1.zw, r2.xy, r2.zw};
#pragma unroll
        for (int x = 0; x < 6; ++x) S[x] = a == 0 ? mul2(w[0], m[x]) : madd2<EXACT>(w[a], m[x], S[x]);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float s = mul_nc(S[2 * r].x, p[0]);
        s = madd<EXACT>(S[2 * r].y, p[1], s);
        s = madd<EXACT>(S[2 * r + 1].x, p[2], s);
        op[r] = add_nc(s, S[2 * r + 1].y);
    }
}

template <bool EXACT, int K, class Pal>
__device__ __forceinline__ void skin_normal(const Pal pal, const float (&w)[4 * K], const uint32_t (&li)[4 * K],
                                            const float (&n)[3], const float (&t)[3], bool normalize,
                                            float (&on)[3], float (&ot)[3]) {
    f2 Nm[6];  // rows of the blended normal matrix, 4th component of each row is padding (0)
#pragma unroll
    for (int a = 0; a < 4 * K; ++a) {
        const f4* e = pal.entry(li[a]);
        const f4 n0 = e[3], n1 = e[4], n2 = e[5];
        const f2 q[6] = {n0.xy, n0.zw, n1.xy, n1.zw, n2.xy, n2.zw};
#pragma unroll
        for (int x = 0; x < 6; ++x) Nm[x] = a == 0 ? mul2(w[0], q[x]) : madd2<EXACT>(w[a], q[x], Nm[x]);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float a = mul_nc(Nm[2 * r].x, n[0]);
        a = madd<EXACT>(Nm[2 * r].y, n[1], a);
        on[r] = madd<EXACT>(Nm[2 * r + 1].x, n[2], a);
        float b = mul_nc(Nm[2 * r].x, t[0]);
        b = madd<EXACT>(Nm[2 * r].y, t[1], b);
        ot[r] = madd<EXACT>(Nm[2 * r + 1].x, t[2], b);
    }
    if (normalize) {  // shader.wgsl:217-218 — v / sqrt(dot(v,v))
        const float ln = sqrtf(madd<true>(on[2], on[2], madd<true>(on[1], on[1], mul_nc(on[0], on[0]))));
        const float lt = sqrtf(madd<true>(ot[2], ot[2], madd<true>(ot[1], ot[1], mul_nc(ot[0], ot[0]))));
#pragma unroll
        for (int r = 0; r < 3; ++r) { on[r] = on[r] / ln; ot[r] = ot[r] / lt; }
    }
}

template <bool EXACT>
__device__ __forceinline__ void morph_add(float mw, const float (&d)[9], float (&p)[3], float (&n)[3], float (&t)[3]) {
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        p[x] = madd<EXACT>(mw, d[x], p[x]);
        n[x] = madd<EXACT>(mw, d[3 + x], n[x]);
        t[x] = madd<EXACT>(mw, d[6 + x], t[x]);
    }
}

struct Outputs {
    float* pos;
    float* nrm;
    float* tan;
};

// One (instance, vertex): skin from (p, n, t), store, feed the bounds.
template <bool EXACT, int K, class Pal>
__device__ __forceinline__ void skin_store(const Pal pal, const float (&w)[4 * K], const uint32_t (&li)[4 * K],
                                           const float (&p)[3], const float (&n)[3], const float (&t)[3],
                                           bool normalize, bool valid, const Outputs& out, size_t o, float (&op)[3]) {
    float on[3], ot[3];
    skin_position<EXACT, K>(pal, w, li, p, op);
    if (valid) *reinterpret_cast<float3*>(out.pos + o) = make_float3(op[0], op[1], op[2]);
    skin_normal<EXACT, K>(pal, w, li, n, t, normalize, on, ot);
    if (valid) {
        *reinterpret_cast<float3*>(out.nrm + o) = make_float3(on[0], on[1], on[2]);
        *reinterpret_cast<float3*>(out.tan + o) = make_float3(ot[0], ot[1], ot[2]);
    }
}

template <bool EXACT, int K, bool LDSPAL>
__global__ __launch_bounds__(256, (K == 1 ? 4 : 2)) void k_skin(const RigDev rig, const CrowdDev cr, const SkinPlan plan) {
    extern __shared__ float4 smem[];
    constexpr uint32_t TV = kTileVerts;
    constexpr int W = 4 * K;
    const uint32_t tid = threadIdx.x;
    // block -> (tile, range): blocks b and b+8 share an XCD (round-robin dispatch); each XCD owns
    // a contiguous chunk of tiles x a chunk of ranges, so a tile's mesh/morph data and a range's
    // palette rows stay in ONE L2.  Placement is a speed matter only.
    const uint32_t xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const uint32_t tc = xcd % plan.tile_chunks, rc = xcd / plan.tile_chunks;
    const uint32_t tile = tc * plan.tiles_per_chunk + k % plan.tiles_per_chunk;
    const uint32_t rng = rc * plan.ranges_per_chunk + k / plan.tiles_per_chunk;
    if (tile >= rig.T || rng >= plan.ranges) return;
    // tiles whose joint list fits LDS are skinned by the LDSPAL=true launch, the rest (rare: no
    // joint locality) by the LDSPAL=false launch that gathers the palette from L2
    if ((rig.tile_lds[tile] != 0) != LDSPAL) return;
    const uint32_t i0 = rng * plan.range;
    const uint32_t i1 = min(cr.I, i0 + plan.range);
    const uint32_t JTP = 1u << plan.jt_shift;
    const uint32_t B = plan.batch;
    const uint32_t M = rig.M;
    const uint32_t P1 = plan.max_pairs + 1;
    f4* pal_s = reinterpret_cast<f4*>(smem);                              // [B][JTP][6]
    MorphPair* pairs_s = reinterpret_cast<MorphPair*>(pal_s + (size_t)B * JTP * 6);  // [max_pairs]
    float* mw_s = reinterpret_cast<float*>(pairs_s + plan.max_pairs);     // [B][M]
    float* ab_s = mw_s + (size_t)B * M;                                   // [B][16][6]
    uint32_t* act_s = reinterpret_cast<uint32_t*>(ab_s + (size_t)B * 16 * 6);  // [B][1 + max_pairs]

    // ---- this thread's vertex ------------------------------------------------------------
    const uint32_t v = tile * TV + tid;
    const bool valid = v < rig.V;
    float p0[3], n0[3], t0[3], w[W];
    uint32_t li[W];
    {
        const float* b = rig.base + (size_t)tile * 9 * TV + tid;
#pragma unroll
        for (int e = 0; e < 3; ++e) { p0[e] = b[e * TV]; n0[e] = b[(3 + e) * TV]; t0[e] = b[(6 + e) * TV]; }
        const float* wt = rig.wts + (size_t)tile * W * TV + tid;
        const uint16_t* lp = rig.lidx + (size_t)tile * W * TV + tid;
#pragma unroll
        for (int a = 0; a < W; ++a) { w[a] = wt[a * TV]; li[a] = lp[a * TV]; }
    }
    // morph: this tile's (tile, target) pairs; their records live in LDS for the whole block
    const uint32_t pbeg = rig.tile_pair_start[tile];
    const uint32_t np = rig.tile_pair_start[tile + 1] - pbeg;
    for (uint32_t c = tid; c < np; c += 256) pairs_s[c] = rig.pairs[pbeg + c];
    const uint32_t jstart = rig.tile_joint_start[tile];
    const uint32_t jcnt = rig.tile_joint_start[tile + 1] - jstart;
    const f4* pal_g = reinterpret_cast<const f4*>(cr.palette);
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const Outputs out{cr.pos, cr.nrm, cr.tan};
    const bool normalize = plan.normalize != 0;

    for (uint32_t ib = i0; ib < i1; ib += B) {
        const uint32_t nb = min(B, i1 - ib);
        __syncthreads();  // previous batch's readers are done with LDS (and pairs_s is visible)
        if (LDSPAL) {
            const uint32_t chunks = (nb << plan.jt_shift) * 6;
            for (uint32_t c = tid; c < chunks; c += 256) {
                const uint32_t row = c / 6, q = c - row * 6;
                const uint32_t b = row >> plan.jt_shift, lj = row & (JTP - 1);
                if (lj < jcnt) {
                    const uint32_t j = rig.tile_joints[jstart + lj];
                    pal_s[c] = pal_g[((size_t)(ib + b) * rig.J + j) * 6 + q];
                }
            }
        }
        for (uint32_t c = tid; c < nb * M; c += 256) mw_s[c] = cr.mw[(size_t)ib * M + c];
        // per instance: which of the tile's pairs have a non-zero weight (zero weights add exactly 0)
        if (tid < nb) {
            uint32_t* act = act_s + tid * P1;
            const float* mwg = cr.mw + (size_t)(ib + tid) * M;
            uint32_t n = 0;
            for (uint32_t q = 0; q < np; ++q)
                if (mwg[pairs_s[q].target] != 0.0f) act[++n] = q;
            act[0] = n;
        }
        __syncthreads();

        for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t i = ib + b;
            const size_t o = ((size_t)i * rig.V + v) * 3;
            const uint32_t* act = act_s + b * P1;
            const uint32_t na = __builtin_amdgcn_readfirstlane(act[0]);
            float op[3];
            auto skin = [&](const float (&p)[3], const float (&n)[3], const float (&t)[3]) {
                if constexpr (LDSPAL) skin_store<EXACT, K>(LdsPal{pal_s + (size_t)b * JTP * 6}, w, li, p, n, t, normalize, valid, out, o, op);
                else skin_store<EXACT, K>(GlobalPal{pal_g + (size_t)i * rig.J * 6}, w, li, p, n, t, normalize, valid, out, o, op);
            };
            if (na == 0) {
                skin(p0, n0, t0);  // no active morph target touches this tile for this instance
            } else {
                // a13: p = p0 + sum_m w_m * delta_m, targets ascending (wave-uniform loop)
                float p[3] = {p0[0], p0[1], p0[2]}, n[3] = {n0[0], n0[1], n0[2]}, t[3] = {t0[0], t0[1], t0[2]};
                for (uint32_t a = 1; a <= na; ++a) {
                    const uint32_t q = __builtin_amdgcn_readfirstlane(act[a]);
                    const MorphPair mp = pairs_s[q];
                    const uint32_t doff = __builtin_amdgcn_readfirstlane(mp.data_off);
                    const float mw = mw_s[b * M + __builtin_amdgcn_readfirstlane((uint32_t)mp.target)];
                    if (__builtin_amdgcn_readfirstlane(mp.dense)) {
                        const float* blk = rig.mblock + doff + tid;
                        float d[9];
#pragma unroll
                        for (int x = 0; x < 9; ++x) d[x] = blk[x * TV];
                        morph_add<EXACT>(mw, d, p, n, t);
                    } else {
                        const uint32_t slot = rig.mmap[__builtin_amdgcn_readfirstlane(mp.map_off) + tid];
                        if (slot) {
                            const float* dp = rig.mblock + doff + (slot - 1) * 9;
                            float d[9];
#pragma unroll
                            for (int x = 0; x < 9; ++x) d[x] = dp[x];
                            morph_add<EXACT>(mw, d, p, n, t);
                        }
                    }
                }
                skin(p, n, t);
            }
            if (plan.aabb) {
                // lanes past the mesh end repeat a real vertex (rig_compile), so no mask is needed
                float a0 = op[0], a1 = op[1], a2 = op[2], b0 = op[0], b1 = op[1], b2 = op[2];
                row_minmax6(a0, a1, a2, b0, b1, b2);
                if ((lane & 15) == 0) {
                    float* dst = ab_s + ((size_t)b * 16 + wave * 4 + (lane >> 4)) * 6;
                    dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = b0; dst[4] = b1; dst[5] = b2;
                }
            }
        }
        if (plan.aabb) {
            __syncthreads();
            if (tid < nb * 6) {
                const uint32_t b = tid / 6, c = tid - b * 6;
                const float* src = ab_s + (size_t)b * 16 * 6 + c;
                float r = src[0];
                for (int q = 1; q < 16; ++q) r = c < 3 ? fminf(r, src[q * 6]) : fmaxf(r, src[q * 6]);
                cr.aabb_part[((size_t)(ib + b) * rig.T + tile) * 6 + c] = r;
            }
        }
    }
}

// a14: bounds of one instance = min/max over its tiles' partial bounds (order-independent, exact)
__global__ __launch_bounds__(256) void k_aabb_reduce(const RigDev rig, const CrowdDev cr) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (uint64_t)cr.I * 6) return;
    const uint32_t i = (uint32_t)(g / 6), c = (uint32_t)(g % 6);
    float r = c < 3 ? __builtin_inff() : -__builtin_inff();
    const float* src = cr.aabb_part + (size_t)i * rig.T * 6 + c;
    for (uint32_t t = 0; t < rig.T; ++t) r = c < 3 ? fminf(r, src[(size_t)t * 6]) : fmaxf(r, src[(size_t)t * 6]);
    cr.aabb[g] = r;
}

__global__ __launch_bounds__(256) void k_fill_mw(float* mw, const float* defaults, uint64_t total, uint32_t M) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < total) mw[g] = defaults[g % M];
}

template <bool EXACT, bool LDSPAL>
hipError_t launch_skin_k(const RigDev& rig, const CrowdDev& cr, const SkinPlan& plan, hipStream_t s) {
    const dim3 grid(8u * plan.tiles_per_chunk * plan.ranges_per_chunk), block(256);
    switch (rig.K) {
        case 1: hipLaunchKernelGGL((k_skin<EXACT, 1, LDSPAL>), grid, block, plan.lds_bytes, s, rig, cr, plan); break;
        case 2: hipLaunchKernelGGL((k_skin<EXACT, 2, LDSPAL>), grid, block, plan.lds_bytes, s, rig, cr, plan); break;
        case 3: hipLaunchKernelGGL((k_skin<EXACT, 3, LDSPAL>), grid, block, plan.lds_bytes, s, rig, cr, plan); break;
        case 4: hipLaunchKernelGGL((k_skin<EXACT, 4, LDSPAL>), grid, block, plan.lds_bytes, s, rig, cr, plan); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace

hipError_t launch_skin(const RigDev& rig, const CrowdDev& cr, const SkinPlan& plan, hipStream_t s) {
    if (cr.I == 0 || rig.T == 0) return hipSuccess;
    hipError_t e = hipSuccess;
    if (plan.lds_tiles)
        e = plan.exact ? launch_skin_k<true, true>(rig, cr, plan, s) : launch_skin_k<false, true>(rig, cr, plan, s);
    if (e == hipSuccess && plan.lds_tiles < rig.T)
        e = plan.exact ? launch_skin_k<true, false>(rig, cr, plan, s) : launch_skin_k<false, false>(rig, cr, plan, s);
    return e;
}

hipError_t launch_aabb_reduce(const RigDev& rig, const CrowdDev& cr, hipStream_t s) {
    if (cr.I == 0) return hipSuccess;
    const uint64_t n = (uint64_t)cr.I * 6;
    hipLaunchKernelGGL(k_aabb_reduce, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, rig, cr);
    return hipGetLastError();
}

hipError_t launch_fill_morph_weights(float* mw, const float* defaults, uint32_t I, uint32_t M, hipStream_t s) {
    const uint64_t n = (uint64_t)I * M;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fill_mw, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, mw, defaults, n, M);
    return hipGetLastError();
}

}  // namespace skinrt
