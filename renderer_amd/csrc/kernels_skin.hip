// kernels_skin.hip — morph-target blending + linear-blend skinning (+ per-tile bounds) for
// gfx950.
//
// Shape: one 256-thread block = one tile of 256 vertices x one range of instances.  A thread
// keeps ITS vertex (base position/normal/tangent, 4K weights, 4K tile-local joint indices) in
// registers for the whole block and loops over the instances, so the shared mesh is read from
// HBM once per range, and every instance costs 36 B of stores per vertex.  The joint palette of
// a batch of instances is staged into LDS (only the joints the tile references: rig_compile
// builds per-tile joint lists and rewrites the vertex indices to list positions), then each
// influence is 6 ds_read_b128.  Stores are float3 per lane, contiguous per wave.
//
// Arithmetic order is the WGSL's (renderer/src/shader/shader.wgsl:127-141, :159-163):
//   S  = ((w0*M[j0] + w1*M[j1]) + w2*M[j2]) + w3*M[j3]   (further weight sets continue the chain)
//   p' = ((S.c0*x + S.c1*y) + S.c2*z) + S.c3             n' = (N.c0*x + N.c1*y) + N.c2*z
// EXACT=true: no FMA contraction -> bit-equal to the fp32 oracle.  EXACT=false (default): the
// same association with fused multiply-adds; within 1e-5 rel / 1e-6 abs.
#include "device_types.hpp"

namespace skinrt {

namespace {

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
template <bool EXACT>
__device__ __forceinline__ float mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_nc(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}

// min over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ float row_reduce(float v, bool is_max) {
    auto step = [&](float x, int ctrl_sel) {
        int yi;
        const int xi = __float_as_int(x);
        switch (ctrl_sel) {
            case 0: yi = __builtin_amdgcn_update_dpp(xi, xi, 0xB1, 0xF, 0xF, false); break;   // quad_perm [1,0,3,2]
            case 1: yi = __builtin_amdgcn_update_dpp(xi, xi, 0x4E, 0xF, 0xF, false); break;   // quad_perm [2,3,0,1]
            case 2: yi = __builtin_amdgcn_update_dpp(xi, xi, 0x141, 0xF, 0xF, false); break;  // row_half_mirror
            default: yi = __builtin_amdgcn_update_dpp(xi, xi, 0x140, 0xF, 0xF, false); break; // row_mirror
        }
        const float y = __int_as_float(yi);
        return is_max ? fmaxf(x, y) : fminf(x, y);
    };
    v = step(v, 0); v = step(v, 1); v = step(v, 2); v = step(v, 3);
    return v;
}

struct LdsPal {
    const float4* base;  // this instance's staged palette: [list position][6]
    __device__ __forceinline__ const float4* entry(uint32_t idx) const { return base + idx * 6; }
};
struct GlobalPal {
    const float4* base;  // this instance's full palette in HBM/L2: [joint][6]
    __device__ __forceinline__ const float4* entry(uint32_t idx) const { return base + idx * 6; }
};

// Position pass then normal/tangent pass: the two accumulators (12 + 9 floats) are never live
// together, and each influence's rows are consumed right after they arrive, which keeps the
// kernel at <= 96 VGPRs (5 waves per SIMD) so other waves cover the LDS latency.
template <bool EXACT, int K, class Pal>
__device__ __forceinline__ void skin_position(const Pal pal, const float (&w)[4 * K], const uint32_t (&li)[4 * K],
                                              const float p[3], float op[3]) {
    float S[12];
#pragma unroll
    for (int a = 0; a < 4 * K; ++a) {
        const float4* e = pal.entry(li[a]);
        const float4 r0 = e[0], r1 = e[1], r2 = e[2];
        const float m[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
#pragma unroll
        for (int x = 0; x < 12; ++x) S[x] = a == 0 ? mul<EXACT>(w[0], m[x]) : madd<EXACT>(w[a], m[x], S[x]);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float s = mul<EXACT>(S[r * 4 + 0], p[0]);
        s = madd<EXACT>(S[r * 4 + 1], p[1], s);
        s = madd<EXACT>(S[r * 4 + 2], p[2], s);
        op[r] = add_nc(s, S[r * 4 + 3]);
    }
}

template <bool EXACT, int K, class Pal>
__device__ __forceinline__ void skin_normal(const Pal pal, const float (&w)[4 * K], const uint32_t (&li)[4 * K],
                                            const float n[3], const float t[3], bool normalize, float on[3], float ot[3]) {
    float Nm[9];
#pragma unroll
    for (int a = 0; a < 4 * K; ++a) {
        const float4* e = pal.entry(li[a]);
        const float4 n0 = e[3], n1 = e[4], n2 = e[5];
        const float q[9] = {n0.x, n0.y, n0.z, n1.x, n1.y, n1.z, n2.x, n2.y, n2.z};
#pragma unroll
        for (int x = 0; x < 9; ++x) Nm[x] = a == 0 ? mul<EXACT>(w[0], q[x]) : madd<EXACT>(w[a], q[x], Nm[x]);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float a = mul<EXACT>(Nm[r * 3 + 0], n[0]);
        a = madd<EXACT>(Nm[r * 3 + 1], n[1], a);
        on[r] = madd<EXACT>(Nm[r * 3 + 2], n[2], a);
        float b = mul<EXACT>(Nm[r * 3 + 0], t[0]);
        b = madd<EXACT>(Nm[r * 3 + 1], t[1], b);
        ot[r] = madd<EXACT>(Nm[r * 3 + 2], t[2], b);
    }
    if (normalize) {  // shader.wgsl:217-218 — v / sqrt(dot(v,v))
        const float ln = sqrtf(madd<true>(on[2], on[2], madd<true>(on[1], on[1], mul<true>(on[0], on[0]))));
        const float lt = sqrtf(madd<true>(ot[2], ot[2], madd<true>(ot[1], ot[1], mul<true>(ot[0], ot[0]))));
#pragma unroll
        for (int r = 0; r < 3; ++r) { on[r] = on[r] / ln; ot[r] = ot[r] / lt; }
    }
}

template <bool EXACT>
__device__ __forceinline__ void morph_add(float mw, const float d[9], float p[3], float n[3], float t[3]) {
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        p[x] = madd<EXACT>(mw, d[x], p[x]);
        n[x] = madd<EXACT>(mw, d[3 + x], n[x]);
        t[x] = madd<EXACT>(mw, d[6 + x], t[x]);
    }
}

template <bool EXACT, int K, bool LDSPAL>
__global__ __launch_bounds__(256, (K == 1 ? 4 : 2)) void k_skin(const RigDev rig, const CrowdDev cr, const SkinPlan plan) {
    extern __shared__ float4 smem[];
    constexpr uint32_t TV = kTileVerts;
    constexpr int W = 4 * K;
    const uint32_t tid = threadIdx.x;
    const uint32_t tile = blockIdx.x % rig.T;
    const uint32_t rng = blockIdx.x / rig.T;
    const uint32_t i0 = rng * plan.range;
    const uint32_t i1 = min(cr.I, i0 + plan.range);
    const uint32_t JTP = 1u << plan.jt_shift;
    const uint32_t B = plan.batch;
    const uint32_t M = rig.M;
    const uint32_t P1 = plan.max_pairs + 1;
    float4* pal_s = smem;                                                 // [B][JTP][6]
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
    // tiles whose joint list fits LDS are skinned by the LDSPAL=true launch, the rest (rare: no
    // joint locality) by the LDSPAL=false launch that gathers the palette from L2
    if ((rig.tile_lds[tile] != 0) != LDSPAL) return;
    const uint32_t jstart = rig.tile_joint_start[tile];
    const uint32_t jcnt = rig.tile_joint_start[tile + 1] - jstart;
    const float4* pal_g = reinterpret_cast<const float4*>(cr.palette);
    const uint32_t lane = tid & 63, wave = tid >> 6;

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
            float p[3] = {p0[0], p0[1], p0[2]}, n[3] = {n0[0], n0[1], n0[2]}, t[3] = {t0[0], t0[1], t0[2]};
            // a13: p = p0 + sum_m w_m * delta_m, targets ascending (wave-uniform loop)
            const uint32_t* act = act_s + b * P1;
            const uint32_t na = __builtin_amdgcn_readfirstlane(act[0]);
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
            float op[3], on[3], ot[3];
            const size_t o = ((size_t)i * rig.V + v) * 3;
            if constexpr (LDSPAL) {
                const LdsPal pal{pal_s + (size_t)b * JTP * 6};
                skin_position<EXACT, K>(pal, w, li, p, op);
                if (valid) *reinterpret_cast<float3*>(cr.pos + o) = make_float3(op[0], op[1], op[2]);
                skin_normal<EXACT, K>(pal, w, li, n, t, plan.normalize != 0, on, ot);
            } else {
                const GlobalPal pal{pal_g + (size_t)i * rig.J * 6};
                skin_position<EXACT, K>(pal, w, li, p, op);
                if (valid) *reinterpret_cast<float3*>(cr.pos + o) = make_float3(op[0], op[1], op[2]);
                skin_normal<EXACT, K>(pal, w, li, n, t, plan.normalize != 0, on, ot);
            }
            if (valid) {
                *reinterpret_cast<float3*>(cr.nrm + o) = make_float3(on[0], on[1], on[2]);
                *reinterpret_cast<float3*>(cr.tan + o) = make_float3(ot[0], ot[1], ot[2]);
            }
            if (plan.aabb) {
                const float inf = __builtin_inff();
                float r6[6];
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    r6[x] = row_reduce(valid ? op[x] : inf, false);
                    r6[3 + x] = row_reduce(valid ? op[x] : -inf, true);
                }
                if ((lane & 15) == 0) {
                    float* dst = ab_s + ((size_t)b * 16 + wave * 4 + (lane >> 4)) * 6;
#pragma unroll
                    for (int x = 0; x < 6; ++x) dst[x] = r6[x];
                }
            }
        }
        if (plan.aabb) {
            __syncthreads();
            if (tid < nb * 6) {
                const uint32_t b = tid / 6, c = tid - b * 6;
                const float* src = ab_s + (size_t)b * 16 * 6 + c;
                float r = src[0];
                for (int k = 1; k < 16; ++k) r = c < 3 ? fminf(r, src[k * 6]) : fmaxf(r, src[k * 6]);
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
    const dim3 grid(rig.T * plan.ranges), block(256);
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
