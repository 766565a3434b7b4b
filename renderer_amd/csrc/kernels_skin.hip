// kernels_skin.hip — morph-target blending + linear-blend skinning (+ per-tile bounds) for
// gfx950.
//
// Shape: one block = one tile of 256 vertices x one range of instances.  A thread keeps ITS
// vertex (base position/normal/tangent, 4K weights, 4K tile-local joint indices) in registers for
// the whole block and loops over the instances, so the shared mesh is read once per range and
// every instance costs 36 B of stores per vertex.  Only the joints the tile references are staged
// in LDS (rig_compile builds per-tile joint lists and rewrites the vertex indices to list
// positions); morph deltas of the (tile, target) pairs with a non-zero weight reach the waves
// through LDS slots.  Three kernels share the arithmetic below:
//   k_skin2     tiles with <= 64 joints, <= 2 weight sets: four self-staging waves per block
//   k_skin_fix  the (tile, instance) pairs k_skin2 hands over (3+ active pairs; every instance of the few tiles whose long
//               joint lists would size k_skin2's palette ring)
//   k_skin      everything else: four compute waves + one loader wave
//
// Arithmetic order is the WGSL's (renderer/src/shader/shader.wgsl:127-141, :159-163):
//   S  = ((w0*M[j0] + w1*M[j1]) + w2*M[j2]) + w3*M[j3]   (further weight sets continue the chain)
//   p' = ((S.c0*x + S.c1*y) + S.c2*z) + S.c3             n' = (N.c0*x + N.c1*y) + N.c2*z
// EXACT=true: no FMA contraction -> bit-equal to the fp32 oracle.  EXACT=false (default): the
// same association with fused multiply-adds; within 1e-5 rel / 1e-6 abs.  Rigs far from the origin get the
// third form by default (AR = 2): positions contraction-free, normals / tangents fused.
#include "device_types.hpp"

namespace skinrt {

namespace {

#ifdef SKINRT_STAMPS
// Diagnostic build only (make stamps): per-segment cycle sums over all waves; never in the product.
__device__ unsigned long long g_stamps[16];
#define STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_[k] += now_ - last_; last_ = now_; } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// Diagnostic builds only (-DSKINRT_ABLATE=bits): 1 = skip the morph arithmetic, 2 = skip the morph
// LDS-DMA, 4 = no instance has an active pair, 8 = dense deltas from registers (no LDS read),
// 16 = dense deltas read but not applied, 32 = k_skin_fix is not launched.  -DSKINRT_FIX_ABLATE=bits:
// 1 = k_skin_fix skips its delta loads, 2 = its stores.  Results are wrong by design; timing experiments.
#ifndef SKINRT_ABLATE
#define SKINRT_ABLATE 0
#endif
// influences whose palette rows are in flight together (a scheduling fence after every SKINRT_INFLIGHT-th; 1 / 3 / 4 / 8 measured: within noise)
#ifndef SKINRT_INFLIGHT
#define SKINRT_INFLIGHT 2
#endif
#ifndef SKINRT_BLOCKS_K1
#define SKINRT_BLOCKS_K1 5  // resident k_skin2 blocks per CU the register budget is set for, one weight set (measured: 6 blocks, which
                            // need blocks of <= 64 instances and a one-instance lookahead to fit LDS: C3 4.43-4.54 ms against 4.38)
#endif
// Timing experiments (results wrong): 2 = influence 0 takes its rows from registers (no LDS read),
// 8 = wave w of a block visits only (w + 1) * K influences (what sorting a tile's vertices by influence count would buy),
// 16 = one influence per vertex, 1024 = k_skin2 requests no palette rows (stale LDS): C3 -0.25 ms, C5 -0.4 ms — what the
// palette's way into LDS costs, four copies per tile and instance (one per wave).  (A further switch sent k_skin2's outputs through an LDS transposition and two block
// barriers per instance, which sorted tiles would need for coalesced stores: +0.9 ms on C5, +0.5 ms on C3 — as much as
// switch 8 saves; sorted tiles were not built.  tools/ubench/store_perm.hip: scattered 12-byte stores are 7x slower.
// Another emulated a third morph pair per instance, fetched into registers by plain loads issued one instance ahead (before
// the previous instance's last two stores) and waited for with vmcnt(3), for 1 in 23 instances — what taking C3's 3-pair
// instances off k_skin_fix would cost k_skin2: +0.12 ms, plus their own 0.17 ms of work, against the 0.37 ms k_skin_fix
// takes for them.  Not built either.)
#ifndef SKINRT_ORDER_ROT
#define SKINRT_ORDER_ROT 23u  // k_skin2: a tile's instance order is rotated by tile * this (see the prologue; 0 / 1 / 7 / 13 / 23 measured)
#endif
#ifndef SKINRT_EXP
#define SKINRT_EXP 0
#endif
// 1 (the product): on rigs that have compacted pairs, k_skin2 takes them from its LDS slots too (its SPARSE
// instantiations); 0: it hands every instance with an active compacted pair to k_skin_fix, as until round 2 (A/B
// builds; same results).  Rigs whose pairs are all dense run the dense-only instantiations either way: carrying the
// per-pair layout flag through the instance loop costs them 2.5 % (C3) to 6 % (C5) of the stage (profiles/r03_ubench.txt).
#ifndef SKINRT_SPARSE_SLOTS
#define SKINRT_SPARSE_SLOTS 1
#endif
// 0 (the product): the skin kernels walk a crowd's instances from the last range to the first, so that the palettes the
// pose stage wrote most recently (still in the memory-side cache) are the first ones read; bit 0: k_skin2 / k_skin
// ascending, bit 1: k_skin_fix walks its lists upwards (A/B builds; same results).  See k_skin2's block mapping.
#ifndef SKINRT_RANGE_ORDER
#define SKINRT_RANGE_ORDER 0
#endif

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

// Barrier that orders LDS only.  __syncthreads() also fences global memory, i.e. waits for every
// outstanding store of the wave (s_waitcnt vmcnt(0)) — here that would drain the 36 B/vertex
// output stream at every batch and serialise the store path with the math.  Written as asm so the
// compiler cannot widen it to vmcnt(0) on account of the LDS-DMA loads it sees in this kernel.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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
// Scalar v_fma_f32 on purpose (the file is built with -fno-slp-vectorize).  Measured twice: v_pk_fma_f32 with the
// weight broadcast by op_sel (no {w, w} pair) needs aligned register pairs for rows and sums — 74 -> 95 VGPRs (K=1),
// 85 -> 119 (K=2) — and the skin stage got slower: C3 4.68 -> 4.76 ms, C5 10.7 -> 11.5 ms (exact: 12.3 -> 13.5).
// Influences a wave has to visit: weight sets beyond the first are mostly zero on real meshes (JOINTS_1 / WEIGHTS_1
// exist for a few regions), and a trailing influence whose weight is zero in all 64 lanes adds w * M = 0 to every
// sum (a sign of zero aside, palette entries being finite), so the wave stops at its last non-zero one: no LDS reads,
// no arithmetic for the rest.  Wave-uniform (SGPR) and fixed for the block's life.  The first set is always visited.
// Instantiated (SKIP) only for rigs where it pays — RigDev::skip_sets, decided by the rig compiler: the branch per set
// keeps the compiler from interleaving the sets' LDS reads and arithmetic, which costs 1-3 % where nothing can be
// skipped (measured on C5, whose synthetic weights are non-zero up to the eighth in every wave).
template <int K>
__device__ __forceinline__ uint32_t wave_influences(const float (&w)[4 * K]) {
    uint32_t n = 4;
#pragma unroll
    for (int a = 4; a < 4 * K; ++a)
        if (__ballot(w[a] != 0.0f) != 0ull) n = (a | 3) + 1;  // whole sets: one branch per set keeps each set's schedule intact
    return n;
}

template <bool EXACT, int K, bool SKIP, class Pal>
__device__ __forceinline__ void blend_position(const Pal pal, const float (&w)[4 * K], const uint32_t (&li)[4 * K], uint32_t ninf,
                                               float (&S)[12]) {  // rows of the blended 3x4
#pragma unroll
    for (int a = 0; a < 4 * K; ++a) {
        if (SKIP && a >= 4 && (a & 3) == 0 && (uint32_t)a >= ninf) break;  // wave-uniform, tested once per weight set
        if (((SKINRT_EXP & 8) && a > 0 && (uint32_t)a >= ninf) || ((SKINRT_EXP & 16) && a > 0)) break;
        const f4* e = pal.entry(li[a]);
        f4 r0, r1, r2;
        if ((SKINRT_EXP & 2) && a == 0) { r0 = f4{w[0], w[1], w[2], w[3]}; r1 = r0 * 2.0f; r2 = r0 * 3.0f; }
        else { r0 = e[0]; r1 = e[1]; r2 = e[2]; }
        const float m[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
#pragma unroll
        for (int x = 0; x < 12; ++x) S[x] = a == 0 ? mul_nc(w[0], m[x]) : madd<EXACT>(w[a], m[x], S[x]);
        // two influences' rows in flight at a time: bounds the live registers (5 waves per SIMD)
        if ((a + 1) % SKINRT_INFLIGHT == 0 && a + 1 < 4 * K) __builtin_amdgcn_sched_barrier(0);
    }
}
template <bool EXACT>
__device__ __forceinline__ void apply_position(const float (&S)[12], const float (&p)[3], float (&op)[3]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float s = mul_nc(S[4 * r], p[0]);
        s = madd<EXACT>(S[4 * r + 1], p[1], s);
        s = madd<EXACT>(S[4 * r + 2], p[2], s);
        op[r] = add_nc(s, S[4 * r + 3]);
    }
}

template <bool EXACT, int K, bool SKIP, class Pal>
__device__ __forceinline__ void blend_normal(const Pal pal, const float (&w)[4 * K], const uint32_t (&li)[4 * K], uint32_t ninf,
                                             float (&Nm)[9]) {  // rows of the blended normal matrix
#pragma unroll
    for (int a = 0; a < 4 * K; ++a) {
        if (SKIP && a >= 4 && (a & 3) == 0 && (uint32_t)a >= ninf) break;  // wave-uniform, tested once per weight set
        if (((SKINRT_EXP & 8) && a > 0 && (uint32_t)a >= ninf) || ((SKINRT_EXP & 16) && a > 0)) break;
        const f4* e = pal.entry(li[a]);
        // the nine normal-matrix entries are packed (kernels_pose.hip: palette_entry): 2 x ds_read_b128 + ds_read_b32.
        // An LDS read costs by the registers it fills (tools/ubench/lds_exec.hip), so this is 84 bytes per influence
        // instead of the 96 of three padded rows (C5, which is bound by these reads: skin stage -0.4 ms).
        f4 n0, n1;
        float n22;
        if ((SKINRT_EXP & 2) && a == 0) { n0 = f4{w[0], w[1], w[2], w[3]}; n1 = n0 * 2.0f; n22 = w[0]; }
        else { n0 = e[3]; n1 = e[4]; n22 = reinterpret_cast<const float*>(e + 5)[0]; }
        const float q[9] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n22};
#pragma unroll
        for (int x = 0; x < 9; ++x) Nm[x] = a == 0 ? mul_nc(w[0], q[x]) : madd<EXACT>(w[a], q[x], Nm[x]);
        if ((a + 1) % SKINRT_INFLIGHT == 0 && a + 1 < 4 * K) __builtin_amdgcn_sched_barrier(0);
    }
}
template <bool EXACT>
__device__ __forceinline__ void apply_normal(const float (&Nm)[9], const float (&n)[3], const float (&t)[3], bool normalize,
                                             float (&on)[3], float (&ot)[3]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float a = mul_nc(Nm[3 * r], n[0]);
        a = madd<EXACT>(Nm[3 * r + 1], n[1], a);
        on[r] = madd<EXACT>(Nm[3 * r + 2], n[2], a);
        float b = mul_nc(Nm[3 * r], t[0]);
        b = madd<EXACT>(Nm[3 * r + 1], t[1], b);
        ot[r] = madd<EXACT>(Nm[3 * r + 2], t[2], b);
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

// One (tile, target) pair applied to this lane's vertex.
//   dense : `src` = this WAVE's share [9][64] (LDS slot, or mblock + wave * 576)
//   sparse: `src` = the whole region [256 B slot map][cnt][9]
template <bool EXACT>
__device__ __forceinline__ void morph_pair(const float* src, bool dense, uint32_t tid, float mw,
                                           float (&p)[3], float (&n)[3], float (&t)[3]) {
    if (dense) {
        float d[9];
#pragma unroll
        for (int x = 0; x < 9; ++x) d[x] = (SKINRT_ABLATE & 8) ? p[x % 3] : src[x * 64 + (tid & 63)];
        if (SKINRT_ABLATE & 16) {
#pragma unroll
            for (int x = 0; x < 9; ++x) asm volatile("" ::"v"(d[x]));
        } else {
            morph_add<EXACT>(mw, d, p, n, t);
        }
    } else {
        const uint32_t slot = reinterpret_cast<const uint8_t*>(src)[tid];
        if (slot) {
            const float* dp = src + kTileVerts / 4 + (slot - 1) * 9;
            float d[9];
#pragma unroll
            for (int x = 0; x < 9; ++x) d[x] = dp[x];
            morph_add<EXACT>(mw, d, p, n, t);
        }
    }
}

// The load half of morph_pair: this lane's nine deltas of one pair; false = the pair does not touch the lane's vertex
// (compacted pairs only: a vertex outside the target's index list is not added to at all, not even a zero).
__device__ __forceinline__ bool morph_load(const float* src, bool dense, uint32_t tid, float (&d)[9]) {
    if (dense) {
#pragma unroll
        for (int x = 0; x < 9; ++x) d[x] = src[x * 64 + (tid & 63)];
        return true;
    }
    const uint32_t slot = reinterpret_cast<const uint8_t*>(src)[tid];
#pragma unroll
    for (int x = 0; x < 9; ++x) d[x] = 0.0f;
    if (slot) {
        const float* dp = src + kTileVerts / 4 + (slot - 1) * 9;
#pragma unroll
        for (int x = 0; x < 9; ++x) d[x] = dp[x];
    }
    return slot != 0;
}

// 16 bytes per lane, global (uniform base + lane byte offset + IMM) -> LDS (lds_base + IMM + lane*16),
// with no register destination.  Written as asm: the compiler's wait-count pass treats an LDS-DMA
// it can see as aliasing every later LDS read and answers with s_waitcnt vmcnt(0), which here would
// also wait for the wave's output stores.  All waits for these loads are placed by hand.  m0 is
// written here and named as a clobber, so the compiler never keeps a value of its own live in it
// across one of these statements (clang warns that m0 is a reserved register: silenced below, the
// statement really does overwrite it).
#pragma clang diagnostic ignored "-Winline-asm"
#define SKINRT_GLDS(IMM)                                                                                  \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:" #IMM             \
                 : : "v"(lane_bytes), "s"(gbase), "s"(lds_base) : "memory", "m0")

// Start the LDS-DMA of up to kMorphSlots active pairs of one instance into this wave's slots.
// act: the instance's records {mblock float offset, weight bits, dense, region floats}.
__device__ __forceinline__ void morph_dma_issue(const float* mblock, const uint4* act, uint32_t na, float* slots,
                                                uint32_t wave, uint32_t lane) {
    const uint32_t ns = (SKINRT_ABLATE & 2) ? 0u : min(na, kMorphSlots);
    const uint32_t lane_bytes = lane * 16;
    for (uint32_t a = 0; a < ns; ++a) {
        const uint4 rc = act[1 + a];
        const uint32_t doff = __builtin_amdgcn_readfirstlane(rc.x);
        const bool dense = __builtin_amdgcn_readfirstlane(rc.z) != 0;
        const uint32_t n4 = dense ? kMorphWaveFloats / 4 : (__builtin_amdgcn_readfirstlane(rc.w) + 3) / 4;
        const float* gbase = mblock + (doff + (dense ? wave * kMorphWaveFloats : 0u));  // wave-uniform
        const uint32_t lds_base = __builtin_amdgcn_readfirstlane(
            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)(slots + a * kMorphWaveFloats));
        if (lane < n4) SKINRT_GLDS(0);
        if (lane + 64 < n4) SKINRT_GLDS(1024);
        if (lane + 128 < n4) SKINRT_GLDS(2048);
    }
}

typedef float float3v __attribute__((ext_vector_type(3)));

// 12-byte store to (wave-uniform base) + (per-lane byte offset).  The s_nop covers the wait state a
// >8-byte store needs before its data registers may be overwritten (the compiler's hazard pass does
// not look inside asm).
__device__ __forceinline__ void store3(float* base, uint32_t byte_off, const float (&x)[3]) {
    float3v d = {x[0], x[1], x[2]};
#ifndef SKINRT_STORE_POLICY
#define SKINRT_STORE_POLICY 1  // nt: the outputs are a pure stream, keep L2 for palettes / morph deltas (measured: -7 %)
#endif
#if SKINRT_STORE_POLICY == 1
#define SKINRT_STORE_MOD " nt"
#elif SKINRT_STORE_POLICY == 2
#define SKINRT_STORE_MOD " sc1"
#elif SKINRT_STORE_POLICY == 3
#define SKINRT_STORE_MOD " sc0 sc1 nt"
#else
#define SKINRT_STORE_MOD ""
#endif
    asm volatile("global_store_dwordx3 %0, %1, %2" SKINRT_STORE_MOD "\n\ts_nop 1" : : "v"(byte_off), "v"(d), "s"(base) : "memory");
}

// the same without the streaming hint: small scattered records (bounds) that L2 should merge
__device__ __forceinline__ void store3_cached(float* base, uint32_t byte_off, const float (&x)[3]) {
    float3v d = {x[0], x[1], x[2]};
    asm volatile("global_store_dwordx3 %0, %1, %2\n\ts_nop 1" : : "v"(byte_off), "v"(d), "s"(base) : "memory");
}

struct Outputs {
    float* pos;
    float* nrm;
    float* tan;
};

// One (instance, vertex): skin from (p, n, t), store, feed the bounds.
template <bool EXACT, bool NEXACT, int K, bool SKIP, class Pal>
__device__ __forceinline__ void skin_store(const Pal pal, const float (&w)[4 * K], const uint32_t (&li)[4 * K], uint32_t ninf,
                                           const float (&p)[3], const float (&n)[3], const float (&t)[3],
                                           bool normalize, bool valid, const Outputs& out, size_t obase, uint32_t olb, float (&op)[3]) {
    // obase (instance offset) is wave-uniform and olb (vertex byte offset) fits 32 bits: the three
    // stores take the SGPR-base + VGPR-offset form and share one offset register.  Left to itself
    // the compiler hoists three 64-bit per-lane pointers out of the instance loop and spills them.
    float on[3], ot[3];
    {
        float S[12];
        blend_position<EXACT, K, SKIP>(pal, w, li, ninf, S);
        apply_position<EXACT>(S, p, op);
    }
    if (valid) store3(out.pos + obase, olb, op);
    {
        float Nm[9];
        blend_normal<NEXACT, K, SKIP>(pal, w, li, ninf, Nm);
        apply_normal<NEXACT>(Nm, n, t, normalize, on, ot);
    }
    if (valid) {
        store3(out.nrm + obase, olb, on);
        store3(out.tan + obase, olb, ot);
    }
}

constexpr uint32_t kSkinThreads = 320;  // 4 compute waves (256 vertices) + 1 loader wave

// Producer / consumer block.  Waves 0-3 own the tile's 256 vertices: inside the instance loop they
// only read LDS, do math and issue stores — never a vector-memory LOAD, whose wait (vmcnt counts
// loads and stores together, in order) would drain the wave's whole output stream.  Wave 4 is the
// loader: while the others work on batch k it stages batch k+1 (palette rows of the tile's joints,
// active morph-pair records) into the other LDS buffer and writes batch k-1's bounds.  One
// LDS-only barrier per batch hands the buffers over.
template <bool EXACT, int K, bool LDSPAL>
__global__ __launch_bounds__(kSkinThreads, (K == 1 ? 5 : (K == 2 ? 4 : 2))) void k_skin(const RigDev rig, const CrowdDev cr, const SkinPlan plan,
                                                                                        const uint32_t first, const uint32_t ntiles, const uint32_t tpc) {
    extern __shared__ float4 smem[];
    constexpr uint32_t TV = kTileVerts;
    constexpr int W = 4 * K;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave == 4;
    // block -> (tile, range): blocks b and b+8 share an XCD (round-robin dispatch); each XCD owns
    // a contiguous chunk of tiles x a chunk of ranges, so a tile's mesh/morph data and a range's
    // palette rows stay in ONE L2.  Placement is a speed matter only.
    const uint32_t xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const uint32_t tc = xcd % plan.tile_chunks, rc = xcd / plan.tile_chunks;
    // (the grid covers this launch's OWN tiles: rig.tile_order[first .. first + ntiles), tpc of them per tile chunk)
    const uint32_t tile_pos = tc * tpc + k % tpc;
    uint32_t rng = rc * plan.ranges_per_chunk + k / tpc;
    if (tile_pos >= ntiles || rng >= plan.ranges) return;
    const uint32_t tile = rig.tile_order[first + tile_pos];
    if (!(SKINRT_RANGE_ORDER & 1)) rng = plan.ranges - 1 - rng;  // last range first: see k_skin2
    // tiles whose joint list fits LDS are skinned by the LDSPAL=true launch, the rest (rare: no
    // joint locality) by the LDSPAL=false launch that gathers the palette from L2
    {
        const uint32_t kind = rig.tile_lds[tile];  // 2: short joint list (k_skin2's tiles unless switched off)
        if (LDSPAL ? !(kind == 1 || (kind >= 2 && !plan.v2)) : kind != 0) return;  // (3: kind 2 whose instances k_skin_fix takes)
    }
    const uint32_t i0 = rng * plan.range;
    const uint32_t i1 = min(cr.I, i0 + plan.range);
    const uint32_t JTP = 1u << plan.jt_shift;
    const uint32_t B = plan.batch;
    const uint32_t M = rig.M;
    const uint32_t P1 = plan.max_pairs + 1;
    // LDS: two buffers of per-batch data, then the block-lifetime morph tables
    f4* pal_s = reinterpret_cast<f4*>(smem);                              // [2][B][JTP][6]
    float* ab_s = reinterpret_cast<float*>(pal_s + (size_t)2 * B * JTP * 6);  // [2][B][16][6]
    // per instance: [0].x = number of active pairs, then one record per active pair:
    // {mblock float offset, weight bits, dense flag, region floats}
    uint4* act_s = reinterpret_cast<uint4*>(ab_s + (size_t)2 * B * 16 * 6);   // [2][B][1 + max_pairs]
    MorphPair* pairs_s = reinterpret_cast<MorphPair*>(act_s + (size_t)2 * B * P1);  // [max_pairs]
    float* ml_s = reinterpret_cast<float*>(pairs_s + plan.max_pairs);     // [4 waves][kMorphSlots][576]
    // processing order of the block's instances (see the loader prologue): active-pair mask per
    // instance-in-range, and the order itself
    uint32_t* mask_s = reinterpret_cast<uint32_t*>(ml_s + (rig.M ? 4 * kMorphSlots * kMorphWaveFloats : 0u));  // [256]
    uint16_t* perm_s = reinterpret_cast<uint16_t*>(mask_s + 256);                    // [256]
    const uint32_t pal_buf = B * JTP * 6, ab_buf = B * 16 * 6, act_buf = B * P1;

    const uint32_t pbeg = rig.tile_pair_start[tile];
    const uint32_t np = rig.tile_pair_start[tile + 1] - pbeg;
    const uint32_t jstart = rig.tile_joint_start[tile];
    const uint32_t jcnt = rig.tile_joint_start[tile + 1] - jstart;
    const f4* pal_g = reinterpret_cast<const f4*>(cr.palette);
    const uint32_t nbatch = (i1 - i0 + B - 1) / B;

    // ---- block prologue: morph tables into LDS (all waves), this thread's vertex (waves 0-3) --
    for (uint32_t c = tid; c < np; c += kSkinThreads) pairs_s[c] = rig.pairs[pbeg + c];
    lds_barrier();  // pairs_s visible to every wave

    if (loader) {
        // ================= loader wave =================
        // Its registers never coexist with the compute waves' (the roles never rejoin), which keeps
        // the whole kernel at 96 VGPRs = 5 waves per SIMD.
        // lane state that does not change from batch to batch: its palette piece (joint, quarter)
        // and, when the tile has at most 64 pairs, its (instance-in-group, pair) slot
        const uint32_t c0 = lane;
        const uint32_t lj0 = c0 / 6, q0 = c0 - lj0 * 6;
        const uint32_t j0 = (LDSPAL && c0 < jcnt * 6) ? rig.tile_joints[jstart + lj0] : 0u;
        const uint32_t G = 1u << plan.pair_gshift;
        const uint32_t gq = lane & (G - 1), gb = lane >> plan.pair_gshift;
        const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
        MorphPair mymp{};
        if (gq < np && np <= 64) mymp = pairs_s[gq];

        // ---- processing order: instances with the same SET of active pairs become neighbours ----
        // Outputs go to each instance's own rows, so the order inside a block is free.  Sorted by
        // active-pair mask, a run of instances finds its pairs already in the waves' LDS slots and
        // only the first of the run streams them in (C3: ~9 distinct sets per 256 instances).
        // mask bit q = pair q active (pairs >= 31 share bit 31, which is never trusted as "same").
        const uint32_t nin = i1 - i0;  // <= 256 (plan.range)
        if (np == 0) {
            for (uint32_t kk = lane; kk < 256; kk += 64) { mask_s[kk] = 0; perm_s[kk] = (uint16_t)kk; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            uint32_t mym[4], rank[4] = {0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t kk = lane + 64 * r;
                uint32_t m = 0xffffffffu;  // padding sorts last
                if (kk < nin) {
                    m = 0;
                    const float* mwg = cr.mw + (size_t)(i0 + kk) * M;
                    for (uint32_t q = 0; q < np; ++q)
                        if (mwg[pairs_s[q].target] != 0.0f) m |= 1u << min(q, 31u);
                }
                mym[r] = m;
                mask_s[kk] = m;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave: its own LDS writes are now readable
            // rank = number of (mask, index) keys below mine; keys are distinct, so ranks are a permutation
            for (uint32_t j = 0; j < 256; j += 4) {
                const uint4 mj = *reinterpret_cast<const uint4*>(mask_s + j);
                const uint32_t mjs[4] = {mj.x, mj.y, mj.z, mj.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t kk = lane + 64 * r;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        rank[r] += (mjs[u] < mym[r] || (mjs[u] == mym[r] && j + u < kk)) ? 1u : 0u;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t kk = lane + 64 * r;
                perm_s[rank[r]] = (uint16_t)kk;  // padded keys take the ranks >= nin
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        const bool exact_mask = np <= 32;
        auto inst_at = [&](uint32_t pos) -> uint32_t { return i0 + perm_s[pos]; };
        // header of the instance at `pos`: {active pairs | reload << 16, instance, 0, 0}; reload = its
        // active set differs from its predecessor's in the block's order (slots must be refilled)
        auto header = [&](uint32_t pos, uint32_t count) -> uint4 {
            const uint32_t m = mask_s[perm_s[pos]];
            const uint32_t prev = pos ? mask_s[perm_s[pos - 1]] : ~m;
            const uint32_t reload = (m != prev || (!exact_mask && (m >> 31))) ? 1u : 0u;
            if (SKINRT_ABLATE & 4) count = 0;
            return make_uint4(count | (reload << 16), inst_at(pos), 0, 0);
        };

        // stage batch `kb` into LDS buffer `buf`
        auto stage = [&](uint32_t kb, uint32_t buf) {
            const uint32_t pos0 = kb * B;
            const uint32_t nb = min(B, nin - pos0);
            uint4* actb = act_s + buf * act_buf;
            // (1) morph weight of this lane's (instance, pair) slot: issue the load first
            float mw_first = 0.0f;
            if (np <= 64 && gb < nb && gq < np) mw_first = cr.mw[(size_t)inst_at(pos0 + gb) * M + mymp.target];
            // (2) palette rows of the tile's joints: a lane owns one 16-byte piece (joint, quarter)
            // and walks the batch's instances, four loads in flight at a time
            if (LDSPAL) {
                const size_t istride = (size_t)rig.J * 6;
                const uint32_t dstride = JTP * 6;
                for (uint32_t c = lane; c < jcnt * 6; c += 64) {
                    uint32_t j = j0, q = q0;
                    if (c != c0) { const uint32_t lj = c / 6; q = c - lj * 6; j = rig.tile_joints[jstart + lj]; }
                    const f4* src = pal_g + (size_t)j * 6 + q;
                    f4* dst = pal_s + buf * pal_buf + c;
                    uint32_t b = 0;
                    for (; b + 4 <= nb; b += 4) {
                        const uint32_t ia = inst_at(pos0 + b), ib1 = inst_at(pos0 + b + 1), ic = inst_at(pos0 + b + 2), id = inst_at(pos0 + b + 3);
                        const f4 x0 = src[(size_t)ia * istride], x1 = src[(size_t)ib1 * istride];
                        const f4 x2 = src[(size_t)ic * istride], x3 = src[(size_t)id * istride];
                        dst[b * dstride] = x0; dst[(b + 1) * dstride] = x1;
                        dst[(b + 2) * dstride] = x2; dst[(b + 3) * dstride] = x3;
                    }
                    for (; b < nb; ++b) dst[b * dstride] = src[(size_t)inst_at(pos0 + b) * istride];
                }
            }
            // (3) the tile's pairs whose weight is non-zero for the instance (zero weights add
            // exactly 0), compacted in target order with a wave ballot.  Up to 64 pairs: groups of
            // G lanes take one instance each, so one round of loads covers 64/G instances.
            if (np <= 64) {
                for (uint32_t b0 = 0; b0 < nb; b0 += 64 >> plan.pair_gshift) {
                    const uint32_t b = b0 + gb;
                    const bool live = b < nb && gq < np;
                    const float mw = b0 == 0 ? mw_first : (live ? cr.mw[(size_t)inst_at(pos0 + b) * M + mymp.target] : 0.0f);
                    const bool on = mw != 0.0f;
                    const unsigned long long seg = (__ballot(on) >> (gb << plan.pair_gshift)) & gmask;
                    if (on) actb[b * P1 + 1 + __popcll(seg & ((1ull << gq) - 1ull))] =
                                make_uint4(mymp.data_off, __float_as_uint(mw), mymp.dense, mymp.floats);
                    if (gq == 0 && b < nb) actb[b * P1] = header(pos0 + b, (uint32_t)__popcll(seg));
                }
            } else {
                for (uint32_t b = 0; b < nb; ++b) {
                    uint4* act = actb + b * P1;
                    const float* mwg = cr.mw + (size_t)inst_at(pos0 + b) * M;
                    uint32_t n = 0;
                    for (uint32_t qq = 0; qq < np; qq += 64) {
                        const uint32_t q = qq + lane;
                        MorphPair mp{};
                        float mw = 0.0f;
                        if (q < np) { mp = pairs_s[q]; mw = mwg[mp.target]; }
                        const bool on = mw != 0.0f;
                        const unsigned long long mask = __ballot(on);
                        if (on) act[1 + n + __popcll(mask & ((1ull << lane) - 1ull))] =
                                    make_uint4(mp.data_off, __float_as_uint(mw), mp.dense, mp.floats);
                        n += __popcll(mask);
                    }
                    if (lane == 0) act[0] = header(pos0 + b, n);
                }
            }
        };
        // bounds of batch `kb` (16 DPP-row partials per instance, written by the compute waves)
        auto flush_bounds = [&](uint32_t kb, uint32_t buf) {
            const uint32_t pos0 = kb * B;
            const uint32_t nb = min(B, nin - pos0);
            for (uint32_t idx = lane; idx < nb * 6; idx += 64) {
                const uint32_t b = idx / 6, c = idx - b * 6;
                const float* src = ab_s + buf * ab_buf + (size_t)b * 16 * 6 + c;
                float r = src[0];
                for (int q = 1; q < 16; ++q) r = c < 3 ? fminf(r, src[q * 6]) : fmaxf(r, src[q * 6]);
                cr.aabb_part[((size_t)inst_at(pos0 + b) * rig.T + tile) * 6 + c] = r;
            }
        };

        stage(0, 0);
        lds_barrier();
        for (uint32_t kb = 0; kb < nbatch; ++kb) {
            const uint32_t buf = kb & 1;
            if (kb + 1 < nbatch) stage(kb + 1, buf ^ 1);
            if (plan.aabb && kb > 0) flush_bounds(kb - 1, buf ^ 1);
            lds_barrier();
        }
        if (plan.aabb) flush_bounds(nbatch - 1, (nbatch - 1) & 1);
        return;
    }

    // ================= compute waves: this thread's vertex =================
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
    constexpr bool SKIP = false;  // (the fallback kernel visits every influence)
    const uint32_t ninf = 4u * K;
    const Outputs out{cr.pos, cr.nrm, cr.tan};
    const uint32_t olb = v * 12;  // byte offset of this lane's vertex inside an instance's output
    const bool normalize = plan.normalize != 0;
    float* my_slots = ml_s + (size_t)wave * kMorphSlots * kMorphWaveFloats;
    // vector-memory ops this wave issues per instance after the morph DMA: 3 stores, or none when no
    // lane of the wave holds a real vertex (tail of the last tile)
    const uint32_t stores_per_instance = __builtin_amdgcn_readfirstlane(__ballot(valid) != 0ull ? 3u : 0u);
    lds_barrier();  // batch 0 is staged

#ifdef SKINRT_STAMPS
    unsigned long long acc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
    for (uint32_t kb = 0; kb < nbatch; ++kb) {
        const uint32_t buf = kb & 1;
        const uint32_t nb = min(B, (i1 - i0) - kb * B);
        // header {active pairs | reload << 16, instance}: the loader orders the block's instances so
        // that equal active sets are neighbours; reload = the slots hold another set.  Kept in
        // SGPRs; the next one is fetched at the top of an instance and made scalar after its morph.
        uint32_t hx, hi;
        {
            const uint2 h0 = *reinterpret_cast<const uint2*>(act_s + buf * act_buf);
            hx = __builtin_amdgcn_readfirstlane(h0.x);
            hi = __builtin_amdgcn_readfirstlane(h0.y);
        }
        bool dma_pending = false;  // this instance's pairs were requested during the previous one
        for (uint32_t b = 0; b < nb; ++b) {
            const uint4* act = act_s + buf * act_buf + b * P1;
            const uint32_t i = hi;
            const uint32_t na = hx & 0xffffu;
            const bool reload = (hx >> 16) != 0;
            const size_t obase = (size_t)i * rig.V * 3;
            uint2 hdr_next = make_uint2(0, 0);
            if (b + 1 < nb) hdr_next = *reinterpret_cast<const uint2*>(act + P1);  // arrives during the morph
            STAMP(0);
            float op[3];
            auto skin = [&](const float (&p)[3], const float (&n)[3], const float (&t)[3]) {
                if constexpr (LDSPAL) skin_store<EXACT, EXACT, K, SKIP>(LdsPal{pal_s + buf * pal_buf + (size_t)b * JTP * 6}, w, li, ninf, p, n, t, normalize, valid, out, obase, olb, op);
                else skin_store<EXACT, EXACT, K, SKIP>(GlobalPal{pal_g + (size_t)i * rig.J * 6}, w, li, ninf, p, n, t, normalize, valid, out, obase, olb, op);
            };
            // Morph data reaches this wave through LDS-DMA requested one instance ahead, so the wait
            // below is "all but my 3 newest vector-memory ops" = the previous instance's stores may
            // still be in flight.  A plain load here would have to wait for them (vmcnt is in order).
            float p[3] = {p0[0], p0[1], p0[2]}, n[3] = {n0[0], n0[1], n0[2]}, t[3] = {t0[0], t0[1], t0[2]};
            if (na) {
                // records of the two slot pairs, requested before the DMA wait so that their LDS
                // latency overlaps it (act[2] is readable LDS even when na == 1; unused then)
                const uint2 r1 = *reinterpret_cast<const uint2*>(&act[1].y);  // {weight bits, dense}
                const uint2 r2 = *reinterpret_cast<const uint2*>(&act[2].y);
                if (reload) {
                    if (!dma_pending) {  // first instance of a batch: request now, wait for everything
                        morph_dma_issue(rig.mblock, act, na, my_slots, wave, lane);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    } else if (stores_per_instance == 3) {
                        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
                STAMP(1);
                // a13: p = p0 + sum_m w_m * delta_m, targets ascending (wave-uniform loop)
                // wave-uniform: weights and flags move to SGPRs
                const float mw1 = __uint_as_float(__builtin_amdgcn_readfirstlane(r1.x));
                const float mw2 = __uint_as_float(__builtin_amdgcn_readfirstlane(r2.x));
                const bool dense1 = __builtin_amdgcn_readfirstlane(r1.y) != 0;
                const bool dense2 = __builtin_amdgcn_readfirstlane(r2.y) != 0;
                uint32_t a = 1;
                if (kMorphSlots >= 2 && !(SKINRT_ABLATE & 1)) {
                    // the two slot pairs: their records are already scalar, so each costs one LDS round trip
                    morph_pair<EXACT>(my_slots, dense1, tid, mw1, p, n, t);
                    if (na >= 2) morph_pair<EXACT>(my_slots + kMorphWaveFloats, dense2, tid, mw2, p, n, t);
                    a = 3;
                }
                for (; a <= ((SKINRT_ABLATE & 1) ? 0u : na); ++a) {
                    const uint4 cur = act[a];
                    const float mw = __uint_as_float(cur.y);
                    const bool dense = __builtin_amdgcn_readfirstlane(cur.z) != 0;
                    if (a <= kMorphSlots) morph_pair<EXACT>(my_slots + (a - 1) * kMorphWaveFloats, dense, tid, mw, p, n, t);
                    else  // more active pairs than slots: plain loads (rare; these do drain the stores)
                        morph_pair<EXACT>(rig.mblock + __builtin_amdgcn_readfirstlane(cur.x) + (dense ? wave * kMorphWaveFloats : 0u),
                                          dense, tid, mw, p, n, t);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // slot reads retired before the slots are refilled
            }
            STAMP(2);
            // request the next instance's pairs before this instance's stores are issued
            dma_pending = false;
            hx = __builtin_amdgcn_readfirstlane(hdr_next.x);
            hi = __builtin_amdgcn_readfirstlane(hdr_next.y);
            if ((hx & 0xffffu) && (hx >> 16)) { morph_dma_issue(rig.mblock, act + P1, hx & 0xffffu, my_slots, wave, lane); dma_pending = true; }
            __builtin_amdgcn_sched_barrier(0);  // keep the 3 stores below younger than the DMA above
            skin(p, n, t);
            STAMP(3);
            if (plan.aabb) {
                // lanes past the mesh end repeat a real vertex (rig_compile), so no mask is needed
                float a0 = op[0], a1 = op[1], a2 = op[2], b0 = op[0], b1 = op[1], b2 = op[2];
                row_minmax6(a0, a1, a2, b0, b1, b2);
                if ((lane & 15) == 0) {
                    // lane id recomputed here (2 VALU ops) instead of kept live across the loop: at 96
                    // VGPRs the loop-invariant row index was the value the allocator spilled, and its
                    // scratch reload (a vector-memory load) drained the stores of every instance
                    uint32_t lid;
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lid));
                    float* dst = ab_s + buf * ab_buf + ((size_t)b * 16 + wave * 4 + (lid >> 4)) * 6;
                    dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = b0; dst[4] = b1; dst[5] = b2;
                }
            }
            STAMP(4);
        }
        lds_barrier();
        STAMP(5);
    }
#ifdef SKINRT_STAMPS
    if (lane == 0)
        for (int q = 0; q < 16; ++q) atomicAdd(&g_stamps[q], acc_[q]);
#endif
}


// =====================================================================================================
// k_skin2 — the self-staging variant.  No loader wave and no batch barrier: a block is 4 independent
// compute waves (64 vertices each) and every wave stages what it needs itself with LDS-DMA, which
// has no register destination and therefore needs no wait until the data is used:
//   * the palette rows of the tile's joints for instance pos+2 are requested at the top of
//     instance pos into the wave's own ring of 3 palette slots;
//   * the morph pairs of a run of instances with the same active set are requested at the first idle
//     instance of the spacer in front of the run (see the prologue), or, without a spacer, in the
//     middle of the previous instance (as in k_skin).
// All waits are COUNTED (vmcnt is in issue order, loads and stores alike): every instance issues at
// least `spi` stores, so "palette of this instance has landed" is s_waitcnt vmcnt(2 * spi), and the
// stores of the last two instances stay in flight.  With no loader the block is 256 threads, five
// blocks = 20 compute waves per CU instead of 16 (throughput follows the number of compute waves).
// Per-block tables (instance order, headers with the first two active pairs' weights, one bounds
// cell per processing position) are built once in the prologue, before any store exists.
// Instances that need more than the two slots go to k_skin_fix.  Tiles with more than 21 joints
// (their palette pieces do not fit two DMA instructions) and rigs with 3-4 weight sets stay on k_skin.

#ifndef SKINRT_PAL_MOD
#define SKINRT_PAL_MOD ""  // cache policy of the palette requests (" nt", " sc1", " sc0 sc1" measured: see profiles/r02_ubench.txt)
#endif
#define SKINRT_GLDS_AT(VOFF, GBASE, LDSB)                                                               \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" SKINRT_PAL_MOD           \
                 : : "v"(VOFF), "s"(GBASE), "s"(LDSB) : "memory", "m0")

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
// wait until at most `n` vector-memory operations of this wave are outstanding (n: wave-uniform)
__device__ __forceinline__ void wait_vmcnt_dyn(uint32_t n) {
    switch (n) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<1>(); break;
        case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 5: wait_vmcnt<5>(); break;
        case 6: wait_vmcnt<6>(); break;
        case 8: wait_vmcnt<8>(); break;
        case 9: wait_vmcnt<9>(); break;
        case 10: wait_vmcnt<10>(); break;
        case 12: wait_vmcnt<12>(); break;
        case 15: wait_vmcnt<15>(); break;
        case 16: wait_vmcnt<16>(); break;
        case 18: wait_vmcnt<18>(); break;
        case 20: wait_vmcnt<20>(); break;
        case 24: wait_vmcnt<24>(); break;
        case 30: wait_vmcnt<30>(); break;
        case 40: wait_vmcnt<40>(); break;
        default: wait_vmcnt<0>(); break;
    }
}

// min / max of six values over the whole wave, continuing row_minmax6: result in lane 63
__device__ __forceinline__ void wave_minmax6(float& a0, float& a1, float& a2, float& b0, float& b1, float& b2) {
    row_minmax6(a0, a1, a2, b0, b1, b2);
#define SKINRT_DPP_BCAST(CTRL, MASK)                                                    \
    asm volatile("s_nop 1\n\t"                                                          \
                 "v_min_f32_dpp %0, %0, %0 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t" \
                 "v_min_f32_dpp %1, %1, %1 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t" \
                 "v_min_f32_dpp %2, %2, %2 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t" \
                 "v_max_f32_dpp %3, %3, %3 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t" \
                 "v_max_f32_dpp %4, %4, %4 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t" \
                 "v_max_f32_dpp %5, %5, %5 " CTRL " row_mask:" MASK " bank_mask:0xf"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2))
    SKINRT_DPP_BCAST("row_bcast:15", "0xa");  // rows 1 and 3 take in lane 15 of rows 0 and 2
    SKINRT_DPP_BCAST("row_bcast:31", "0xc");  // rows 2 and 3 take in lane 31: lane 63 holds all four rows
#undef SKINRT_DPP_BCAST
}


// Which of a tile's (tile, target) pairs are active for one instance, and whether the self-staging kernel
// can morph it.  k_skin2 morphs from its two LDS slots per wave only: an instance with 1 or 2 active pairs — dense or
// compacted (round 3: a compacted pair's region fits a slot as well) — in a tile with at most 128 pairs.  Every other
// non-idle (tile, instance) is "slow": it goes on the tile's work list (k_classify) and is served by the fix role
// (fix_role).  k_classify and k_skin2's prologue both call this, so they agree on every (tile, instance).
// (Round 3 also built the alternative for 3 and more active pairs — k_skin2 morphing them itself IN LINE, two pairs at a
// time, each wave requesting only the pairs that move its own vertices and waiting for them: bit-identical results, and
// slower — C3 4.09 -> 4.58 ms, C5 10.6 -> 10.9 ms.  The wait is `vmcnt(0)`, i.e. for every store the wave has in
// flight, and in a kernel that keeps the HBM write queues full that is many microseconds, which the CU's other waves
// do not cover; k_skin_fix's waves, running beside k_skin2, have hardly any stores to wait for.  Removed again.)
// Tiles with 33 .. 128 pairs (round 3; a VRM face: a hundred blend shapes over a few thousand vertices): a 32-bit set
// cannot name their pairs, but an instance k_skin2 can take has at most two active ones, and two indices can: there
// `mask` is 0 for an idle instance and else  1 << 31 | second << 8 | first  (second = first for one active pair) —
// still "equal masks = equal sets", which is all the run sort asks of it.  Until then every non-idle instance of such a
// tile went on the work lists (tools/exp_face.py with 120 targets: an idle face — three active targets — 5.05 ms
// against 4.06 with 52 targets).
constexpr uint32_t kSkin2MaxPairs = 128;  // pairs per tile up to which k_skin2 morphs the instances with one or two active ones itself
struct PairClass {
    uint32_t mask;            // np <= 32: bit q = pair q has a non-zero weight; 33 .. 128 pairs: the encoding above
    bool slow;
    float w_first, w_second;  // weights of the first two active pairs (set for instances with at most two)
};
// `active` = the instance's row of CrowdDev::amask (bit t = morph weight t is non-zero), `words` = its length:
// the np tests read one or two cached 8-byte words instead of np floats scattered over the instance's weight row.
// sparse_ok: compacted pairs may sit in k_skin2's slots (RigDev::sparse_pairs: the rig has some and runs the SPARSE instantiations)
__device__ __forceinline__ PairClass classify_pairs(const MorphPair* pairs, uint32_t np, const uint64_t* active, uint32_t words, const float* mwg,
                                                    bool sparse_ok) {
    PairClass pc{0u, false, 0.0f, 0.0f};
    if (np == 0) return pc;  // (rigs without morph targets have no amask at all)
    const uint64_t a0 = active[0];
    if (np > 32) {  // (block-uniform: np is the tile's)
        uint32_t count = 0, q1 = 0, q2 = 0;
        bool compacted = false;
        for (uint32_t q = 0; q < np; ++q) {
            const MorphPair mp = pairs[q];
            const uint32_t t = mp.target;
            const uint64_t wd = words == 1 ? a0 : active[t >> 6];
            if ((wd >> (t & 63u)) & 1ull) {
                if (count == 0) q1 = q2 = q;
                else if (count == 1) q2 = q;
                ++count;
                compacted |= !sparse_ok && !mp.dense;
            }
        }
        if (count == 0 || (SKINRT_ABLATE & 4)) return pc;
        pc.mask = 0x80000000u | (q2 << 8) | q1;  // (an instance on the lists keeps a non-zero mask too; only its class is looked at)
        pc.slow = np > kSkin2MaxPairs || count > 2 || compacted;
        if (!pc.slow) {
            pc.w_first = mwg[pairs[q1].target];
            if (count == 2) pc.w_second = mwg[pairs[q2].target];
        }
        return pc;
    }
    uint32_t sparse = 0;
    for (uint32_t q = 0; q < np; ++q) {
        const MorphPair mp = pairs[q];
        const uint32_t t = mp.target;
        const uint64_t wd = words == 1 ? a0 : active[t >> 6];
        if ((wd >> (t & 63u)) & 1ull) pc.mask |= 1u << q;
        if (!sparse_ok && !mp.dense) sparse |= 1u << q;
    }
    pc.slow = pc.mask != 0 && (__popc(pc.mask) > 2 || (pc.mask & sparse) != 0);
    if (SKINRT_ABLATE & 4) { pc.slow = false; pc.mask = 0; }
    if (pc.mask != 0 && !pc.slow) {  // at most two active pairs: their weights go into the instance's header
        const uint32_t q1 = (uint32_t)__ffs((int)pc.mask) - 1;
        pc.w_first = mwg[pairs[q1].target];
        const uint32_t m2 = pc.mask & (pc.mask - 1);
        if (m2) pc.w_second = mwg[pairs[(uint32_t)__ffs((int)m2) - 1].target];
    }
    return pc;
}

// amask[i][w] bit t = morph weight 64 w + t of instance i is non-zero: one wave per (instance, word), lanes = targets.
__global__ __launch_bounds__(256) void k_active_mask(const CrowdDev cr, uint32_t M, uint32_t words) {
    const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= cr.I * words) return;  // wave-uniform
    const uint32_t i = g / words, w = g - i * words, t = w * 64 + lane;
    const float v = t < M ? cr.mw[(size_t)i * M + t] : 0.0f;
    const uint64_t b = __ballot(v != 0.0f);
    if (lane == 0) cr.amask[g] = b;
}

// Work lists of k_skin_fix, built before the skin kernels start so that both can run at the same time.
// Block = (tile, slice of kClassifyRounds * 256 instances).  A thread classifies one instance per round; all rounds'
// loads are in flight together (no barrier between rounds), the slow instances are compacted inside the block (ballot,
// per-wave totals in LDS) and the block reserves its entries with ONE atomicAdd on the tile's count.  History: one
// atomicAdd per wave (160 on the same counter for C3) took 0.10 ms; one block per tile walking all instances round by
// round, 0.06 ms (40 dependent rounds); this form is bound by nothing in particular.
// The count must be updated by an atomic even where a block is the tile's only writer: written with a plain store,
// counts were LOST on the GPU (C5: 2-76 tiles of an instance unserved, different ones every run).  Neighbouring
// tiles' counts share a 128-byte line, their blocks run on different XCDs, and the memset before this kernel has just
// written the whole line; a device-scope atomic is performed at the memory side and is not subject to whichever L2
// writes that line back last.  Tiles whose count the kernel does not touch keep the zero of the launch's memset.
constexpr uint32_t kClassifyRounds = 8;
__global__ __launch_bounds__(256) void k_classify(const RigDev rig, const CrowdDev cr, uint32_t slices, uint32_t words) {
    __shared__ uint32_t wave_n[kClassifyRounds * 4];  // [round][wave]
    __shared__ uint32_t block_base;
    const uint32_t tile = blockIdx.x / slices, slice = blockIdx.x - tile * slices;
    if (tile >= rig.T || rig.tile_lds[tile] < 2) return;
    const bool all = rig.tile_lds[tile] == 3;  // a joint list longer than k_skin2's ring is sized for: every instance is listed
    const uint32_t pbeg = rig.tile_pair_start[tile], np = rig.tile_pair_start[tile + 1] - pbeg;
    if (np == 0 && !all) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t i0 = slice * (kClassifyRounds * 256u);
    uint32_t mine = 0;                    // bit r: my instance of round r is slow
    uint32_t rank[kClassifyRounds];       // its position among the wave's slow instances of that round
#pragma unroll
    for (uint32_t r = 0; r < kClassifyRounds; ++r) {
        const uint32_t inst = i0 + r * 256u + tid;
        bool slow = false;
        if (inst < cr.I)
            slow = all || classify_pairs(rig.pairs + pbeg, np, cr.amask + (size_t)inst * words, words, cr.mw + (size_t)inst * rig.M, rig.sparse_pairs != 0).slow;
        const uint64_t bslow = __ballot(slow);
        if (lane == 0) wave_n[r * 4 + wave] = (uint32_t)__popcll(bslow);
        rank[r] = (uint32_t)__popcll(bslow & ((1ull << lane) - 1ull));
        if (slow) mine |= 1u << r;
    }
    __syncthreads();
    uint32_t before[kClassifyRounds], total = 0;  // entries of the block that precede my wave's of round r
#pragma unroll
    for (uint32_t r = 0; r < kClassifyRounds; ++r) {
        const uint4 n = *reinterpret_cast<const uint4*>(wave_n + r * 4);
        before[r] = total + (wave > 0 ? n.x : 0u) + (wave > 1 ? n.y : 0u) + (wave > 2 ? n.z : 0u);
        total += n.x + n.y + n.z + n.w;
    }
    if (total == 0) return;  // block-uniform
    if (tid == 0) block_base = atomicAdd(cr.fix + tile, total);
    __syncthreads();
    const uint32_t base = block_base;
#pragma unroll
    for (uint32_t r = 0; r < kClassifyRounds; ++r)
        if ((mine >> r) & 1u) {
            const uint32_t e = base + before[r] + rank[r];
            if (e < cr.I) cr.fix_inst[(size_t)tile * cr.I + e] = i0 + r * 256u + tid;  // (an instance is listed at most once per tile)
        }
}

// fix_role — the (tile, instance) pairs the self-staging role leaves out (k_classify lists them): instances with three or more active
// morph pairs in the tile, with a compacted pair, or in a tile with more than 128 pairs.  A block
// takes kFixChunk consecutive work-list entries (k_skin2 blocks append theirs together, so the
// tile — and with it the block's vertex data — rarely changes inside a chunk).  Per entry the
// palette rows of the tile's joints are staged in LDS (at most 126 16-byte pieces), the active
// pairs' deltas come by plain loads (L2-resident), and resident blocks cover each other's latency.
// Same arithmetic, in the same order, as the staged kernels (a13: targets ascending).
#ifndef SKINRT_FIX_ABLATE
#define SKINRT_FIX_ABLATE 0
#endif
constexpr uint32_t kFixChunk = 8;
// instances of a chunk morphed together (kFixChunk is a multiple): 0 = the product's choice per rig (launch_skin) — 4 for
// K = 1 (119 VGPRs: fits beside k_skin2's five blocks), 2 for K = 2 (4 needs 131-134 VGPRs and no longer fits beside
// k_skin2's four: C5 +8 %; 2 against 1: C5 -1.7 %, 10.49 / 10.47 / 10.46 -> 10.30 / 10.33 / 10.18).  C3 does not care
// (4.21-4.24 ms with 1, 2 or 4); a face under tracking (tools/exp_face.py: 52 targets all active) 6.90 -> 5.58 ms with 4.
// K = 1 rigs that hand whole tiles over (kind 3: every instance listed, most of them with no active pair at all) take 2:
// there the lists are long, k_skin_fix runs as long as k_skin2, and the batch's 36 extra registers only cost —
// C3's avatar cut into three skins (bench.py --multi-skin): k_skin_fix 4.80 -> 4.42 ms, the stage 4.64 -> 4.33.
#ifndef SKINRT_FIX_BATCH
#define SKINRT_FIX_BATCH 0
#endif
// dynamic LDS of a k_skin_fix block: two palette staging buffers of `pieces` 16-byte pieces (SkinPlan::fix_pieces: 128, or 256 where
// tiles with more than 21 joints are served), bounds partials, the tile's first 64 pair records
constexpr uint32_t fix_lds_bytes(uint32_t pieces) { return 2 * pieces * 16 + 4 * 6 * 4 + 32 + 64 * (uint32_t)sizeof(MorphPair); }
// fb = this block's index among the kernel's `fix_blocks` blocks (a multiple of 8 * ceil(T / 8))
template <int AR, int K, bool SKIP, int BATCH, uint32_t PIECES>
__device__ __forceinline__ void fix_role(const RigDev& rig, const CrowdDev& cr, const SkinPlan& plan, float4* smem, uint32_t fb, uint32_t fix_blocks) {
    constexpr uint32_t TV = kTileVerts;
    constexpr int W = 4 * K;
    constexpr bool EXACT = AR != 0, NEXACT = AR == 1;  // AR 2: positions contraction-free, normals / tangents fused
    constexpr uint32_t kFixBatch = SKINRT_FIX_BATCH ? SKINRT_FIX_BATCH : (uint32_t)BATCH;
    static_assert(kFixChunk % kFixBatch == 0, "a chunk is a whole number of batches");
    constexpr uint32_t pcap = PIECES;                                                     // 128, 256 or 384 (= plan.fix_pieces)
    f4* pal_s = reinterpret_cast<f4*>(smem);                                             // [2][pcap]
    float (*part)[6] = reinterpret_cast<float (*)[6]>(pal_s + 2 * pcap);                 // [4][6] (+ 32 B pad)
    MorphPair* pairs_l = reinterpret_cast<MorphPair*>(reinterpret_cast<char*>(part) + 4 * 6 * 4 + 32);  // the tile's first 64 pair records
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> (tile, chunk lane): consecutive block ids go round-robin to the 8 XCDs; every block of
    // a tile gets the same id modulo 8 (tile % 8: neighbouring tiles, which tend to be equally busy,
    // go to different XCDs), so the tile's deltas are fetched into ONE L2
    const uint32_t tiles_per_xcd = (rig.T + 7) / 8;
    const uint32_t kb = fb >> 3;
    const uint32_t tile = (kb % tiles_per_xcd) * 8 + (fb & 7);
    const uint32_t chunk_lane = kb / tiles_per_xcd, chunk_lanes = fix_blocks / (8 * tiles_per_xcd);
    if (tile >= rig.T) return;
    const uint32_t count = min(cr.fix[tile], cr.I);
    if (chunk_lane * kFixChunk >= count) return;
    const Outputs out{cr.pos, cr.nrm, cr.tan};
    const bool normalize = plan.normalize != 0;
    const uint32_t M = rig.M;
    const f4* pal_g = reinterpret_cast<const f4*>(cr.palette);
    // the block's tile: vertex data, joint list, pair records — loaded once
    const bool valid = tile * TV + tid < rig.V;
    float p0[3], n0[3], t0[3], w[W];
    uint32_t li[W];
    {
        const float* b = rig.base + (size_t)tile * 9 * TV + tid;
#pragma unroll
        for (int x = 0; x < 3; ++x) { p0[x] = b[x * TV]; n0[x] = b[(3 + x) * TV]; t0[x] = b[(6 + x) * TV]; }
        const float* wt = rig.wts + (size_t)tile * W * TV + tid;
        const uint16_t* lp = rig.lidx + (size_t)tile * W * TV + tid;
#pragma unroll
        for (int a = 0; a < W; ++a) { w[a] = wt[a * TV]; li[a] = lp[a * TV]; }
    }
    const uint32_t ninf = SKIP ? wave_influences<K>(w) : 4u * K;
    const uint32_t jstart = rig.tile_joint_start[tile];
    const uint32_t npieces = (rig.tile_joint_start[tile + 1] - jstart) * 6;  // <= plan.fix_pieces (<= 256 = the block's threads) for k_skin2's tiles
    const uint32_t joff = tid < npieces ? rig.tile_joints[jstart + tid / 6] * 6 + tid % 6 : 0u;
    // (PIECES = 384: tiles of 43 .. 64 joints handed over whole — threads 0 .. 127 move a second piece, tid + 256)
    const uint32_t joff2 = (PIECES > 256 && tid + 256 < npieces) ? rig.tile_joints[jstart + (tid + 256) / 6] * 6 + (tid + 256) % 6 : 0u;
    const uint32_t pbeg = rig.tile_pair_start[tile], pend = rig.tile_pair_start[tile + 1];
    if (tid < min(pend - pbeg, 64u)) pairs_l[tid] = rig.pairs[pbeg + tid];
    __syncthreads();
    const uint32_t* list = cr.fix_inst + (size_t)tile * cr.I;
    uint32_t flip = 0;  // palette staging buffer in use
    const uint32_t np64 = min(pend - pbeg, 64u);
    // kFixBatch instances at a time: the morph walks the tile's pairs ONCE for all of them — a pair's deltas are read
    // (from L2) once per batch, not once per instance that uses it.  Where most pairs are active (a face under tracking:
    // 20 pairs per tile, all non-zero) those reads were the kernel's time.  Per instance the pairs are still applied in
    // ascending order (a13), so the sums are the same.  The NEXT batch's palette pieces and pair weights are requested
    // while this one is worked on, into the registers this batch has just finished with.
    const uint32_t stride = chunk_lanes * kFixChunk;
    uint32_t e0 = chunk_lane * kFixChunk;
    uint32_t ents = 0;  // the chunk's instances in one load (lane k = entry e0 + k)
    // (the list is walked from its END, k_classify having filled it roughly in ascending instance order: like k_skin2's
    // ranges, so that the two kernels ask for the same instances' palettes at about the same time)
    if (!(SKINRT_RANGE_ORDER & 2)) list += count - 1;
    constexpr int kListStep = (SKINRT_RANGE_ORDER & 2) ? 1 : -1;
    if (e0 + lane < count && lane < kFixChunk) ents = list[kListStep * (int)(e0 + lane)];
    uint32_t inst[kFixBatch];
    f4 piece[kFixBatch];
    f4 piece2[PIECES > 256 ? kFixBatch : 1];
    float mwq[kFixBatch];  // lane q: weight of pair q (first 64 pairs) for instance b; 0 for b past the batch's end
    {
        const uint32_t nb0 = min(kFixBatch, min(count, e0 + kFixChunk) - e0);
#pragma unroll
        for (uint32_t b = 0; b < kFixBatch; ++b) {
            inst[b] = __builtin_amdgcn_readlane(ents, b < nb0 ? b : 0u);
            piece[b] = f4{0.0f, 0.0f, 0.0f, 0.0f};
            if (b < nb0 && tid < npieces) piece[b] = pal_g[(size_t)inst[b] * rig.J * 6 + joff];
            if constexpr (PIECES > 256) { if (b < nb0 && tid + 256 < npieces) piece2[b] = pal_g[(size_t)inst[b] * rig.J * 6 + joff2]; }
            mwq[b] = (b < nb0 && lane < np64) ? cr.mw[(size_t)inst[b] * M + pairs_l[lane].target] : 0.0f;
        }
    }
    for (; e0 < count; e0 += stride) {
        const uint32_t e1 = min(count, e0 + kFixChunk);
        const uint32_t e0n = e0 + stride;  // this block's next chunk
        uint32_t ents_next = 0;
        if (e0n + lane < count && lane < kFixChunk) ents_next = list[kListStep * (int)(e0n + lane)];
        for (uint32_t b0 = e0; b0 < e1; b0 += kFixBatch) {
            const uint32_t nb = min(kFixBatch, e1 - b0);
            // the batch after this one: the rest of the chunk, else the head of the next chunk
            uint32_t nnb = 0, inst_n[kFixBatch];
            if (b0 + kFixBatch < e1) {
                nnb = min(kFixBatch, e1 - b0 - kFixBatch);
#pragma unroll
                for (uint32_t b = 0; b < kFixBatch; ++b) inst_n[b] = __builtin_amdgcn_readlane(ents, (b0 - e0 + kFixBatch + (b < nnb ? b : 0u)) & 63u);
            } else {
                if (e0n < count) nnb = min(kFixBatch, min(count, e0n + kFixChunk) - e0n);
#pragma unroll
                for (uint32_t b = 0; b < kFixBatch; ++b) inst_n[b] = __builtin_amdgcn_readlane(ents_next, b < nnb ? b : 0u);
            }
            float p[kFixBatch][3], n[kFixBatch][3], t[kFixBatch][3];
            uint64_t act[kFixBatch];
            uint64_t any = 0;
#pragma unroll
            for (uint32_t b = 0; b < kFixBatch; ++b) {
#pragma unroll
                for (int x = 0; x < 3; ++x) { p[b][x] = p0[x]; n[b][x] = n0[x]; t[b][x] = t0[x]; }
                act[b] = (SKINRT_FIX_ABLATE & 1) ? 0ull : __ballot(mwq[b] != 0.0f);
                any |= act[b];
            }
            for (; any; any &= any - 1) {
                const uint32_t q = (uint32_t)__builtin_ctzll(any);
                const MorphPair mp = pairs_l[q];
                if (!((mp.wave_mask >> wave) & 1u)) continue;  // nothing of this wave's 64 vertices moves: adds exactly 0
                float d[9];
                const bool has = morph_load(rig.mblock + mp.data_off + (mp.dense ? wave * kMorphWaveFloats : 0u), mp.dense != 0, tid, d);
#pragma unroll
                for (uint32_t b = 0; b < kFixBatch; ++b) {
                    if (!((act[b] >> q) & 1ull)) continue;  // wave-uniform
                    const float mw = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(mwq[b]), q));
                    if (has) morph_add<EXACT>(mw, d, p[b], n[b], t[b]);
                }
            }
            if (pend - pbeg > 64) {  // pairs beyond the first 64 (rare): per instance, still ascending
#pragma unroll
                for (uint32_t b = 0; b < kFixBatch; ++b) {
                    if (b >= nb) continue;
                    const float* mwg = cr.mw + (size_t)inst[b] * M;
                    for (uint32_t q = pbeg + 64; q < pend; ++q) {
                        const MorphPair mp = rig.pairs[q];
                        const float mw = mwg[mp.target];
                        if (mw == 0.0f || !((mp.wave_mask >> wave) & 1u)) continue;  // wave-uniform; adds exactly 0
                        const bool dense = mp.dense != 0;
                        morph_pair<EXACT>(rig.mblock + mp.data_off + (dense ? wave * kMorphWaveFloats : 0u), dense, tid, mw, p[b], n[b], t[b]);
                    }
                }
            }
            // the weight registers are free: the next batch's weights
#pragma unroll
            for (uint32_t b = 0; b < kFixBatch; ++b)
                mwq[b] = (b < nnb && lane < np64) ? cr.mw[(size_t)inst_n[b] * M + pairs_l[lane].target] : 0.0f;
#pragma unroll
            for (uint32_t b = 0; b < kFixBatch; ++b) {
                if (b >= nb) continue;  // block-uniform
                // stage this instance's palette rows; the buffers alternate, so the one barrier below
                // also separates the previous entry's reads of this buffer from these writes
                f4* ps = pal_s + flip * pcap;
                flip ^= 1u;
                if (tid < npieces) ps[tid] = piece[b];
                if (b < nnb && tid < npieces) piece[b] = pal_g[(size_t)inst_n[b] * rig.J * 6 + joff];  // and the next batch's piece
                if constexpr (PIECES > 256) {
                    if (tid + 256 < npieces) ps[tid + 256] = piece2[b];
                    if (b < nnb && tid + 256 < npieces) piece2[b] = pal_g[(size_t)inst_n[b] * rig.J * 6 + joff2];
                }
                lds_barrier();  // LDS only: __syncthreads() would also wait for this wave's stores
                float op[3];
                skin_store<EXACT, NEXACT, K, SKIP>(LdsPal{ps}, w, li, ninf, p[b], n[b], t[b], normalize, valid && !(SKINRT_FIX_ABLATE & 2), out,
                                                   (size_t)inst[b] * rig.V * 3, (tile * TV + tid) * 12, op);
                if (plan.aabb) {
                    float a0 = op[0], a1 = op[1], a2 = op[2], c0 = op[0], c1 = op[1], c2 = op[2];
                    wave_minmax6(a0, a1, a2, c0, c1, c2);
                    if ((tid & 63) == 63) {
                        float* dst = part[wave];
                        dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = c0; dst[4] = c1; dst[5] = c2;
                    }
                    lds_barrier();  // LDS only: __syncthreads() would also wait for this wave's stores
                    if (tid < 6) {
                        float r = part[0][tid];
                        for (int q = 1; q < 4; ++q) r = tid < 3 ? fminf(r, part[q][tid]) : fmaxf(r, part[q][tid]);
                        cr.aabb_part[((size_t)inst[b] * rig.T + tile) * 6 + tid] = r;
                    }
                    // part[] is next written after the next entry's first barrier
                }
            }
            // pieces of next-batch entries past this batch's end (nnb > nb: a short batch before a full one) are still to fetch
#pragma unroll
            for (uint32_t b = 0; b < kFixBatch; ++b) {
                if (b >= nb && b < nnb && tid < npieces) piece[b] = pal_g[(size_t)inst_n[b] * rig.J * 6 + joff];
                if constexpr (PIECES > 256) { if (b >= nb && b < nnb && tid + 256 < npieces) piece2[b] = pal_g[(size_t)inst_n[b] * rig.J * 6 + joff2]; }
                inst[b] = inst_n[b];
            }
        }
        ents = ents_next;
    }
}


// Bounds of a wave's 64 positions in 20 VALU operations instead of wave_minmax6's 42: minima are carried
// as maxima of the negated value (exact), and the six values are reduced by TRANSPOSE steps — gfx950's
// v_permlane32_swap / v_permlane16_swap exchange half-waves / rows between two registers, so one swap + one
// v_max halves TWO values at once.  Result (one register): lanes 8k .. 8k+7 all hold component
// kBoundsComp[k] of the wave's bounds, k = lane / 8:  +max x, +max z, +max y, +max z, -min x, -min z, -min y, -min z.
__device__ __forceinline__ float vmax_raw(uint32_t a, uint32_t b) {  // plain v_max_f32: fmaxf() adds a canonicalize per input
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float wave_bounds_packed(const float (&v)[3], bool upper_half_of_row) {
    float r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {  // [max partials of c | max partials of -c], 32 lanes each
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[c]), __float_as_uint(-v[c]), false, false);
        r[c] = vmax_raw(sw[0], sw[1]);
    }
    // rows of 16: r[0] = [x x -x -x], r[1] = [y y -y -y], r[2] = [z z -z -z]
    const auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[0]), __float_as_uint(r[1]), false, false);
    float a = vmax_raw(s01[0], s01[1]);  // rows: x, y, -x, -y
    const auto s22 = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[2]), __float_as_uint(r[2]), false, false);
    float b = vmax_raw(s22[0], s22[1]);  // rows: z, z, -z, -z
    // halve the rows (lane l with lane l ^ 8), merge the two registers, then finish inside the 8-lane groups
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b));
    float m = upper_half_of_row ? b : a;
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(m));
    return m;
}

constexpr uint32_t kSkin2Threads = 256;

// LONG: tiles with 22 .. kSkin2MaxJoints joints (a wave then moves its palette rows with up to six LDS-DMA instructions instead of two)
template <int AR, int K, int LOOK, bool SKIP, bool SPARSE, bool LONG>
__global__ __launch_bounds__(kSkin2Threads, (K == 1 ? SKINRT_BLOCKS_K1 : 4)) void k_skin2(const RigDev rig, const CrowdDev cr, const SkinPlan plan) {
    extern __shared__ float4 smem[];
    constexpr uint32_t TV = kTileVerts;
    constexpr int W = 4 * K;
    constexpr bool EXACT = AR != 0, NEXACT = AR == 1;  // AR 2: positions contraction-free, normals / tangents fused
    constexpr uint32_t L = LOOK, R = L + 1;  // palette lookahead (plan.look2) and ring slots
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t xcd = blockIdx.x & 7, kblk = blockIdx.x >> 3;
    const uint32_t tc = xcd % plan.tile_chunks, rc = xcd / plan.tile_chunks;
    // an XCD's tiles: every tile_chunks-th tile (work per tile varies along the mesh — morph activity, empty weight
    // sets — and interleaving evens it out over the XCDs: C3 -1.9 %), or one contiguous run where an instance range's
    // palette is too large for one L2 and neighbouring tiles sharing joints matters more (C5: +1 % interleaved)
    // (plan.run2 = tiles per run: 1 = fully interleaved, tiles_per_chunk2 = one contiguous chunk per XCD)
    const uint32_t jt = kblk % plan.tiles_per_chunk2;
    uint32_t tile = ((jt / plan.run2) * plan.tile_chunks + tc) * plan.run2 + jt % plan.run2;
    uint32_t rng = rc * plan.ranges_per_chunk2 + kblk / plan.tiles_per_chunk2;
    if (plan.list2_n) {  // one of two launches (SkinPlan::split2): the position just computed indexes this launch's tiles
        if (tile >= plan.list2_n || rng >= plan.ranges2) return;
        tile = rig.tile_order2[plan.list2_first + tile];
    }
    if (tile >= rig.T || rng >= plan.ranges2) return;
    if (rig.tile_lds[tile] != 2) return;  // other tiles: k_skin
    // The grid walks the instance ranges from the LAST to the first.  The pose stage has just written the palettes in
    // ascending instance order, and what of them is still in the memory-side cache (256 MB, written lines allocate,
    // least recently written go first) is the END of the crowd: C3's pose stage writes 0.4 GB, so walking upwards
    // every first touch of a palette row went to HBM, against the store stream (round 3: 4.19-4.22 -> 4.06 ms;
    // tools/exp_split.py).  SKINRT_RANGE_ORDER=3 (measurement builds): ascending, as until then.
    if (!(SKINRT_RANGE_ORDER & 1)) rng = plan.ranges2 - 1 - rng;
    const uint32_t i0 = rng * plan.range2;
    const uint32_t i1 = min(cr.I, i0 + plan.range2);
    const uint32_t nin = i1 - i0;  // <= kSkin2Range
    const uint32_t JTP = plan.jt2;  // joints per palette slot
    const uint32_t M = rig.M;
    // LDS: block tables, then per wave {palette ring, morph slots}
    uint4* hdr_s = reinterpret_cast<uint4*>(smem);                              // [range] instance headers, processing order
    MorphPair* pairs_s = reinterpret_cast<MorphPair*>(hdr_s + kSkin2Range);     // [max_pairs]
    float* bb_s = reinterpret_cast<float*>(pairs_s + plan.max_pairs);           // [range][6] bounds per processing position
    float* wave_s = bb_s + kSkin2Range * 6;
    // prologue scratch, overlaid on the wave areas (unused until the pipeline starts)
    uint32_t* cnt_s = reinterpret_cast<uint32_t*>(wave_s);                      // [8] counters
    uint32_t* mask_s = cnt_s + 8;                                               // [range] active-pair mask per instance-in-range
    uint16_t* perm_s = reinterpret_cast<uint16_t*>(mask_s + kSkin2Range);       // [range] sorted order
    uint16_t* start_s = perm_s + kSkin2Range;                                   // [runs] sorted position where run j starts
    float2* w12_s = reinterpret_cast<float2*>(start_s + kSkin2Range);           // [range] weights of the first two active pairs
    uint64_t* key_s = reinterpret_cast<uint64_t*>(w12_s + kSkin2Range);         // [range] sort keys (class, mask, rotated index)
    const uint32_t pal_slot_f = JTP * 24;                                       // floats per palette slot
    const uint32_t per_wave_f = R * pal_slot_f + (M ? kMorphSlots * kMorphWaveFloats : 0u);
    float* pal_ring = wave_s + (size_t)wave * per_wave_f;                       // [R][JTP][6] f4
    float* my_slots = pal_ring + R * pal_slot_f;                                // [kMorphSlots][576]

    const uint32_t pbeg = rig.tile_pair_start[tile];
    const uint32_t np = rig.tile_pair_start[tile + 1] - pbeg;
    const uint32_t jstart = rig.tile_joint_start[tile];
    const uint32_t jcnt = rig.tile_joint_start[tile + 1] - jstart;  // <= 21 (LONG: <= kSkin2MaxJoints)

    // ---------------- block prologue (all 256 threads; no store has been issued yet) ----------------
    for (uint32_t c = tid; c < np; c += kSkin2Threads) pairs_s[c] = rig.pairs[pbeg + c];
    if (plan.aabb)
        for (uint32_t c = tid; c < kSkin2Range * 6; c += kSkin2Threads) bb_s[c] = -__builtin_inff();  // cells hold {-min, +max}: all folded with max
    __syncthreads();
    // pairs that move at least one of THIS wave's 64 vertices (bit q, first 128 pairs): the others are
    // neither requested nor applied by this wave (their deltas here are all zero)
    const uint64_t my_pairs_lo = __ballot(lane < np && ((pairs_s[min(lane, np ? np - 1 : 0u)].wave_mask >> wave) & 1u));
    const uint64_t my_pairs_hi = __ballot(lane + 64 < np && ((pairs_s[min(lane + 64, np ? np - 1 : 0u)].wave_mask >> wave) & 1u));
    auto mine = [&](uint32_t q) -> bool { return (((q & 64u) ? my_pairs_hi : my_pairs_lo) >> (q & 63u)) & 1ull; };  // q < 128 = kSkin2MaxPairs, wave-uniform
    // active-pair mask of instance-in-range `tid` (bit min(q, 31) = pair q has a non-zero weight) and its class:
    // 0 idle, 1 fast (1 or 2 active pairs: the LDS slots, requested ahead), 3 handed over (3 or more active pairs, or a
    // tile with more than 128 pairs: on the work list of k_skin_fix already) or padding.  Classes 0 and 1 are this block's.
    uint32_t my_mask = 0xffffffffu, my_cls = 3u;
    {
        float w_first = 0.0f, w_second = 0.0f;
        if (tid < nin) {
            const uint32_t words = (M + 63) / 64;
            const PairClass pc = classify_pairs(pairs_s, np, cr.amask + (size_t)(i0 + tid) * words, words, cr.mw + (size_t)(i0 + tid) * M, SPARSE);
            my_mask = pc.mask;
            my_cls = pc.slow ? 3u : (pc.mask == 0 ? 0u : 1u);
            w_first = pc.w_first;
            w_second = pc.w_second;
        }
        const uint64_t bfast = __ballot(my_cls <= 1u);
        if (lane == 0 && wave < 2) cnt_s[4 + wave] = (uint32_t)__popcll(bfast);
        if (tid < kSkin2Range) { mask_s[tid] = my_mask; w12_s[tid] = make_float2(w_first, w_second); }
    }
    __syncthreads();
    const uint32_t nproc = cnt_s[4] + cnt_s[5];  // instances this block processes itself (idle + fast)
    // processing order: equal masks become neighbours (rank sort on distinct (mask, index) keys)
    if (tid < kSkin2Range) {
        // Ties are broken by the instance's index ROTATED by a tile-dependent amount: neighbouring tiles then walk a
        // range's instances out of phase (C3: 4.13 -> 4.00 ms with the morphs ablated, where every tile's order was the
        // natural one; 1-1.5 % in the full build, whose orders already differ by their active-pair masks).  The stores are
        // not what gains — the store stream alone is indifferent to the phase (tools/ubench/store_phase.hip) — so it is the
        // read side: in phase, every tile asks for the same instance's palette rows at the same moment.  The result does
        // not depend on the order.
        // (Starting the block's four WAVES at different points of the order as well: slower, 3.9 -> 4.1 ms ablated — the
        // waves' four 768-byte pieces of an instance row want to be written together.)
        const uint32_t rot = nin ? (tile * SKINRT_ORDER_ROT) % nin : 0u;
        const uint32_t rkey = tid < nin ? (tid + rot < nin ? tid + rot : tid + rot - nin) : tid;
        // one 64-bit key per instance: class, then mask, then rotated index (all keys distinct)
        key_s[tid] = ((uint64_t)my_cls << 40) | ((uint64_t)my_mask << 8) | rkey;
    }
    __syncthreads();
    if (tid < kSkin2Range) {
        const uint64_t mine = key_s[tid];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < kSkin2Range; j += 2) {
            const ulonglong2 kj = *reinterpret_cast<const ulonglong2*>(key_s + j);
            rank += (kj.x < mine ? 1u : 0u) + (kj.y < mine ? 1u : 0u);
        }
        perm_s[rank] = (uint16_t)tid;
    }
    __syncthreads();
    // Run structure of the sorted order.  Instances with no active pair (mask 0, sorted first) never
    // read the morph slots, so they are dealt out as SPACERS between runs of equal masks: the pairs
    // of the run after a spacer are requested at the spacer's first instance, at least L instances
    // before they are read, and the ordinary palette wait then covers them — no store has to drain.
    // Runs that get no spacer (too few idle instances) request their pairs one instance ahead and
    // wait for all but that instance's last stores ("near" reload).
    const bool in_range = tid < nproc;
    const uint32_t sm = in_range ? mask_s[perm_s[tid]] : 0xffffffffu;
    const uint32_t sprev = (in_range && tid) ? mask_s[perm_s[tid - 1]] : 0u;
    const bool nz = in_range && sm != 0;
    const bool start = nz && (tid == 0 || sm != sprev);
    const uint64_t bstart = __ballot(start), bzero = __ballot(in_range && !nz);
    if (wave < 2 && lane == 0) { cnt_s[wave * 2] = (uint32_t)__popcll(bstart); cnt_s[wave * 2 + 1] = (uint32_t)__popcll(bzero); }
    __syncthreads();
    const uint32_t runs = cnt_s[0] + cnt_s[2];
    const uint32_t zeros = cnt_s[1] + cnt_s[3];
    const uint32_t run = (wave == 1 ? cnt_s[0] : 0u) + (uint32_t)__popcll(bstart & ((2ull << lane) - 1ull)) - 1u;  // of a non-zero element
    if (start && tid < kSkin2Range) start_s[run] = (uint16_t)tid;
    uint32_t nsp = 0, sp = 0;  // spacers, instances per spacer
    if (runs > 1) {
        nsp = min(runs - 1, zeros / L);
        if (nsp) sp = zeros / nsp;
    }
    const uint32_t z0 = zeros - nsp * sp;  // idle instances left over: processed first
    __syncthreads();
    // header of one instance, written at its processing position:
    //   x = active pairs (0, 1 or 2) | near << 8 | prefetch << 10 | pairs to prefetch << 11 |
    //       first pair compacted << 13 | second compacted << 14 | first pair << 16 | second << 24;
    //   y = instance;   z, w = weights of those two pairs
    // near: its pairs are requested during the previous instance; prefetch (idle instances only):
    // request the named pairs, which belong to the run after this spacer.
    if (in_range) {
        const uint32_t k = perm_s[tid];
        const uint32_t inst = i0 + k;
        uint32_t m = sm, fpos, near = 0, pf = 0;
        if (nz) {
            fpos = z0 + (tid - zeros) + min(run, nsp) * sp;
            near = (start && run > nsp) ? 1u : 0u;
        } else if (tid < z0) {
            fpos = tid;
        } else {
            const uint32_t idx = tid - z0, sj = idx / sp, within = idx - sj * sp;
            const uint32_t nxt = start_s[sj + 1];
            fpos = z0 + (nxt - zeros) + sj * sp + within;
            if (within == 0) { pf = 1; m = mask_s[perm_s[nxt]]; }  // describe the next run's pairs
        }
        uint32_t na, q1 = 0, q2 = 0;  // na <= 2
        if (np > 32) {  // PairClass::mask of a tile with 33 .. 128 pairs: the two indices themselves
            q1 = m & 0xffu;
            q2 = (m >> 8) & 0xffu;
            na = m ? (q1 == q2 ? 1u : 2u) : 0u;
        } else {
            na = (uint32_t)__popc(m);
            if (na) {
                q1 = (uint32_t)__ffs((int)m) - 1;
                const uint32_t m2 = m & (m - 1);
                q2 = m2 ? (uint32_t)__ffs((int)m2) - 1 : q1;
            }
        }
        float w1 = 0.0f, w2 = 0.0f;
        if (na >= 1) {
            if (!pf) {
                const float2 w12 = w12_s[k];
                w1 = w12.x;
                w2 = na >= 2 ? w12.y : w12.x;  // one active pair: both header weights name it (as q2 = q1)
            }
        }
        const uint32_t sparse12 = na ? ((pairs_s[q1].dense ? 0u : 1u) << 13) | ((pairs_s[q2].dense ? 0u : 1u) << 14) : 0u;
        const uint32_t x = pf ? ((1u << 10) | (na << 11) | (q1 << 16) | (q2 << 24))
                              : (na | (near << 8) | sparse12 | (q1 << 16) | (q2 << 24));
        hdr_s[fpos] = make_uint4(x, inst, __float_as_uint(w1), __float_as_uint(w2));
    }
    // this thread's vertex
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
    const uint32_t ninf = (SKINRT_EXP & 8) ? (wave + 1) * K : (SKIP ? wave_influences<K>(w) : 4u * K);
    // palette pieces this lane moves: piece c = (list position c / 6, 16-byte quarter c % 6), c = lane and lane + 64
    const uint32_t npieces = jcnt * 6;  // <= 126 (LONG: <= 384)
    uint32_t joff0 = 0, joff1 = 0;      // byte offsets inside one instance's palette
    if (lane < npieces) joff0 = (rig.tile_joints[jstart + lane / 6] * 6 + lane % 6) * 16;
    if (lane + 64 < npieces) joff1 = (rig.tile_joints[jstart + (lane + 64) / 6] * 6 + (lane + 64) % 6) * 16;
    uint32_t joffx[4] = {0, 0, 0, 0};  // (LONG: pieces 128 .. 383)
    if constexpr (LONG) {
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t c = lane + 128 + 64 * q;
            if (c < npieces) joffx[q] = (rig.tile_joints[jstart + c / 6] * 6 + c % 6) * 16;
        }
    }
    const Outputs out{cr.pos, cr.nrm, cr.tan};
    const uint32_t olb = v * 12;
    const bool normalize = plan.normalize != 0;
    const uint32_t wave_valid = __builtin_amdgcn_readfirstlane(__ballot(valid) != 0ull ? 1u : 0u);
    // stores this wave issues per instance: position / normal / tangent (none if the wave holds no
    // real vertex)
    constexpr uint32_t spi = 3u;
    // a wave none of whose lanes holds a real vertex (tail of the last tile) stores nothing and its bounds repeat a
    // real vertex of another wave: it sits the loop out, which makes "3 stores per instance" a constant of the waits
    const uint32_t nloop = wave_valid ? nproc : 0u;
    const char* pal_g = reinterpret_cast<const char*>(cr.palette);
    const size_t pal_stride = (size_t)rig.J * 96;
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)pal_ring);
    const uint32_t bb_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)bb_s);
    // the cell component this lane folds into (lanes 8k: see wave_bounds_packed): {-min x,y,z, +max x,y,z} = components 0 .. 5
    const uint32_t bb_lane = bb_lds + ((0x21205453u >> ((lane >> 3) * 4)) & 7u) * 4;
    auto pal_dma = [&](uint32_t inst, uint32_t slot) {  // request one instance's palette rows into ring slot `slot`
        if (SKINRT_EXP & 1024) return;  // timing experiment: no palette reads at all (stale LDS)
        const char* gbase = pal_g + (size_t)inst * pal_stride;
        const uint32_t lds0 = ring_lds + slot * pal_slot_f * 4;
        if (lane < npieces) SKINRT_GLDS_AT(joff0, gbase, lds0);
        if (lane + 64 < npieces) {
            const uint32_t lds1 = lds0 + 1024;
            SKINRT_GLDS_AT(joff1, gbase, lds1);
        }
        if constexpr (LONG) {
            if (npieces > 128) {  // wave-uniform.  (More requests per instance only make the counted waits more conservative: they count stores.)
                if (lane + 128 < npieces) { const uint32_t lds2 = lds0 + 2048; SKINRT_GLDS_AT(joffx[0], gbase, lds2); }
                if (lane + 192 < npieces) { const uint32_t lds3 = lds0 + 3072; SKINRT_GLDS_AT(joffx[1], gbase, lds3); }
                if (npieces > 256) {
                    if (lane + 256 < npieces) { const uint32_t lds4 = lds0 + 4096; SKINRT_GLDS_AT(joffx[2], gbase, lds4); }
                    if (lane + 320 < npieces) { const uint32_t lds5 = lds0 + 5120; SKINRT_GLDS_AT(joffx[3], gbase, lds5); }
                }
            }
        }
    };
    auto morph_dma = [&](uint32_t hx) {  // request the first (up to) two active pairs of an instance into the slots
        if (SKINRT_ABLATE & 2) return;
        const uint32_t ns = min(hx & 0xffu, kMorphSlots);
        const uint32_t lane_bytes = lane * 16;
        for (uint32_t a = 0; a < ns; ++a) {
            const uint32_t q = (hx >> (16 + 8 * a)) & 0xffu;
            if (!mine(q)) continue;
            const MorphPair mp = pairs_s[q];
            const uint32_t doff = __builtin_amdgcn_readfirstlane(mp.data_off);
            const bool dense = __builtin_amdgcn_readfirstlane((uint32_t)mp.dense) != 0;
            const uint32_t n4 = dense ? kMorphWaveFloats / 4 : (__builtin_amdgcn_readfirstlane(mp.floats) + 3) / 4;
            const float* gbase = rig.mblock + (doff + (dense ? wave * kMorphWaveFloats : 0u));
            const uint32_t lds_base = __builtin_amdgcn_readfirstlane(
                (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)(my_slots + a * kMorphWaveFloats));
            if (lane < n4) SKINRT_GLDS(0);
            if (lane + 64 < n4) SKINRT_GLDS(1024);
            if (lane + 128 < n4) SKINRT_GLDS(2048);
        }
    };
    __syncthreads();  // headers visible

    // ---------------- per-wave pipeline ----------------
    for (uint32_t q = 0; q < min(L, nloop); ++q) pal_dma(__builtin_amdgcn_readfirstlane(hdr_s[q].y), q);  // q < L < R
    uint32_t hx, hi, hz, hw;
    {
        const uint4 h0 = hdr_s[0];
        hx = __builtin_amdgcn_readfirstlane(h0.x);
        hi = __builtin_amdgcn_readfirstlane(h0.y);
        hz = __builtin_amdgcn_readfirstlane(h0.z);
        hw = __builtin_amdgcn_readfirstlane(h0.w);
    }
    // the first run's pairs (wherever it starts: the slots are idle until then)
    if (runs) morph_dma(z0 ? __builtin_amdgcn_readfirstlane(hdr_s[z0].x) : hx);
    wait_vmcnt<0>();
    bool dma_pending = true;  // "this instance's morph pairs were requested": true for instance 0 (just waited for everything)
    uint32_t inst_ahead = nloop > L ? __builtin_amdgcn_readfirstlane(hdr_s[L].y) : 0u;  // instance at pos + L
    uint32_t slot_cur = 0, slot_ahead = L % R;  // ring slots of instance pos and of instance pos + L
    for (uint32_t pos = 0; pos < nloop; ++pos) {
        const uint32_t i = hi;
        const uint32_t na = (SKINRT_ABLATE & 1) ? 0u : (hx & 0xffu);
        const bool reload = ((hx >> 8) & 1u) != 0;
        const float mw1 = __uint_as_float(hz), mw2 = __uint_as_float(hw);
        const bool dense1 = !SPARSE || ((hx >> 13) & 1u) == 0, dense2 = !SPARSE || ((hx >> 14) & 1u) == 0;
        const size_t obase = (size_t)i * rig.V * 3;
        // (1) wait: this instance's palette was requested L instances ago and every instance since
        // issued at least spi stores; its morph pairs (if they had to be reloaded) were requested
        // before the previous instance's last spi - 1 stores.  First instance: everything has landed.
        if (pos) {
            if (reload && dma_pending) wait_vmcnt<spi - 1>();
            else wait_vmcnt<spi * L>();
        }
        // (2) request the palette of instance pos + L, and fetch the instance id after that one
        uint32_t ahead_next = 0;
        if (pos + L < nloop) {
            pal_dma(inst_ahead, slot_ahead);
            if (pos + L + 1 < nloop) ahead_next = hdr_s[pos + L + 1].y;
        }
        if ((hx >> 10) & 1u) morph_dma((hx & 0xffff0000u) | ((hx >> 11) & 3u));  // spacer: the next run's pairs
        uint4 hdr_next = make_uint4(0, 0, 0, 0);
        if (pos + 1 < nloop) hdr_next = hdr_s[pos + 1];
        // (3) morph (a13): p = p0 + sum_m w_m * delta_m, targets ascending
        float p[3] = {p0[0], p0[1], p0[2]}, n[3] = {n0[0], n0[1], n0[2]}, t[3] = {t0[0], t0[1], t0[2]};
        if (na) {
            if (mine((hx >> 16) & 0xffu)) morph_pair<EXACT>(my_slots, dense1, tid, mw1, p, n, t);
            if (na >= 2 && mine(hx >> 24)) morph_pair<EXACT>(my_slots + kMorphWaveFloats, dense2, tid, mw2, p, n, t);
        }
        // (4) position
        float op[3];
        {
            float S[12];
            blend_position<EXACT, K, SKIP>(LdsPal{reinterpret_cast<const f4*>(pal_ring + slot_cur * pal_slot_f)}, w, li, ninf, S);
            apply_position<EXACT>(S, p, op);
        }
        if (valid) store3(out.pos + obase, olb, op);
        // (5) request the next instance's morph pairs: every slot read of this instance has retired
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        hx = __builtin_amdgcn_readfirstlane(hdr_next.x);
        hi = __builtin_amdgcn_readfirstlane(hdr_next.y);
        hz = __builtin_amdgcn_readfirstlane(hdr_next.z);
        hw = __builtin_amdgcn_readfirstlane(hdr_next.w);
        inst_ahead = __builtin_amdgcn_readfirstlane(ahead_next);
        dma_pending = false;
        if ((hx >> 8) & 1u) { morph_dma(hx); dma_pending = true; }
        __builtin_amdgcn_sched_barrier(0);  // the stores below stay younger than the requests above
        // (6) normal / tangent
        float on[3], ot[3];
        {
            float Nm[9];
            blend_normal<NEXACT, K, SKIP>(LdsPal{reinterpret_cast<const f4*>(pal_ring + slot_cur * pal_slot_f)}, w, li, ninf, Nm);
            apply_normal<NEXACT>(Nm, n, t, normalize, on, ot);
        }
        if (valid) {
            store3(out.nrm + obase, olb, on);
            store3(out.tan + obase, olb, ot);
        }
        // (7) bounds of this wave's 64 vertices (lanes past the mesh end repeat a real vertex), folded
        // into the block's cell for this position with LDS atomics: nothing is returned, so nothing
        // waits.  Per-wave records in HBM (two 12-byte stores per wave and instance) cost 0.8-1.2 ms
        // of the C3 frame as scattered partial-line writes.
        slot_cur = slot_cur + 1 == R ? 0u : slot_cur + 1;
        slot_ahead = slot_ahead + 1 == R ? 0u : slot_ahead + 1;
        if (plan.aabb) {
            const float m = wave_bounds_packed(op, (lane & 8u) != 0);
            if ((lane & 7u) == 0) {
                // asm: left to the compiler, an LDS atomic is wrapped in a lane loop that pre-reduces equal addresses
                const uint32_t cell = bb_lane + pos * 24;
                asm volatile("ds_max_f32 %0, %1" : : "v"(cell), "v"(m) : "memory");
            }
        }
    }
    // ---------------- block epilogue: one bounds record per (instance, tile) ----------------
    if (plan.aabb) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the asm atomics above
        __syncthreads();
        if (tid < nproc) {
            const float* cell = bb_s + tid * 6;
            float* dst = cr.aabb_part + ((size_t)hdr_s[tid].y * rig.T + tile) * 6;
            // non-temporal on large crowds (plan.aabb == 2, skinrt_evaluate): 56 MB of records per C3 frame would otherwise sit
            // in the memory-side cache in place of palettes this stage has yet to read (stage -0.02 ms, k_aabb_reduce +0.009
            // reading them from HBM).  Small crowds, whose palettes fit that cache with room to spare, keep plain stores
            // (C2 +5 % with nt).
            if (plan.aabb == 2) {
#pragma unroll
                for (int c = 0; c < 6; ++c)  // asm: as two branches of builtin stores the compiler merges them and drops the hint
                    asm volatile("global_store_dword %0, %1, off nt" : : "v"(dst + c), "v"(c < 3 ? -cell[c] : cell[c]) : "memory");
            } else {
#pragma unroll
                for (int c = 0; c < 6; ++c) dst[c] = c < 3 ? -cell[c] : cell[c];
            }
        }
    }
}

// The fix role's kernel.  (Tried: the role inside the k_skin2 launch, its block groups dealt out between the
// self-staging groups.  Slower — 5.7-6.0 ms against 4.7 on C3: a launch has ONE LDS size, so every fix block takes
// one of the CU's five 31 KiB block slots from k_skin2; as a kernel of its own it fits into what k_skin2 leaves free.)
template <int AR, int K, bool SKIP, int BATCH, uint32_t PIECES>
__global__ __launch_bounds__(256) void k_skin_fix(const RigDev rig, const CrowdDev cr, const SkinPlan plan) {
    extern __shared__ float4 smem[];
    fix_role<AR, K, SKIP, BATCH, PIECES>(rig, cr, plan, smem, blockIdx.x, gridDim.x);
}

// a14: bounds of one instance = min/max over its tiles' partial bounds (order-independent, exact).
// One 192-thread block per instance: an instance's partials are 6*T contiguous floats, thread t owns
// component t % 6 and strides by 192 (a multiple of 6), so every load is coalesced.
constexpr uint32_t kAabbThreads = 192;
__global__ __launch_bounds__(kAabbThreads) void k_aabb_reduce(const RigDev rig, const CrowdDev cr) {
    __shared__ float part[kAabbThreads];
    const uint32_t i = blockIdx.x, t = threadIdx.x, c = t % 6;
    const float* src = cr.aabb_part + (size_t)i * rig.T * 6;  // [T][6]
    const uint32_t n = rig.T * 6;
    float r = c < 3 ? __builtin_inff() : -__builtin_inff();
    for (uint32_t e = t; e < n; e += kAabbThreads) r = c < 3 ? fminf(r, src[e]) : fmaxf(r, src[e]);
    part[t] = r;
    __syncthreads();
    if (t < 6) {
        for (uint32_t q = t + 6; q < kAabbThreads; q += 6) r = c < 3 ? fminf(r, part[q]) : fmaxf(r, part[q]);
        cr.aabb[(size_t)i * 6 + t] = r;
    }
}

__global__ __launch_bounds__(256) void k_fill_mw(float* mw, const float* defaults, uint64_t total, uint32_t M) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < total) mw[g] = defaults[g % M];
}

template <bool EXACT, bool LDSPAL>
hipError_t launch_skin_k(const RigDev& rig, const CrowdDev& cr, const SkinPlan& plan, hipStream_t s) {
    // this launch's tiles in RigDev::tile_order: the L2-gather form takes kind 0, the LDS-staged form kind 1 — and kinds 2 / 3 where
    // the self-staging kernel is switched off
    const uint32_t first = LDSPAL ? rig.order_n0 : 0u;
    const uint32_t ntiles = LDSPAL ? (plan.v2 ? rig.order_n1 : rig.T - rig.order_n0) : rig.order_n0;
    if (ntiles == 0) return hipSuccess;
    const uint32_t tpc = (ntiles + plan.tile_chunks - 1) / plan.tile_chunks;
    const dim3 grid(8u * tpc * plan.ranges_per_chunk), block(kSkinThreads);
    switch (rig.K) {
        case 1: hipLaunchKernelGGL((k_skin<EXACT, 1, LDSPAL>), grid, block, plan.lds_bytes, s, rig, cr, plan, first, ntiles, tpc); break;
        case 2: hipLaunchKernelGGL((k_skin<EXACT, 2, LDSPAL>), grid, block, plan.lds_bytes, s, rig, cr, plan, first, ntiles, tpc); break;
        case 3: hipLaunchKernelGGL((k_skin<EXACT, 3, LDSPAL>), grid, block, plan.lds_bytes, s, rig, cr, plan, first, ntiles, tpc); break;
        case 4: hipLaunchKernelGGL((k_skin<EXACT, 4, LDSPAL>), grid, block, plan.lds_bytes, s, rig, cr, plan, first, ntiles, tpc); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace

hipError_t launch_skin(const RigDev& rig, const CrowdDev& cr, const SkinPlan& plan, hipStream_t s, const SkinConcurrency& conc) {
    if (cr.I == 0 || rig.T == 0) return hipSuccess;
    // what the kernel's indexing assumes about the plan (checked here, on the host, every launch)
    if (plan.batch == 0 || plan.range == 0 || plan.range > 256 || plan.range % plan.batch != 0 ||
        (uint64_t)plan.ranges * plan.range < cr.I ||
        plan.tile_chunks == 0 || 8 % plan.tile_chunks != 0 ||
        (uint64_t)plan.tiles_per_chunk * plan.tile_chunks < rig.T ||
        (uint64_t)plan.ranges_per_chunk * (8 / plan.tile_chunks) < plan.ranges)
        return hipErrorInvalidValue;
    if (plan.v2 && (plan.range2 == 0 || plan.range2 > kSkin2Range || (uint64_t)plan.ranges2 * plan.range2 < cr.I ||
                    plan.lds_bytes2 > kSkin2MaxLds || rig.K > 2 || plan.jt2 == 0 || plan.jt2 > kSkin2MaxJoints || (plan.fix_pieces != 128 && plan.fix_pieces != 256 && plan.fix_pieces != 384) || plan.look2 < 1 || plan.look2 > 2 ||
                    (uint64_t)plan.ranges_per_chunk2 * (8 / plan.tile_chunks) < plan.ranges2 ||
                    plan.run2 == 0 || plan.tiles_per_chunk2 % plan.run2 != 0 || plan.tiles_per_chunk2 < plan.tiles_per_chunk))
        return hipErrorInvalidValue;
    hipError_t e = hipSuccess;
    // the loader-wave kernel: tiles of kinds 1 and 0, and kind 2 as well where the self-staging kernel is switched off.
    // Its LDS need is checked only when it is launched at all (skinrt_create_crowd refuses such rigs).
    const bool others = (plan.kinds & 3u) != 0;
    auto launch_k_skin = [&](hipStream_t ks) -> hipError_t {
        hipError_t ke = hipSuccess;
        if (plan.lds_bytes > 64 * 1024) return hipErrorInvalidValue;
        if ((plan.kinds & 2u) || (!plan.v2 && (plan.kinds & 4u)))
            ke = plan.exact ? launch_skin_k<true, true>(rig, cr, plan, ks) : launch_skin_k<false, true>(rig, cr, plan, ks);
        if (ke == hipSuccess && (plan.kinds & 1u))
            ke = plan.exact ? launch_skin_k<true, false>(rig, cr, plan, ks) : launch_skin_k<false, false>(rig, cr, plan, ks);
        return ke;
    };
    bool others_done = false;
    // tiles of kind 2 (short joint lists): the self-staging kernel unless it is switched off
    if (plan.v2 && (plan.kinds & 4u)) {
        // Work lists first (k_classify), then the two consumers at the same time: k_skin_fix (latency-bound, few
        // stores, 5 KiB of LDS) on the side stream fits into the wave slots, registers and LDS that k_skin2's five
        // blocks per CU leave free, so it costs the stage almost nothing; one after the other it costs its full
        // 0.5 ms (C3).  Fork / join by events: the caller's stream sees the stage as one unit.
        // (a tile needs at least 3 pairs to hand an instance over: without such a tile there is no list to build and no
        // k_skin_fix to launch; the active-pair bit rows are k_skin2's own input either way)
        // The tiles k_skin2 leaves to k_skin (joint lists over 64 entries, gathered from L2; every tile on rigs with 3 - 4 weight sets) go the same way, in front of k_skin_fix on the
        // side stream, instead of as a tail behind k_skin2 (in front of k_skin2 on its own stream was measured too: slower).
        const bool lists = cr.fix && (plan.max_pairs > 2 || (plan.kinds & 8u));
        const bool want_fix = lists && !(SKINRT_ABLATE & 32);
        const bool side = (want_fix || others) && conc.side != nullptr && conc.fork != nullptr && conc.join != nullptr;
        if (cr.fix) {
            e = hipMemsetAsync(cr.fix, 0, (size_t)rig.T * 4, s);  // empty work lists
            if (e != hipSuccess) return e;
            const uint32_t slices = (cr.I + kClassifyRounds * 256u - 1) / (kClassifyRounds * 256u);
            if ((uint64_t)rig.T * slices > 0x7fffffffull) return hipErrorInvalidValue;
            const uint32_t words = (rig.M + 63) / 64;  // 0: a rig without morph targets whose long-list tiles go to k_skin_fix
            if ((words && !cr.amask) || (uint64_t)cr.I * words > 0x7fffffffull) return hipErrorInvalidValue;
            if (words) hipLaunchKernelGGL(k_active_mask, dim3((cr.I * words + 3) / 4), dim3(256), 0, s, cr, rig.M, words);
            if (lists) hipLaunchKernelGGL(k_classify, dim3(rig.T * slices), dim3(256), 0, s, rig, cr, slices, words);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        if (side && (e = hipEventRecord(conc.fork, s)) != hipSuccess) return e;  // the pose stage's outputs and the lists are complete here
        auto launch_fix = [&]() -> hipError_t {  // everything that runs beside k_skin2
            hipStream_t fs = s;
            hipError_t fe;
            if (side) {
                if ((fe = hipStreamWaitEvent(conc.side, conc.fork, 0)) != hipSuccess) return fe;
                fs = conc.side;
                if (others) {
                    others_done = true;
                    if ((fe = launch_k_skin(fs)) != hipSuccess) return fe;
                }
            }
            if (!want_fix) return side ? hipEventRecord(conc.join, conc.side) : hipSuccess;
            // one block per (tile, chunk lane).  Lists are uneven (tiles with many pairs hand over most), so there are
            // enough lanes that the longest list is a few chunks per block; blocks past a list's end leave at once
            const uint32_t lanes = std::min((cr.I + kFixChunk - 1) / kFixChunk, std::max(1u, conc.fix_blocks / rig.T));
            const dim3 fgrid(8u * ((rig.T + 7) / 8) * lanes), fblock(256);
#define SKINRT_LAUNCHF(EX, KK, SK, BB)                                                                                                     \
    do {                                                                                                                                   \
        if (plan.fix_pieces == 128) hipLaunchKernelGGL((k_skin_fix<EX, KK, SK, BB, 128>), fgrid, fblock, fix_lds_bytes(128), fs, rig, cr, plan); \
        else if (plan.fix_pieces == 256) hipLaunchKernelGGL((k_skin_fix<EX, KK, SK, BB, 256>), fgrid, fblock, fix_lds_bytes(256), fs, rig, cr, plan); \
        else hipLaunchKernelGGL((k_skin_fix<EX, KK, SK, BB, 384>), fgrid, fblock, fix_lds_bytes(384), fs, rig, cr, plan);                 \
    } while (0)
#define SKINRT_LAUNCHFA(KK, SK, BB)                                                        \
    do {                                                                                   \
        if (plan.exact == 1) SKINRT_LAUNCHF(1, KK, SK, BB);                                \
        else if (plan.exact == 2) SKINRT_LAUNCHF(2, KK, SK, BB);                           \
        else SKINRT_LAUNCHF(0, KK, SK, BB);                                                \
    } while (0)
            // instances morphed per read of a pair's deltas (see SKINRT_FIX_BATCH): 4 for one weight set, unless whole
            // tiles are handed over (kind 3) — then, and for two weight sets, 2
            if (rig.K == 1 && (plan.kinds & 8u)) SKINRT_LAUNCHFA(1, false, 2);
            else if (rig.K == 1) SKINRT_LAUNCHFA(1, false, 4);
            else if (rig.skip_sets) SKINRT_LAUNCHFA(2, true, 2);
            else SKINRT_LAUNCHFA(2, false, 2);
#undef SKINRT_LAUNCHFA
#undef SKINRT_LAUNCHF
            if ((fe = hipGetLastError()) != hipSuccess) return fe;
            if (side) return hipEventRecord(conc.join, conc.side);
            return hipSuccess;
        };
        const bool beside = want_fix || (side && others);
        if (beside && !conc.skin2_first && (e = launch_fix()) != hipSuccess) return e;
        const SkinPlan& whole_plan = plan;
        auto launch2 = [&](const SkinPlan& plan) -> hipError_t {  // (`plan`: this launch's — the macros below name it)
        const dim3 grid(8u * plan.tiles_per_chunk2 * plan.ranges_per_chunk2), block(kSkin2Threads);
#define SKINRT_LAUNCH2(EX, KK, LL, SK)                                                                                     \
    do {                                                                                                                   \
        if (plan.jt2 > 21) {  /* long lists: one instantiation for dense and compacted pairs; over 64 KiB of LDS wants to be announced */ \
            if (plan.lds_bytes2 > 64u * 1024u) {                                                                          \
                const hipError_t ae = hipFuncSetAttribute((const void*)k_skin2<EX, KK, LL, SK, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes2); \
                if (ae != hipSuccess) return ae;                                                                           \
            }                                                                                                              \
            hipLaunchKernelGGL((k_skin2<EX, KK, LL, SK, true, true>), grid, block, plan.lds_bytes2, s, rig, cr, plan);                  \
        }                                                                                                                  \
        else if (rig.sparse_pairs) hipLaunchKernelGGL((k_skin2<EX, KK, LL, SK, true, false>), grid, block, plan.lds_bytes2, s, rig, cr, plan); \
        else hipLaunchKernelGGL((k_skin2<EX, KK, LL, SK, false, false>), grid, block, plan.lds_bytes2, s, rig, cr, plan);   \
    } while (0)
#define SKINRT_LAUNCH2K(EX, LL)                                                            \
    do {                                                                                   \
        if (rig.K == 1) SKINRT_LAUNCH2(EX, 1, LL, false);                                  \
        else if (rig.skip_sets) SKINRT_LAUNCH2(EX, 2, LL, true);                           \
        else SKINRT_LAUNCH2(EX, 2, LL, false);                                             \
    } while (0)
        if (plan.look2 == 1) { if (plan.exact == 1) SKINRT_LAUNCH2K(1, 1); else if (plan.exact == 2) SKINRT_LAUNCH2K(2, 1); else SKINRT_LAUNCH2K(0, 1); }
        else { if (plan.exact == 1) SKINRT_LAUNCH2K(1, 2); else if (plan.exact == 2) SKINRT_LAUNCH2K(2, 2); else SKINRT_LAUNCH2K(0, 2); }
#undef SKINRT_LAUNCH2K
#undef SKINRT_LAUNCH2
        return hipGetLastError();
        };
        if (!whole_plan.split2) e = launch2(whole_plan);
        else {
            // two launches, each over its own tiles with the ring its class needs; the long lists first (fewer, slower blocks:
            // not the ones to end the stage with)
            for (int cls = 1; cls >= 0 && e == hipSuccess; --cls) {
                SkinPlan pc = whole_plan;
                pc.list2_first = cls ? rig.order2_short : 0u;
                pc.list2_n = cls ? rig.order2_long : rig.order2_short;
                if (pc.list2_n == 0) continue;
                if (!cls) { pc.jt2 = whole_plan.jt2s; pc.look2 = whole_plan.look2s; pc.lds_bytes2 = whole_plan.lds_bytes2s; }
                const uint32_t tpc = (pc.list2_n + pc.tile_chunks - 1) / pc.tile_chunks;
                pc.run2 = std::min<uint32_t>(pc.run2, std::max<uint32_t>(1u, tpc));
                pc.tiles_per_chunk2 = (tpc + pc.run2 - 1) / pc.run2 * pc.run2;
                e = launch2(pc);
            }
        }
        if (e == hipSuccess && beside && conc.skin2_first) e = launch_fix();
        if (side && beside) {  // join even after a failed launch: the side stream's work may already be enqueued
            const hipError_t je = hipStreamWaitEvent(s, conc.join, 0);
            if (e == hipSuccess) e = je;
        }
    }
    const bool need_k_skin = others || (!plan.v2 && (plan.kinds & 4u));
    if (e == hipSuccess && need_k_skin && !others_done) e = launch_k_skin(s);
    return e;
}

#ifdef SKINRT_STAMPS
void debug_dump_stamps() {
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof h) != hipSuccess) return;
    static const char* names[16] = {"C inst-top(act read)", "C morph DMA wait", "C morph", "C skin (to last store)", "C aabb",
                                    "C barrier wait", "", "", "L stage", "L flush bounds", "L barrier wait",
                                    "", "", "", "", ""};
    double tot = 0;
    for (int q = 0; q < 12; ++q) tot += (double)h[q];
    for (int q = 0; q < 12; ++q)
        if (h[q]) printf("stamp %-22s %6.2f %%  %8.1f Mcycles\n", names[q], 100.0 * (double)h[q] / tot, (double)h[q] * 1e-6);
    fflush(stdout);
    for (int q = 0; q < 16; ++q) h[q] = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), h, sizeof h);
}
#endif

// What this translation unit was compiled with (skinrt_build_flags): anything but 0 is a measurement build.
uint32_t skin_build_flags() {
    uint32_t f = 0;
    if (SKINRT_ABLATE != 0 || SKINRT_FIX_ABLATE != 0) f |= SKINRT_BUILD_ABLATE;
    if (SKINRT_EXP != 0) f |= SKINRT_BUILD_EXP;
#ifdef SKINRT_STAMPS
    f |= SKINRT_BUILD_STAMPS;
#endif
    if (SKINRT_RANGE2 != 80 || SKINRT_MORPH_SLOTS != 2 || SKINRT_INFLIGHT != 2 || SKINRT_BLOCKS_K1 != 5 || SKINRT_ORDER_ROT != 23u ||
        SKINRT_STORE_POLICY != 1 || sizeof(SKINRT_PAL_MOD) != 1 || SKINRT_SPARSE_SLOTS != 1 || SKINRT_FIX_BATCH != 0 || SKINRT_RANGE_ORDER != 0)
        f |= SKINRT_BUILD_TUNING;
    return f;
}

hipError_t launch_aabb_reduce(const RigDev& rig, const CrowdDev& cr, hipStream_t s) {
    if (cr.I == 0) return hipSuccess;
    hipLaunchKernelGGL(k_aabb_reduce, dim3(cr.I), dim3(kAabbThreads), 0, s, rig, cr);
    return hipGetLastError();
}

hipError_t launch_fill_morph_weights(float* mw, const float* defaults, uint32_t I, uint32_t M, hipStream_t s) {
    const uint64_t n = (uint64_t)I * M;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fill_mw, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, mw, defaults, n, M);
    return hipGetLastError();
}

}  // namespace skinrt
