// Microbenchmark: a TRANSPOSED skin kernel for C3's shape — lane = instance instead of lane = vertex.
//
// The library's k_skin2 keeps a VERTEX per lane and fetches, per instance, the palette entries of that vertex's
// joints from LDS: 4 influences x 84 B per (vertex, instance), and those LDS reads are what does not hide under the
// store stream.  Here a wave owns 64 INSTANCES of one tile: the tile's <= 8 joints' palette entries of the
// lane's instance sit in REGISTERS for the wave's life, a vertex's weights and joint indices are wave-uniform (scalar
// loads, one 64-byte record per vertex), the joint index selects registers through a scalar branch, and the results
// reach the [instance][vertex][3] output rows through an LDS transposition (lane = instance writes, lane = vertex reads).
// A wave produces ONE of the three outputs (role 0 position, 1 normal, 2 tangent), so the transposition buffer is
// 64 x NV x 12 B.  Questions: (1) is the store stream of NV-vertex pieces (192 B for NV = 16) as fast as k_skin2's
// 768-byte pieces; (2) what does the whole thing take against the store floor.
//   hipcc --offload-arch=gfx950 -O3 -o skin_t skin_t.hip && ./skin_t
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float float3v __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CONSTANT __attribute__((address_space(4)))
template <class T> __device__ __forceinline__ const CONSTANT T* as_const(const T* p) { return (const CONSTANT T*)(uintptr_t)p; }
// S += w * P with w in an SGPR; asm so that the compiler cannot merge the eight joint cases into one block fed by copies
__device__ __forceinline__ void fmac_s(float& s, float w, float p) { asm("v_fmac_f32 %0, %1, %2" : "+v"(s) : "s"(w), "v"(p)); }
__device__ __forceinline__ void mul_s(float& s, float w, float p) { asm("v_mul_f32 %0, %1, %2" : "=v"(s) : "s"(w), "v"(p)); }
typedef uint32_t u16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void store3(float* base, unsigned byte_off, float x, float y, float z) {
    float3v d = {x, y, z};
    asm volatile("global_store_dwordx3 %0, %1, %2 nt\n\ts_nop 1" : : "v"(byte_off), "v"(d), "s"(base) : "memory");
}

// ---- the old pattern, store only (tools/ubench/store_phase.hip, rot 23), for the floor on the same arrays ----
__global__ __launch_bounds__(256) void k_old(float* pos, float* nrm, float* tan, unsigned V, unsigned T, unsigned I, unsigned R, unsigned ranges, unsigned fmas) {
    const unsigned tpx = (T + 7) / 8, xcd = blockIdx.x & 7, kb = blockIdx.x >> 3;
    const unsigned tile = (kb % tpx) * 8 + xcd, rng = kb / tpx;
    if (tile >= T || rng >= ranges) return;
    const unsigned v = tile * 256 + threadIdx.x, i0 = rng * R, n = min(I, i0 + R) - i0;
    const unsigned start = __builtin_amdgcn_readfirstlane((tile * 23u) % n);
    float x = (float)threadIdx.x, y = x + 1.f, z = x + 2.f;
    const unsigned olb = v * 12;
    for (unsigned q = 0; q < n; ++q) {
        unsigned i = start + q; if (i >= n) i -= n;
        for (unsigned f = 0; f < fmas; ++f) { x = __builtin_fmaf(x, 1.0001f, y); y = __builtin_fmaf(y, 0.9999f, z); z = __builtin_fmaf(z, 1.0002f, x); }
        const size_t o = (size_t)(i0 + i) * V * 3;
        if (v < V) { store3(pos + o, olb, x, y, z); store3(nrm + o, olb, y, z, x); store3(tan + o, olb, z, x, y); }
    }
}

struct Args {
    const float* pal;       // [I][J][24]
    const uint32_t* vtx;    // [T][256][16]: p(3) n(3) t(3) w(4) li(2 x u16 packed) pad
    const uint32_t* tjoint; // [T][8] joint ids of the tile's list
    const float* mw;        // [I][M]
    const uint32_t* tpair;  // [T][NP] target of the tile's q-th pair
    const float* delta;     // [T][NP][256][12]: dp(3) dn(3) dt(3) pad
    float* out0; float* out1; float* out2;
    float* bounds;          // [I][T][6]
    unsigned V, T, I, J, M, np;
    unsigned mode;          // 0 store only (no palette, no arithmetic), 1 full, 3 full with fixed joints
};

template <int NV, int ROLE>
__device__ __forceinline__ void run(const Args& a, unsigned tile, unsigned ig, float* buf) {
    constexpr int C = ROLE == 0 ? 12 : 9;
    constexpr int PITCH = NV * 3 + 1;
    const unsigned lane = threadIdx.x;
    const unsigned i0 = ig * 64;
    const unsigned inst = min(i0 + lane, a.I - 1);
    float* const out = ROLE == 0 ? a.out0 : (ROLE == 1 ? a.out1 : a.out2);
    float* const obase = out + (size_t)i0 * a.V * 3;
    float P[8][C];
    float wq[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.mode) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned joint = as_const(a.tjoint)[tile * 8 + j];
            const float* e = a.pal + ((size_t)inst * a.J + joint) * 24;
            if (ROLE == 0) {
                const f4 r0 = *reinterpret_cast<const f4*>(e), r1 = *reinterpret_cast<const f4*>(e + 4), r2 = *reinterpret_cast<const f4*>(e + 8);
                const float m[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
#pragma unroll
                for (int x = 0; x < C; ++x) P[j][x] = m[x];
            } else {
                const f4 r0 = *reinterpret_cast<const f4*>(e + 12), r1 = *reinterpret_cast<const f4*>(e + 16);
                const float m[9] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, e[20]};
#pragma unroll
                for (int x = 0; x < C; ++x) P[j][x] = m[x];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if ((unsigned)q < a.np) wq[q] = a.mw[(size_t)inst * a.M + as_const(a.tpair)[tile * a.np + q]];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int x = 0; x < C; ++x) P[j][x] = (float)(lane + j + x);
    }
    float bmin[3] = {1e30f, 1e30f, 1e30f}, bmax[3] = {-1e30f, -1e30f, -1e30f};
    const CONSTANT uint32_t* vt = as_const(a.vtx) + (size_t)tile * 256 * 16;
    const CONSTANT float* dl = as_const(a.delta) + (size_t)tile * a.np * 256 * 12;
    // every palette register has landed before the first store exists: no vector-memory wait inside the loop
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int x = 0; x < C; ++x) asm volatile("" : "+v"(P[j][x]));
    u16v rec = *reinterpret_cast<const CONSTANT u16v*>(vt);
    for (unsigned g = 0; g < 256 / NV; ++g) {
        const unsigned v0 = tile * 256 + g * NV;
        for (unsigned vl = 0; vl < NV; ++vl) {
            float o[3];
            if (a.mode) {
                const u16v rv = rec;
                {   // next vertex's record: requested now, used in the next iteration (the tile's last vertex re-reads itself)
                    const unsigned nx = (a.mode & 8) ? 0u : min(g * NV + vl + 1, 255u);  // mode 8: one record for every vertex (scalar cache always hits)
                    rec = *reinterpret_cast<const CONSTANT u16v*>(vt + nx * 16);
                }
                const uint32_t r[16] = {rv.s0, rv.s1, rv.s2, rv.s3, rv.s4, rv.s5, rv.s6, rv.s7, rv.s8, rv.s9, rv.sa, rv.sb, rv.sc, rv.sd, rv.se, rv.sf};
                float p[3] = {__uint_as_float(r[ROLE * 3 + 0]), __uint_as_float(r[ROLE * 3 + 1]), __uint_as_float(r[ROLE * 3 + 2])};
                for (unsigned q = 0; q < a.np; ++q) {
                    const CONSTANT float* d = dl + ((size_t)q * 256 + g * NV + vl) * 12 + ROLE * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) p[c] = __builtin_fmaf(wq[q], d[c], p[c]);
                }
                float S[C];
#pragma unroll
                for (int k = 0; k < ((a.mode & 32) ? 1 : 4); ++k) {
                    const float w = __uint_as_float(r[9 + k]);
                    const unsigned j = (a.mode & 2) ? (unsigned)(2 * k + 1) : (r[13 + (k >> 1)] >> (16 * (k & 1))) & 0xffffu;  // mode 2: fixed joints, no branch
                    if (k == 0) {
                        switch (j) {
#define CASE(JJ) case JJ: _Pragma("unroll") for (int x = 0; x < C; ++x) mul_s(S[x], w, P[JJ][x]); break;
                            CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) default: _Pragma("unroll") for (int x = 0; x < C; ++x) mul_s(S[x], w, P[7][x]); break;
#undef CASE
                        }
                    } else {
                        switch (j) {
#define CASE(JJ) case JJ: _Pragma("unroll") for (int x = 0; x < C; ++x) fmac_s(S[x], w, P[JJ][x]); break;
                            CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) default: _Pragma("unroll") for (int x = 0; x < C; ++x) fmac_s(S[x], w, P[7][x]); break;
#undef CASE
                        }
                    }
                }
                if (ROLE == 0) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        o[c] = __builtin_fmaf(S[4 * c + 2], p[2], __builtin_fmaf(S[4 * c + 1], p[1], S[4 * c] * p[0])) + S[4 * c + 3];
                        bmin[c] = fminf(bmin[c], o[c]);
                        bmax[c] = fmaxf(bmax[c], o[c]);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) o[c] = __builtin_fmaf(S[3 * c + 2], p[2], __builtin_fmaf(S[3 * c + 1], p[1], S[3 * c] * p[0]));
                }
            } else {
                o[0] = P[0][0] + (float)vl; o[1] = P[1][1]; o[2] = P[2][2];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) buf[lane * PITCH + vl * 3 + c] = o[c];
        }
        // flush: lane = (row, vertex-in-group); NV lanes cover one instance row's NV * 12 contiguous bytes
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
        for (unsigned k = 0; k < ((a.mode & 16) ? 0u : (unsigned)NV); ++k) {
            const unsigned e = k * 64 + lane, row = e / NV, vl = e % NV;
            const float* s = buf + row * PITCH + vl * 3;
            const float x = s[0], y = s[1], z = s[2];
            if (i0 + row < a.I && v0 + vl < a.V) store3(obase, (row * a.V + v0 + vl) * 12, x, y, z);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (ROLE == 0 && a.mode && i0 + lane < a.I) {
        float* b = a.bounds + ((size_t)inst * a.T + tile) * 6;
#pragma unroll
        for (int c = 0; c < 3; ++c) { b[c] = bmin[c]; b[3 + c] = bmax[c]; }
    }
}

template <int NV>
__global__ __launch_bounds__(64) void k_t(const Args a) {
    extern __shared__ float buf[];
    const unsigned tpx = (a.T + 7) / 8, xcd = blockIdx.x & 7, kb = blockIdx.x >> 3;
    unsigned tile = (kb % tpx) * 8 + xcd, rest = kb / tpx;
    unsigned role = rest % 3, ig = rest / 3;
    if (a.mode & 4) {  // instance groups fastest: co-resident waves walk the same tile's vertex records (scalar-cache hits)
        const unsigned groups = (a.I + 63) / 64;
        ig = kb % groups; rest = kb / groups; role = rest % 3; tile = (rest / 3) * 8 + xcd;
    }
    if (tile >= a.T || ig * 64 >= a.I) return;
    if (role == 0) run<NV, 0>(a, tile, ig, buf);
    else if (role == 1) run<NV, 1>(a, tile, ig, buf);
    else run<NV, 2>(a, tile, ig, buf);
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static float frand() { return (float)(rnd() & 0xffff) / 65536.0f; }

int main(int argc, char** argv) {
    const unsigned V = 60000, T = (V + 255) / 256, I = argc > 1 ? atoi(argv[1]) : 10000, J = 200, M = 32, NP = 3;
    const size_t n = (size_t)I * V * 3;
    float *pos, *nrm, *tan;
    CK(hipMalloc(&pos, n * 4)); CK(hipMalloc(&nrm, n * 4)); CK(hipMalloc(&tan, n * 4));
    // synthetic rig
    std::vector<float> pal((size_t)I * J * 24);
    for (auto& x : pal) x = frand() - 0.5f;
    std::vector<uint32_t> vtx((size_t)T * 256 * 16), tj(T * 8), tp(T * NP);
    for (size_t v = 0; v < (size_t)T * 256; ++v) {
        uint32_t* r = &vtx[v * 16];
        for (int c = 0; c < 9; ++c) { float f = frand() * 2 - 1; r[c] = *reinterpret_cast<uint32_t*>(&f); }
        for (int k = 0; k < 4; ++k) { float f = frand() * 0.25f; r[9 + k] = *reinterpret_cast<uint32_t*>(&f); }
        r[13] = (rnd() % 8) | ((rnd() % 8) << 16);
        r[14] = (rnd() % 8) | ((rnd() % 8) << 16);
        r[15] = 0;
    }
    for (unsigned t = 0; t < T; ++t) for (int j = 0; j < 8; ++j) tj[t * 8 + j] = (t * 3 + j * 5) % J;
    for (unsigned t = 0; t < T; ++t) for (unsigned q = 0; q < NP; ++q) tp[t * NP + q] = (t + q * 7) % M;
    std::vector<float> mw((size_t)I * M), dl((size_t)T * NP * 256 * 12);
    for (auto& x : mw) x = (rnd() & 3) ? 0.f : frand();
    for (auto& x : dl) x = frand() * 0.01f;
    float *d_pal, *d_mw, *d_dl, *d_bounds; uint32_t *d_vtx, *d_tj, *d_tp;
    CK(hipMalloc(&d_pal, pal.size() * 4)); CK(hipMemcpy(d_pal, pal.data(), pal.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_vtx, vtx.size() * 4)); CK(hipMemcpy(d_vtx, vtx.data(), vtx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_tj, tj.size() * 4)); CK(hipMemcpy(d_tj, tj.data(), tj.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_tp, tp.size() * 4)); CK(hipMemcpy(d_tp, tp.data(), tp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_mw, mw.size() * 4)); CK(hipMemcpy(d_mw, mw.data(), mw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_dl, dl.size() * 4)); CK(hipMemcpy(d_dl, dl.data(), dl.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_bounds, (size_t)I * T * 6 * 4));
    hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    auto timeit = [&](const char* name, auto&& launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(ea));
            launch();
            CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, ea, eb));
            if (rep) best = std::min(best, ms);
        }
        printf("%-58s : %7.3f ms  %6.2f TB/s of stores\n", name, best, n * 12.0 / best / 1e9);
        fflush(stdout);
    };
    const unsigned R = 80, ranges = (I + R - 1) / R, tpx = (T + 7) / 8;
    for (unsigned fmas : {0u, 37u})
        timeit(fmas ? "old pattern (lane = vertex), stores + 111 dependent FMAs" : "old pattern (lane = vertex), stores only",
               [&] { hipLaunchKernelGGL(k_old, dim3(8 * tpx * ranges), dim3(256), 0, 0, pos, nrm, tan, V, T, I, R, ranges, fmas); });
    Args a{d_pal, d_vtx, d_tj, d_mw, d_tp, d_dl, pos, nrm, tan, d_bounds, V, T, I, J, M, NP, 0};
    const unsigned groups = (I + 63) / 64, grid = 8 * tpx * groups * 3;
    for (unsigned mode : {0u, 1u}) {
        a.mode = mode;
        a.np = NP;
        timeit(mode ? "transposed NV=16 (192-B pieces), full, 3 pairs" : "transposed NV=16 (192-B pieces), stores only",
               [&] { hipLaunchKernelGGL(k_t<16>, dim3(grid), dim3(64), 64 * 49 * 4, 0, a); });
        timeit(mode ? "transposed NV=32 (384-B pieces), full, 3 pairs" : "transposed NV=32 (384-B pieces), stores only",
               [&] { hipLaunchKernelGGL(k_t<32>, dim3(grid), dim3(64), 64 * 97 * 4, 0, a); });
        if (mode) {
            a.np = 0;
            a.mode = 1 | 16;
            timeit("transposed NV=16, no morph pairs, NO FLUSH (nothing stored)", [&] { hipLaunchKernelGGL(k_t<16>, dim3(grid), dim3(64), 64 * 49 * 4, 0, a); });
            a.mode = 1 | 16 | 32;
            timeit("transposed NV=16, no morph pairs, no flush, ONE influence", [&] { hipLaunchKernelGGL(k_t<16>, dim3(grid), dim3(64), 64 * 49 * 4, 0, a); });
            a.mode = 1 | 32;
            timeit("transposed NV=16, no morph pairs, one influence", [&] { hipLaunchKernelGGL(k_t<16>, dim3(grid), dim3(64), 64 * 49 * 4, 0, a); });
            a.mode = 9;
            timeit("transposed NV=16, full, no morph pairs, every vertex reads record 0", [&] { hipLaunchKernelGGL(k_t<16>, dim3(grid), dim3(64), 64 * 49 * 4, 0, a); });
            timeit("transposed NV=32, full, no morph pairs, every vertex reads record 0", [&] { hipLaunchKernelGGL(k_t<32>, dim3(grid), dim3(64), 64 * 97 * 4, 0, a); });
            a.mode = 5;
            timeit("transposed NV=16, full, no morph pairs, instance groups fastest", [&] { hipLaunchKernelGGL(k_t<16>, dim3(grid), dim3(64), 64 * 49 * 4, 0, a); });
            a.np = NP;
            timeit("transposed NV=16, full, 3 pairs, instance groups fastest", [&] { hipLaunchKernelGGL(k_t<16>, dim3(grid), dim3(64), 64 * 49 * 4, 0, a); });
            a.np = 0;
            a.mode = 3;
            timeit("transposed NV=16, full, no morph pairs, fixed joints (no scalar branches)", [&] { hipLaunchKernelGGL(k_t<16>, dim3(grid), dim3(64), 64 * 49 * 4, 0, a); });
            a.mode = 1;
            timeit("transposed NV=16, full, no morph pairs", [&] { hipLaunchKernelGGL(k_t<16>, dim3(grid), dim3(64), 64 * 49 * 4, 0, a); });
            timeit("transposed NV=32, full, no morph pairs", [&] { hipLaunchKernelGGL(k_t<32>, dim3(grid), dim3(64), 64 * 97 * 4, 0, a); });
        }
    }
    return 0;
}
