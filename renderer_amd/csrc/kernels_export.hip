// kernels_export.hip — the crowd's outputs rewritten on the device into the byte layouts the reference's
// render core uploads: interleaved skin vertices and the 128-entry SkinMatrix palette.
// Pure data movement (HBM-bound): nothing is recomputed, every float is copied bit for bit.
//
//   ColorSkinVertex   76 B = position f32x3 | color f32x4      | normal f32x3 | tangent f32x3 | joint_index u16x4 | joint_weight f32x4
//   TextureSkinVertex 68 B = position f32x3 | tex_coords f32x2 | normal f32x3 | tangent f32x3 | joint_index u16x4 | joint_weight f32x4
//     (renderer/src/renderer/buffer/vertex.rs:60-69, :82-91)
//   SkinMatrix       112 B = transform [[f32;4];4] column-major | normal [[f32;4];3] columns padded with 0
//     (renderer/src/renderer/uniform/skin.rs:12-17, :26-37), 128 per instance, entries >= J zero (:41-50)
#include "device_types.hpp"

namespace skinrt {

namespace {

// The exported vertices are a span [v_first, v_first + V) of every instance's Vall vertices (one section of the rig, or
// the whole mesh); pos / nrm / tan / joints0 / weights0 arrive already offset to the span's first vertex.
struct ExportSrc {
    const float* pos;        // [I][Vall][3] + v_first * 3
    const float* nrm;
    const float* tan;
    const float* attr;       // [V][na] or nullptr (zeros)
    const uint32_t* joints0; // [V][2]: four u16 joint indices of weight set 0, as stored by the loader
    const float* weights0;   // [V][4]
    uint32_t Vall;           // vertices per instance in pos / nrm / tan
};

// Block = 256 vertices x a group of kExportGroup consecutive instances.  The shared per-vertex tables (caller's colour /
// texture coordinates, joint indices, weights) are scattered into the records' places in LDS ONCE per block; per instance
// only the nine dwords of position / normal / tangent are rewritten (read as coalesced dword rows, the next instance's
// already in flight while this one's records are written out), and the finished 256 records — one contiguous run of the
// destination — go out as 16-byte streaming stores.  (History: every destination dword looked up on its own, no LDS:
// 1.8 TB/s of read + written bytes; one instance per block, all fields re-scattered: 3.4-3.7 TB/s.)
constexpr uint32_t kExportGroup = 8;
template <uint32_t NA>
__global__ __launch_bounds__(256) void k_export_interleaved(const ExportSrc src, uint32_t V, uint32_t tiles, uint32_t first, uint32_t count, uint32_t* dst) {
    constexpr uint32_t S = 3 + NA + 3 + 3 + 2 + 4;
    __shared__ __align__(16) uint32_t rec[256 * S];
    const uint32_t g = blockIdx.x / tiles, t = blockIdx.x - g * tiles, tid = threadIdx.x;
    const uint32_t v0 = t * 256, nv = min(256u, V - v0);
    const uint32_t k0 = g * kExportGroup, k1 = min(count, k0 + kExportGroup);
    // fields in dwords: position 0-2 | attr 3.. | normal 3+NA.. | tangent 6+NA.. | joint_index 9+NA (2) | joint_weight 11+NA (4)
#pragma unroll
    for (uint32_t r = 0; r < NA; ++r) {
        const uint32_t d = tid + 256u * r;
        if (d < nv * NA) rec[(d / NA) * S + 3 + d % NA] = src.attr ? __float_as_uint(src.attr[(size_t)v0 * NA + d]) : 0u;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t d = tid + 256u * r;
        if (d < nv * 2) rec[(d / 2) * S + 9 + NA + d % 2] = src.joints0[(size_t)v0 * 2 + d];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t d = tid + 256u * r;
        if (d < nv * 4) rec[(d / 4) * S + 11 + NA + d % 4] = __float_as_uint(src.weights0[(size_t)v0 * 4 + d]);
    }
    const uint32_t field[3] = {0u, 3u + NA, 6u + NA};
    uint32_t slot[3];  // where this thread's dword d = tid + 256 r of an array goes inside a record run (without the field offset)
#pragma unroll
    for (int r = 0; r < 3; ++r) { const uint32_t d = tid + 256u * r; slot[r] = (d / 3) * S + d % 3; }
    float x[3][3];
    auto fetch = [&](uint32_t k) {
        const size_t inst_off = (size_t)(first + k) * src.Vall * 3 + (size_t)v0 * 3;
        const float* arrays[3] = {src.pos + inst_off, src.nrm + inst_off, src.tan + inst_off};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const uint32_t d = tid + 256u * r;
                x[a][r] = d < nv * 3 ? __builtin_nontemporal_load(arrays[a] + d) : 0.0f;
            }
    };
    if (k0 < k1) fetch(k0);
    for (uint32_t k = k0; k < k1; ++k) {
        if (k != k0) __syncthreads();  // the previous instance's records have been read out of LDS
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int r = 0; r < 3; ++r)
                if (tid + 256u * r < nv * 3) rec[slot[r] + field[a]] = __float_as_uint(x[a][r]);
        if (k + 1 < k1) fetch(k + 1);  // in flight while this instance's records go out
        __syncthreads();
        const size_t out0 = ((size_t)k * V + v0) * S;  // dwords; v0 * S is a multiple of 4, k * V * S need not be
        const uint32_t nd = nv * S;
        uint32_t* out = dst + out0;
        if ((out0 & 3u) == 0) {
            // streaming stores: the records are written once and read by somebody else's kernels
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            for (uint32_t d = tid * 4; d + 4 <= nd; d += 1024) __builtin_nontemporal_store(*reinterpret_cast<const u4*>(rec + d), reinterpret_cast<u4*>(out + d));
            if (tid < (nd & 3u)) out[(nd & ~3u) + tid] = rec[(nd & ~3u) + tid];
        } else {
            for (uint32_t d = tid; d < nd; d += 256) out[d] = rec[d];
        }
    }
}

// One thread per (instance, palette slot): 24 floats in (joint rows, packed normal rows, padding), 28 floats out.
// The skin's joints are entries [j_first, j_first + J) of every instance's P palette entries; an instance's record has
// `slots` SkinMatrix entries (128 = the reference's uniform array; any count >= J for a storage buffer), those >= J zero.
__global__ __launch_bounds__(256) void k_export_skin_uniform(const float* palette, uint32_t P, uint32_t j_first, uint32_t J, uint32_t slots,
                                                             uint32_t first, uint32_t count, float4* dst) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (uint64_t)count * slots) return;
    const uint32_t k = (uint32_t)(g / slots), j = (uint32_t)(g - (uint64_t)k * slots);
    float4* o = dst + (size_t)g * 7;
    if (j >= J) {
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int q = 0; q < 7; ++q) o[q] = z;
        return;
    }
    const float4* e = reinterpret_cast<const float4*>(palette + ((size_t)(first + k) * P + j_first + j) * kPalFloats);
    const float4 r0 = e[0], r1 = e[1], r2 = e[2], n0 = e[3], n1 = e[4], n2 = e[5];
    o[0] = make_float4(r0.x, r1.x, r2.x, 0.0f);  // column 0 of the 4x4, last row 0,0,0,1
    o[1] = make_float4(r0.y, r1.y, r2.y, 0.0f);
    o[2] = make_float4(r0.z, r1.z, r2.z, 0.0f);
    o[3] = make_float4(r0.w, r1.w, r2.w, 1.0f);
    // normal matrix columns, pad_vec3; the palette holds the nine entries packed row by row (n0 = n00 n01 n02 n10, ...)
    o[4] = make_float4(n0.x, n0.w, n1.z, 0.0f);
    o[5] = make_float4(n0.y, n1.x, n1.w, 0.0f);
    o[6] = make_float4(n0.z, n1.y, n2.x, 0.0f);
}

}  // namespace

hipError_t launch_export_interleaved(const CrowdDev& cr, uint32_t Vall, uint32_t v_first, uint32_t V, const uint32_t* joints0,
                                     const float* weights0, const float* attr, uint32_t layout, uint32_t first, uint32_t count, void* dst,
                                     hipStream_t s) {
    if (count == 0 || V == 0) return hipSuccess;
    if ((uint64_t)first + count > cr.I || (uint64_t)v_first + V > Vall || (reinterpret_cast<uintptr_t>(dst) & 15u)) return hipErrorInvalidValue;
    const size_t vo = (size_t)v_first;
    const ExportSrc src{cr.pos + vo * 3, cr.nrm + vo * 3, cr.tan + vo * 3, attr, joints0 + vo * 2, weights0 + vo * 4, Vall};
    const uint32_t tiles = (V + 255) / 256;
    const uint32_t groups = (count + kExportGroup - 1) / kExportGroup;
    if ((uint64_t)tiles * groups > 0x7fffffffull) return hipErrorInvalidValue;
    const dim3 grid(tiles * groups), block(256);
    if (layout == SKINRT_LAYOUT_COLOR_SKIN)
        hipLaunchKernelGGL(k_export_interleaved<4>, grid, block, 0, s, src, V, tiles, first, count, static_cast<uint32_t*>(dst));
    else
        hipLaunchKernelGGL(k_export_interleaved<2>, grid, block, 0, s, src, V, tiles, first, count, static_cast<uint32_t*>(dst));
    return hipGetLastError();
}

hipError_t launch_export_skin_uniform(const CrowdDev& cr, uint32_t P, uint32_t j_first, uint32_t J, uint32_t slots, uint32_t first, uint32_t count,
                                      void* dst, hipStream_t s) {
    if (count == 0 || slots == 0) return hipSuccess;
    const uint64_t threads = (uint64_t)count * slots;
    if ((uint64_t)first + count > cr.I || J > slots || (uint64_t)j_first + J > P || (reinterpret_cast<uintptr_t>(dst) & 15u) ||
        (threads + 255) / 256 > 0x7fffffffull)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_export_skin_uniform, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, cr.palette, P, j_first, J, slots, first, count,
                       static_cast<float4*>(dst));
    return hipGetLastError();
}

}  // namespace skinrt
