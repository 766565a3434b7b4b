// device_types.hpp — kernel argument blocks shared by the API and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "rig_compile.hpp"

namespace skinrt {

// Device copies of CompiledRig's tables (rig_compile.hpp documents each).
struct RigDev {
    uint32_t N, J, V, K, M, clips, Cmax, levels, T;
    const int32_t* parent;
    const float* rest_trs;
    const float* rest_mat;
    const uint8_t* node_matrix;
    const uint32_t* level_start;
    const uint32_t* level_nodes;
    const int32_t* node_chan;
    const ChanRec* chans;
    const uint32_t* clip_chan_count;
    const WeightSrc* weight_src;       // [clips][M] the SKINRT_PATH_WEIGHTS channel that drives target m in that clip, or chan = -1
    uint32_t weight_channel_clips;     // clips that have one (0: CrowdDev::mw_in is never used)
    const float* clip_length;          // [clips] largest key time (skinrt_advance_clocks)
    const float* times;
    const float* values;
    const uint32_t* joint_node;
    const float* ibm;
    const float* base;
    const float* wts;
    const uint16_t* lidx;
    const uint32_t* tile_joint_start;
    const uint32_t* tile_joints;
    const uint8_t* tile_lds;
    const uint32_t* tile_order;   // [T] tile indices by kind: kind 0 (order_n0 of them), kind 1 (order_n1), then kinds 2 and 3 — the loader-wave kernel's grid
    const uint32_t* tile_order2;  // CompiledRig::split_joints != 0: the kind-2 tiles, short lists first (order2_short of them), then the long ones (order2_long)
    uint32_t order2_short, order2_long;
    uint32_t order_n0, order_n1;  //     covers ITS tiles only (a grid over all tiles spent 1.9 ms dispatching empty blocks for two real tiles)
    const uint32_t* tile_pair_start;
    const MorphPair* pairs;
    const float* mblock;
    uint32_t skip_sets;  // CompiledRig::skip_sets
    uint32_t sparse_pairs;  // the rig has compacted (tile, target) pairs and k_skin2 takes them from its slots (its SPARSE instantiations)
};

// Per-crowd state and outputs (HBM layout: DESIGN.md §3).
struct CrowdDev {
    uint32_t I;
    const float* tau;      // [I]
    const uint32_t* clip;  // [I]
    const float* mw;       // [I][M] morph weights the skin stage uses
    const float* mw_in;    // [I][M] weights set by the host / rig defaults; differs from mw only for rigs with weights
                           // channels, where the pose stage writes mw = channel sample or mw_in (k_sample_weights)
    const float* root;     // [I][12] rows, or nullptr
    float* world;          // [I][N][12]
    float* palette;        // [I][J][24]
    uint32_t* keyidx;      // [I][Cmax]
    float* pos;            // [I][V][3]
    float* nrm;            // [I][V][3]
    float* tan;            // [I][V][3]
    float* aabb_part;      // [I][T][6]
    float* aabb;           // [I][6]
    // (tile, instance) pairs the self-staging skin kernel hands to k_skin_fix (3+ active morph
    // pairs, compacted pairs, tiles with more than 32 pairs), kept per tile:
    // fix[tile] = number of instances handed over, fix_inst[tile * I + k] = the k-th of them
    uint32_t* fix;         // [T] counters, or nullptr (rig without morph targets)
    uint32_t* fix_inst;    // [T][I]
    uint64_t* amask;       // [I][ceil(M / 64)] bit t = morph weight t is non-zero (rebuilt by every skin stage), or nullptr
};

constexpr uint32_t kSkin2MaxLds = 80 * 1024;  // dynamic LDS of a k_skin2 block: at least two blocks per CU (gfx950: 160 KiB per CU, a workgroup may take all of it)

// How the skinning grid is cut (computed once per crowd by plan_skin()).
struct SkinPlan {
    uint32_t range;     // instances per block (one block = one vertex tile x one instance range)
    uint32_t ranges;    // ceil(I / range)
    // XCD-aware grid: 8 = tile_chunks x range chunks; XCD x owns tile chunk x % tile_chunks
    // and range chunk x / tile_chunks
    uint32_t tile_chunks;       // 1, 2, 4 or 8
    uint32_t tiles_per_chunk;   // ceil(T / tile_chunks)
    uint32_t ranges_per_chunk;  // ceil(ranges / (8 / tile_chunks))
    uint32_t batch;     // instances whose palettes are LDS-resident at once
    uint32_t jt_shift;  // log2 of the padded tile joint-list length
    uint32_t lds_tiles; // tiles whose joint list is LDS-staged (the rest gather from L2)
    uint32_t pair_gshift; // log2 of the lane-group size the loader uses to compact one instance's pairs
    uint32_t max_pairs; // largest number of (tile, target) morph pairs in one tile
    uint32_t lds_bytes; // dynamic LDS per block
    uint32_t exact;     // 0: FMA, 1: contraction-free arithmetic (bit-equal to the fp32 oracle), 2: positions contraction-free, normals / tangents FMA
    uint32_t normalize; // shader.wgsl:217-218 variant
    uint32_t aabb;      // write per-(instance,tile) bounds
    uint32_t debug;     // unused
    // self-staging kernel (k_skin2): tiles of kind 2 when v2 != 0
    uint32_t v2;                 // 1: kind-2 tiles go to k_skin2, 0: to k_skin
    uint32_t range2;             // instances per k_skin2 block (<= 128)
    uint32_t ranges2;            // ceil(I / range2)
    uint32_t ranges_per_chunk2;  // ceil(ranges2 / (8 / tile_chunks))
    uint32_t lds_bytes2;         // dynamic LDS per k_skin2 block
    uint32_t kinds;              // bit k set: the rig has tiles of kind k
    uint32_t jt2;                // k_skin2: joints per palette slot = longest joint list among its tiles (<= kSkin2MaxJoints)
    uint32_t fix_pieces;         // k_skin_fix: 16-byte palette pieces its staging buffers hold (128; 256 where a tile of kind 2 / 3 has more than 21 joints, 384 where more than 42)
    uint32_t run2;               // k_skin2: an XCD's tiles come in runs of run2 consecutive tiles, the runs dealt round-robin over
                                 // the tile chunks (1: fully interleaved; tiles_per_chunk2: one contiguous chunk per XCD)
    uint32_t tiles_per_chunk2;   // tiles_per_chunk rounded up to whole runs
    uint32_t look2;              // k_skin2: palette LDS-DMA runs this many instances ahead (1 or 2); ring = look2 + 1 slots
    // k_skin2 as two launches (split2 != 0; CompiledRig::split_joints): jt2 / look2 / lds_bytes2 above are the LONG class's (and the
    // whole kernel's where there is no split), these the short class's; list2_n != 0: the launch covers rig.tile_order2[list2_first ..
    // + list2_n) instead of all tiles (set per launch by launch_skin)
    uint32_t split2, jt2s, look2s, lds_bytes2s;
    uint32_t list2_first, list2_n;
};

// Side stream + fork / join events for the part of the skin stage that runs next to k_skin2 (k_skin_fix);
// all null: one after the other on the caller's stream.
struct SkinConcurrency {
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    uint32_t skin2_first = 0;     // enqueue k_skin2 before k_skin_fix
    uint32_t fix_blocks = 32768;  // k_skin_fix grid: about this many blocks
};

enum PosePath : uint32_t { kPoseWave = 0, kPoseSplit = 1 };

// kernels_pose.hip
// local_t: [N][I][12] scratch for the lane-per-instance channel sampling, or nullptr (sampling inside the pose kernels)
hipError_t launch_pose(const RigDev& rig, const CrowdDev& cr, uint32_t cubic_quirk, PosePath path, float* local_t, hipStream_t s);
hipError_t launch_sample_weights(const RigDev& rig, const CrowdDev& cr, float* mw_out, uint32_t quirk, hipStream_t s);
// tau[i] = fmodf(tau[i] + dt * rate[i], clip_length[clip[i]]) (skinrt_advance_clocks)
hipError_t launch_advance_clocks(const RigDev& rig, float* tau, const uint32_t* clip, uint32_t I, float dt, const float* rate, hipStream_t s);
// InstanceUniform records (uniform/transform.rs:8-14) of the listed nodes for instances [first, first + count)
hipError_t launch_export_instance_uniform(const CrowdDev& cr, uint32_t N, uint32_t first, uint32_t count, const uint32_t* nodes,
                                          uint32_t node_count, uint32_t stride, void* dst, hipStream_t s);
// kernels_skin.hip
hipError_t launch_skin(const RigDev& rig, const CrowdDev& cr, const SkinPlan& plan, hipStream_t s, const SkinConcurrency& conc);
hipError_t launch_aabb_reduce(const RigDev& rig, const CrowdDev& cr, hipStream_t s);
uint32_t skin_build_flags();  // SKINRT_BUILD_* bits of the skin kernels' translation unit
// kernels_export.hip: reference-layout records written on the device (instances [first, first + count))
// vertices [v_first, v_first + v_count) of every instance's V (a section, or the whole mesh); attr counts from v_first
hipError_t launch_export_interleaved(const CrowdDev& cr, uint32_t V, uint32_t v_first, uint32_t v_count, const uint32_t* joints0,
                                     const float* weights0, const float* attr, uint32_t layout, uint32_t first, uint32_t count, void* dst,
                                     hipStream_t s);
// palette entries [j_first, j_first + j_count) of every instance's P entries (one skin's joints) as `slots` >= j_count
// SkinMatrix records per instance (128: the reference's uniform array)
hipError_t launch_export_skin_uniform(const CrowdDev& cr, uint32_t P, uint32_t j_first, uint32_t j_count, uint32_t slots, uint32_t first,
                                      uint32_t count, void* dst, hipStream_t s);
// kernels_geom.hip: calculate_tangent on the device (load time)
size_t tangent_work_bytes(uint32_t vertices, uint32_t corners);
hipError_t tangent_static_setup(int device);                             // once per device, that device current
hipError_t tangent_static_area(int device, void** ptr, size_t* bytes);   // the device's copy of the static work area
// run `launch(arg)` (which enqueues on `s`) as the next user of the device's static work area: ordered behind the previous one
hipError_t tangent_static_use(int device, hipStream_t s, hipError_t (*launch)(void*), void* arg);
hipError_t launch_tangents(const float* pos, const uint32_t* indices, uint32_t vertices, uint32_t corners, void* work, size_t work_bytes,
                           float* out, hipStream_t s);
// the corner buckets of a mesh, sorted once (inside `work`), and calculate_tangent per (instance, vertex) of a crowd from them
hipError_t launch_topology(const uint32_t* indices, uint32_t vertices, uint32_t corners, void* work, size_t work_bytes,
                           const uint32_t** out_start, const uint32_t** out_adj, hipStream_t s);
hipError_t launch_crowd_normals(const float* pos_all, const uint32_t* indices, const uint32_t* start, const uint32_t* adj,
                                uint32_t vertices, uint32_t first, uint32_t count, float* out_all, hipStream_t s);
// placement.hip: the three output arrays, steered over the HBM's regions and measured
hipError_t allocate_outputs(int device, size_t bytes_each, uint32_t V, uint32_t I, uint32_t mode, uint32_t max_filler_gib, hipStream_t s,
                            void** pos, void** nrm, void** tan, skinrt_placement_info* info);
hipError_t launch_fill_morph_weights(float* mw, const float* defaults, uint32_t I, uint32_t M, hipStream_t s);

}  // namespace skinrt
