// rig_compile.hpp — host-side validation + compilation of the flat loader output
// (skinrt_rig_desc) into the tables the HIP kernels read.  Pure host C++; no device calls.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/skinrt.h"

namespace skinrt {

constexpr uint32_t kTileVerts = 256;   // vertices per skinning tile = threads per block
constexpr uint32_t kTileJointCap = 64; // largest tile-local joint list that is LDS-staged
constexpr uint32_t kSkin2MaxJoints = 64;  // longest tile joint list the self-staging kernel takes (= kTileJointCap): 384 16-byte palette pieces = six LDS-DMA instructions per wave
constexpr uint32_t kPalFloats = 24;    // palette entry: 3x4 joint rows, 3x3 normal rows packed, 3 floats of padding
// Joints farther than this from the origin of their bind space make the skinning sums cancel: a fused and an
// unfused multiply-add then differ by a few ulp(extent) ~ 2.4e-7 * extent, which at 4 units reaches the
// 1e-6 absolute parity bar.  Beyond it skinrt_config.exact_arithmetic = 0 selects the contraction-free kernels.
constexpr float kExactBindExtent = 4.0f;

// one animation channel, keyframes in the shared times[] / values[] pools
struct ChanRec {
    uint32_t key_count;
    uint32_t time_off;   // into times[]
    uint32_t value_off;  // into values[]
    uint16_t local;      // index of the channel inside its clip (key_index column)
    uint8_t interp;      // SKINRT_INTERP_*
    uint8_t path;        // SKINRT_PATH_*
};

// one morph target's effect on one vertex tile
struct MorphPair {
    // float offset of the pair's region in mblock[] (multiple of 4 floats):
    //   dense : [4 waves][9 attrs][64 lanes] — each wave's 576 floats are contiguous so the wave
    //           can pull its share into LDS with three 16-byte LDS-DMA loads
    //   sparse: [256 B slot map: slot+1 per tile lane, 0 = untouched][cnt][9], cnt < kMorphDenseMin
    uint32_t data_off;
    uint32_t floats;    // size of the region in floats (sparse: <= kMorphWaveFloats)
    uint16_t target;
    uint16_t dense;
    uint32_t wave_mask;  // bit w: some vertex of wave w (lanes 64w .. 64w+63) has a non-zero delta; other waves skip the pair
};
#ifndef SKINRT_RANGE2
#define SKINRT_RANGE2 80
#endif
constexpr uint32_t kSkin2Range = SKINRT_RANGE2;         // instances per block of the self-staging skinning kernel
constexpr uint32_t kMorphWaveFloats = 576;    // one wave's share of a dense pair = one LDS slot

constexpr uint32_t kMorphDenseMin = 57;       // touched vertices from which a pair is stored dense
                                              // (a sparse region, 64 + 56*9 floats, still fits a slot)
#ifndef SKINRT_SPARSE_SLOTS
#define SKINRT_SPARSE_SLOTS 1  // see kernels_skin.hip
#endif
#ifndef SKINRT_MORPH_SLOTS
#define SKINRT_MORPH_SLOTS 2
#endif
constexpr uint32_t kMorphSlots = SKINRT_MORPH_SLOTS;  // LDS slots per wave: active pairs streamed per instance

// LDS of one k_skin2 block (kernels_skin.hip: block tables, then per wave {palette ring of look + 1 slots, morph slots};
// the prologue's scratch is overlaid on the wave areas) and the blocks per CU it allows (at most 7: the register budget).
// Shared by the plan (api.hip) and by the rig compiler, which sizes the ring by the joint lists it lets the kernel take.
inline uint32_t skin2_lds_bytes(uint32_t tile_joints, uint32_t look, bool morph, uint32_t max_pairs) {
    const uint32_t slots = morph ? kMorphSlots * kMorphWaveFloats * 4 : 0;
    const uint32_t tables = kSkin2Range * (16 + 24) + max_pairs * 16;  // headers, bounds cells, pair records
    const uint32_t scratch = 32 + kSkin2Range * (4 + 2 + 2 + 8 + 8) + 64;
    const uint32_t waves = 4 * ((look + 1) * 96u * tile_joints + slots);
    return tables + (waves > scratch ? waves : scratch);
}
inline uint32_t skin2_blocks_per_cu(uint32_t lds_bytes) {
    const uint32_t b = (160u * 1024) / (lds_bytes ? lds_bytes : 1u);
    return b < 7 ? b : 7;
}

// who drives morph target m of an instance playing a given clip: a SKINRT_PATH_WEIGHTS channel, or nobody (chan < 0)
struct WeightSrc {
    int32_t chan;     // global channel index, or -1: the weight the host set
    uint16_t first;   // first target of the channel's mesh section (component = m - first)
    uint16_t width;   // targets of that section = floats per key of the channel
};

// a skin's joints inside the palette
struct SkinSpan {
    uint32_t first, count;
};

struct CompiledRig {
    // J = palette entries: every skin's joints one skin after the other, then one entry per distinct node of an
    // un-skinned section (joint = the node's world matrix, inverse bind = identity)
    uint32_t N = 0, J = 0, V = 0, K = 0, M = 0, clips = 0, Cmax = 0, levels = 0, T = 0;
    std::vector<SkinSpan> skins;
    std::vector<skinrt_section_layout> sections;
    uint32_t tangent_stride = 3;
    // skin-stage arithmetic chosen for this rig (skinrt_config.exact_arithmetic; 0 = by bind_extent)
    uint32_t exact = 0;
    float bind_extent = 0.0f;           // largest |translation component| of an inverse bind matrix
    // nodes (DFS pre-order)
    std::vector<int32_t> parent;        // [N]
    std::vector<float> rest_trs;        // [N][10] t3 q4 s3 (Matrix nodes: decomposed, node.rs:76-82)
    std::vector<float> rest_mat;        // [N][12] 3 rows x 4 (Matrix nodes only, else unused)
    std::vector<uint8_t> node_matrix;   // [N] 1 = NodeTransform::Matrix
    std::vector<uint32_t> level_start;  // [levels+1]
    std::vector<uint32_t> level_nodes;  // [N] nodes ordered by depth (stable)
    // clips
    std::vector<int32_t> node_chan;     // [clips][N][3] global channel index or -1
    std::vector<ChanRec> chans;
    std::vector<uint32_t> clip_chan_count;  // [clips]
    std::vector<WeightSrc> weight_src;      // [clips][M]
    uint32_t weight_channel_clips = 0;      // clips that carry at least one SKINRT_PATH_WEIGHTS channel
    std::vector<float> clip_length;         // [clips] largest key time of the clip (AnimationChannelAsset.length, max over channels)
    std::vector<float> times, values;
    // skin
    std::vector<uint32_t> joint_node;   // [J]
    std::vector<float> ibm;             // [J][12] 3 rows x 4
    // mesh, tiled: [T][attr][kTileVerts]
    std::vector<float> base;            // [T][9][256]  px py pz nx ny nz tx ty tz
    std::vector<float> wts;             // [T][4K][256]
    std::vector<uint16_t> lidx;         // [T][4K][256] tile-local joint index (LDS tiles) or global joint id
    std::vector<uint32_t> tile_joint_start;  // [T+1]
    std::vector<uint32_t> tile_joints;       // concatenated tile joint lists (global ids, ascending)
    std::vector<uint8_t> tile_lds;           // [T] 0 = joint list exceeds kTileJointCap, 1 = fits, 2 = at most kSkin2MaxJoints joints, 3 = as 2 but
                                             // longer than the lists k_skin2's palette ring is sized for: all its instances go to k_skin_fix
    uint32_t tile_kind_count[4] = {0, 0, 0, 0}; // tiles of each kind
    uint32_t max_tile_joints = 0;            // over LDS tiles
    uint32_t max_tile_joints_any = 0;
    uint32_t max_tile_joints2 = 0;           // over the self-staging kernel's tiles (kind 2): <= kSkin2MaxJoints
    // k_skin2 in two launches (0: one): kind-2 tiles with at most split_joints joints run with a palette ring sized for them
    // (split_joints_short = their longest list), the longer ones with the ring of max_tile_joints2 — a rig of mostly short
    // lists does not pay the long ones' LDS on every tile
    uint32_t split_joints = 0, split_joints_short = 0;
    uint32_t lds_tiles = 0;
    uint64_t extra_set_waves = 0;            // (wave, weight set beyond the first) units of the tiled mesh
    uint64_t skippable_set_waves = 0;        // those whose weights are zero in all 64 lanes, trailing
    bool skip_sets = false;                  // the skin kernels that stop a wave after its last used set are worth it
    // morph: (tile, target) pairs, targets ascending inside a tile
    std::vector<uint32_t> tile_pair_start;  // [T+1]
    std::vector<MorphPair> pairs;
    std::vector<float> mblock;          // pair regions (see MorphPair::data_off)
    std::vector<float> default_weights; // [M]
    uint32_t max_tile_pairs = 0;
    uint64_t morph_entries = 0;         // nonzero (vertex, target) deltas kept
    uint32_t promoted_pairs = 0;        // (tile, target) pairs small enough to compact but stored dense (see compile_impl)
    // kept for skinrt_export_interleaved
    std::vector<uint16_t> joints0;      // [V][4] (indices into the vertex's own skin, as given; zeros for un-skinned sections)
    std::vector<float> weights0;        // [V][4]
};

// Returns SKINRT_OK or a negative SKINRT_E_* with `err` describing the problem.
int compile_rig(const skinrt_rig_desc* desc, const skinrt_config* cfg, CompiledRig& out, std::string& err);
void fill_rig_info(const CompiledRig& r, uint64_t device_bytes, skinrt_rig_info* info);

// glam Mat4::to_scale_rotation_translation, restated (node.rs:76-82); m column-major [16].
void decompose_affine(const float* m, float* t3, float* q4, float* s3);

}  // namespace skinrt
