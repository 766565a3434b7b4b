// flatten.hpp — walks the loader output (Vec<SceneAsset>, Vec<AnimationAsset>) once at load time
// and produces the flat rig the C ABI takes (skinrt_rig_desc).  This is where the path's "node
// ordering" is fixed: DFS pre-order over scene.nodes, then children, in file order
// (trees built by load_node / load_scene, loader/gltf/mod.rs:1168-1217).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/skinrt.h"
#include "asset.hpp"

namespace skinrt_host {

using namespace renderer_asset;

struct FlattenOptions {
    int scene = -1;                 // -1: every scene in order; else only scenes[scene]
    // null: EVERY node with a mesh becomes a section — all skins of the scenes (an avatar usually has several over one
    // node tree: body / face / hair) and the un-skinned meshes; else only the meshes that use this skin
    const AssetIndex* skin = nullptr;
    bool include_unskinned = true;  // with skin == null: meshes on nodes without a skin become un-skinned sections
};

// a primitive's vertices inside the rig's concatenated vertex arrays
struct PrimitiveRange {
    uint32_t node;        // flat node index of the mesh node
    uint32_t primitive;   // index in MeshAsset::primitives
    uint32_t first_vertex, vertex_count;
    uint32_t section;     // index in FlatRig::sections
};

// One SkinAsset (skin.rs:5-11) over the flat nodes
struct FlatSkin {
    AssetIndex id;
    std::vector<uint32_t> joint_node;
    std::vector<float> inverse_bind;  // empty = identity
};

struct FlatChannel {
    uint32_t node, path, interpolation;
    std::vector<float> times;   // [Kf]
    std::vector<float> values;  // [Kf][w] or [Kf][3][w] (in, value, out)
};

struct FlatTarget {
    bool dense = true;
    std::vector<uint32_t> indices;            // !dense: strictly increasing vertex ids
    std::vector<float> position, normal, tangent;  // [count][3], [count][3] or empty, [count][4] or empty
};

struct FlatRig {
    // nodes, DFS pre-order
    std::vector<AssetIndex> node_id;
    std::vector<int32_t> parent;
    std::vector<float> translation, rotation, scale, matrix;
    std::vector<uint8_t> has_matrix;
    // skins in the order the DFS meets them, and the mesh sections (skinrt_section_desc: node, skin or -1, vertex and
    // target ranges).  Loaders that produce one skinned mesh (PMX) may instead fill the short form below and leave
    // both empty.
    std::vector<FlatSkin> skins;
    std::vector<skinrt_section_desc> sections;
    // short form: one skin, one section
    std::vector<uint32_t> joint_node;
    std::vector<float> inverse_bind;  // empty = identity
    // clips (one per AnimationAsset, channels in file order)
    std::vector<std::vector<FlatChannel>> clips;
    std::vector<float> clip_length;   // max over channels of AnimationChannelAsset::length
    // mesh: primitives of every mesh node, concatenated in DFS order (a section = one node's primitives)
    std::vector<float> position, normal, tangent;  // tangent stride 4
    std::vector<std::vector<uint16_t>> joints;     // K x [V][4]
    std::vector<std::vector<float>> weights;       // K x [V][4]
    std::vector<FlatTarget> targets;
    std::vector<float> default_weights;
    uint64_t sanitized_tangent_deltas = 0;         // non-finite morph tangent deltas replaced by zero (flatten.cpp)
    std::vector<PrimitiveRange> ranges;
    // triangle corners of every TriangleList primitive as indices into the RIG's vertex array (PrimitiveAsset.indices,
    // primitive.rs:47, offset by the primitive's first vertex; a primitive without indices contributes 3f, 3f+1, 3f+2 as
    // calculate_tangent reads it, tangent.rs:42-51): what skinrt_create_topology takes for per-frame normals.  Whole
    // triangles only (len() / 3 per primitive, tangent.rs:29); other modes have no faces.
    std::vector<uint32_t> triangles;

    uint32_t node_count() const { return (uint32_t)parent.size(); }
    uint32_t vertex_count() const { return (uint32_t)(position.size() / 3); }

    // Descriptor whose pointers reference this object: valid until it is modified or destroyed.
    const skinrt_rig_desc* desc();

private:
    skinrt_rig_desc desc_{};
    std::vector<skinrt_clip_desc> clip_descs_;
    std::vector<std::vector<skinrt_channel_desc>> channel_descs_;
    std::vector<skinrt_morph_target_desc> target_descs_;
    std::vector<const uint16_t*> joint_ptrs_;
    std::vector<const float*> weight_ptrs_;
    std::vector<skinrt_skin_desc> skin_descs_;
};

// Returns SKINRT_OK or a negative SKINRT_E_* (never throws / aborts; the reference panics on the
// same inputs: unknown channel target, joint id that is no node, attribute length mismatch).
int flatten(const std::vector<SceneAsset>& scenes, const std::vector<AnimationAsset>& animations,
            const FlattenOptions& opt, FlatRig& out, std::string& err);

}  // namespace skinrt_host
