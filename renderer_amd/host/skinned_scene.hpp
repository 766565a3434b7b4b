// skinned_scene.hpp — the caller's side of the C ABI in C++: owns the handles, turns error
// codes into one exception type at THIS layer only (nothing crosses the C boundary), and mirrors
// the two reference objects that consume the path's output:
//   SkinMatrix / SkinUniformBuffer   renderer/src/renderer/uniform/skin.rs:12-23,59-71
//   ColorSkinVertex / TextureSkinVertex  renderer/src/renderer/buffer/vertex.rs:60-102
//   InstanceUniform / InstanceUniformBuffer  renderer/src/renderer/uniform/transform.rs:8-20,51-59
// SkinnedScene::render() is what Entity::render (renderer/src/client/entity/mod.rs:34) would
// call once per frame for an asset-backed entity.
#pragma once
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/skinrt.h"
#include "flatten.hpp"

namespace skinrt_host {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error("skinrt error " + std::to_string(c) + ": " + what), code(c) {}
};

inline void check(int rc) {
    if (rc != SKINRT_OK) throw Error(rc, skinrt_last_error());
}

constexpr size_t MAX_JOINTS = SKINRT_MAX_JOINTS_REFERENCE;  // uniform/skin.rs:10

struct SkinMatrix {        // uniform/skin.rs:12-17, 112 bytes
    float transform[4][4]; // column-major
    float normal[3][4];    // inverse-transpose columns, w = 0
};
static_assert(sizeof(SkinMatrix) == 112, "SkinMatrix layout");

#pragma pack(push, 1)
struct InstanceUniform {   // uniform/transform.rs:8-14, 124 bytes
    float transform[4][4]; // column-major
    float normal[3][4];    // inverse-transpose columns, w = 0
    uint8_t padding[12];
};
#pragma pack(pop)
static_assert(sizeof(InstanceUniform) == SKINRT_INSTANCE_UNIFORM_BYTES, "InstanceUniform layout");

struct ColorSkinVertex {   // buffer/vertex.rs:60-69, 76 bytes
    float position[3], color[4], normal[3], tangent[3];
    uint16_t joint_index[4];
    float joint_weight[4];
};
struct TextureSkinVertex { // buffer/vertex.rs:82-91, 68 bytes
    float position[3], tex_coords[2], normal[3], tangent[3];
    uint16_t joint_index[4];
    float joint_weight[4];
};
static_assert(sizeof(ColorSkinVertex) == 76 && sizeof(TextureSkinVertex) == 68, "vertex layouts");

class SkinnedScene {
public:
    // One context per device, one rig per (scenes, animations), one crowd of `instances`.
    SkinnedScene(const std::vector<SceneAsset>& scenes, const std::vector<AnimationAsset>& animations,
                 uint32_t instances, const skinrt_config* config = nullptr, const FlattenOptions& opt = {}) {
        std::string err;
        const int rc = flatten(scenes, animations, opt, flat_, err);
        if (rc != SKINRT_OK) throw Error(rc, err);
        skinrt_config defaults{};
        defaults.struct_size = sizeof defaults;
        defaults.cubic_quirk = 1;  // reproduce animation.rs:57 as written
        check(skinrt_init(config ? config : &defaults, &ctx_));
        try {
            check(skinrt_upload_rig(ctx_, flat_.desc(), &rig_));
            check(skinrt_create_crowd(ctx_, rig_, instances, &crowd_));
        } catch (...) {
            release();
            throw;
        }
        instances_ = instances;
    }
    SkinnedScene(const SkinnedScene&) = delete;
    SkinnedScene& operator=(const SkinnedScene&) = delete;
    ~SkinnedScene() { release(); }

    const FlatRig& rig() const { return flat_; }
    uint32_t instances() const { return instances_; }

    // tau[i] = fmod(clock_i, clip_length(clip_i)) is the caller's job
    void set_clocks(const float* tau, const uint32_t* clip_id) { check(skinrt_set_clocks(crowd_, tau, clip_id)); }
    void set_morph_weights(const float* w) { check(skinrt_set_morph_weights(crowd_, w)); }
    void set_root_transforms(const float* m) { check(skinrt_set_root_transforms(crowd_, m)); }

    void render(uint32_t stages = SKINRT_STAGE_ALL) { check(skinrt_evaluate(crowd_, stages)); }
    void synchronize() { check(skinrt_synchronize(ctx_)); }
    // Normals (or tangents) of what the frame produced: calculate_tangent (tangent.rs:15-58) per instance on the skinned
    // positions, over the scene's triangle lists (FlatRig::triangles); enqueued behind render().  The topology is built on
    // first use.  The reference runs the generator once, at load, on the bind pose (loader/gltf/mod.rs:1086-1089).
    void recompute_normals(uint32_t first, uint32_t count, uint32_t output = SKINRT_OUTPUT_NORMALS) {
        if (!topology_) {
            static const uint32_t none = 0;  // a scene without triangle lists has NO faces (a NULL list would mean "every three vertices")
            check(skinrt_create_topology(ctx_, flat_.vertex_count(), flat_.triangles.empty() ? &none : flat_.triangles.data(),
                                         (uint32_t)flat_.triangles.size(), &topology_));
        }
        check(skinrt_recompute_normals(crowd_, topology_, first, count, output));
    }

    // SkinUniformBuffer::update for one instance and one of the scene's skins: 128 SkinMatrix entries, unused ones zero
    std::vector<SkinMatrix> skin_uniform(uint32_t instance, uint32_t skin = 0) {
        std::vector<SkinMatrix> out(MAX_JOINTS);
        check(skinrt_export_skin_uniform_of(crowd_, skin, instance, out.data()));
        return out;
    }
    // InstanceUniformBuffer::update for a node that carries an un-skinned mesh (or any other node)
    InstanceUniform instance_uniform(uint32_t instance, uint32_t node) {
        InstanceUniform out;
        check(skinrt_export_instance_uniform(crowd_, instance, node, &out));
        return out;
    }
    // one mesh node's vertices (FlatRig::sections) in the layout VertexBuffer::new takes for a skinned draw
    std::vector<TextureSkinVertex> texture_skin_vertices_of(uint32_t section, uint32_t instance, const float* tex_coords = nullptr) {
        std::vector<TextureSkinVertex> out(flat_.sections.at(section).vertex_count);
        check(skinrt_export_interleaved_section(crowd_, section, instance, SKINRT_LAYOUT_TEXTURE_SKIN, tex_coords, out.data()));
        return out;
    }
    std::vector<TextureSkinVertex> texture_skin_vertices(uint32_t instance, const float* tex_coords = nullptr) {
        std::vector<TextureSkinVertex> out(flat_.vertex_count());
        check(skinrt_export_interleaved(crowd_, instance, SKINRT_LAYOUT_TEXTURE_SKIN, tex_coords, out.data()));
        return out;
    }
    std::vector<ColorSkinVertex> color_skin_vertices(uint32_t instance, const float* color = nullptr) {
        std::vector<ColorSkinVertex> out(flat_.vertex_count());
        check(skinrt_export_interleaved(crowd_, instance, SKINRT_LAYOUT_COLOR_SKIN, color, out.data()));
        return out;
    }

    template <class T>
    std::vector<T> download(int (*getter)(skinrt_crowd*, T**, uint64_t*)) {
        T* dev = nullptr;
        uint64_t n = 0;
        check(getter(crowd_, &dev, &n));
        std::vector<T> host(n);
        check(skinrt_download(ctx_, dev, host.data(), n * sizeof(T)));
        return host;
    }

    skinrt_crowd* crowd() { return crowd_; }

private:
    void release() {
        if (topology_) skinrt_destroy_topology(topology_);
        if (crowd_) skinrt_destroy_crowd(crowd_);
        if (rig_) skinrt_destroy_rig(rig_);
        if (ctx_) skinrt_destroy_ctx(ctx_);
        topology_ = nullptr; crowd_ = nullptr; rig_ = nullptr; ctx_ = nullptr;
    }
    FlatRig flat_;
    skinrt_ctx* ctx_ = nullptr;
    skinrt_rig* rig_ = nullptr;
    skinrt_crowd* crowd_ = nullptr;
    skinrt_topology* topology_ = nullptr;
    uint32_t instances_ = 0;
};

}  // namespace skinrt_host
