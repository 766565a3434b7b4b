// asset.hpp — C++ mirror of the reference's asset types on the skinned-scene path.
//
// The reference's caller is Rust (crates/renderer-asset); no Rust toolchain exists in this
// image, so the host side above the C ABI is written in C++ with the reference's names, field
// order and ownership (owned trees by value, sharing only through shared_ptr where the
// reference uses Arc).  INTEGRATION.md shows the Rust binding that replaces this mirror.
//
//   AssetIndex, BundleAssetType      renderer-asset/src/index.rs:66-90
//   DecomposedTransform, NodeTransform   renderer-asset/src/node.rs:9-39
//   NodeAsset                        renderer-asset/src/node.rs:89-99
//   SceneAsset                       renderer-asset/src/scene.rs:3-7
//   SkinAsset                        renderer-asset/src/skin.rs:5-11
//   MeshAsset                        renderer-asset/src/mesh.rs:3-8
//   PrimitiveAsset*                  renderer-asset/src/primitive.rs:14-47
//   Animation*                       renderer-asset/src/animation.rs:8-40
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <optional>
#include <string>
#include <tuple>
#include <variant>
#include <vector>

namespace renderer_asset {

using Mat4 = std::array<float, 16>;  // column-major, glam Mat4::to_cols_array
using Vec3 = std::array<float, 3>;
using Quat = std::array<float, 4>;   // x y z w

inline Mat4 mat4_identity() { return {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; }

using BundleIndex = std::array<uint8_t, 32>;  // sha256 of the bundle (index.rs:44-63)

enum class BundleAssetType { Node, Skin, Texture, Material };

struct AssetIndex {
    enum class Kind { Bundle, BundlePath, BundleTypeIndex } kind = Kind::Bundle;
    BundleIndex bundle{};
    std::string path;                              // BundlePath
    BundleAssetType type = BundleAssetType::Node;  // BundleTypeIndex
    size_t index = 0;                              // BundleTypeIndex

    static AssetIndex type_index(const BundleIndex& b, BundleAssetType t, size_t i) {
        AssetIndex a;
        a.kind = Kind::BundleTypeIndex; a.bundle = b; a.type = t; a.index = i;
        return a;
    }
    auto key() const { return std::tie(kind, bundle, path, type, index); }
    bool operator==(const AssetIndex& o) const { return key() == o.key(); }
    bool operator<(const AssetIndex& o) const { return key() < o.key(); }
};

struct MatrixNodeTransform { Mat4 m = mat4_identity(); };

struct DecomposedTransform {
    Vec3 translation{0, 0, 0};
    Quat rotation{0, 0, 0, 1};
    Vec3 scale{1, 1, 1};
};

using NodeTransform = std::variant<DecomposedTransform, MatrixNodeTransform>;  // default = identity TRS

struct SkinAsset {
    AssetIndex id;
    std::vector<Mat4> inverse_bind_matrices;  // empty when the file has no accessor (loader/gltf/mod.rs:1139-1145)
    std::vector<AssetIndex> joint_ids;
    std::optional<AssetIndex> skeleton;
};

enum class PrimitiveAssetMode { Points, LineStrip, LineList, TriangleStrip, TriangleList };

struct PrimitiveAssetAttributes {
    std::vector<std::array<float, 3>> position;
    std::vector<std::array<float, 3>> normal;
    std::vector<std::array<float, 4>> tangent;
    std::vector<std::vector<std::array<float, 2>>> tex_coord;
    std::vector<std::vector<std::array<float, 4>>> color;
    std::vector<std::vector<std::array<uint16_t, 4>>> joints;  // K sets
    std::vector<std::vector<std::array<float, 4>>> weights;    // K sets
};

struct PrimitiveAssetMorphTarget {
    std::vector<std::array<float, 3>> position;
    std::vector<std::array<float, 3>> normal;   // may be empty
    std::vector<std::array<float, 4>> tangent;  // may be empty
};

struct PrimitiveAsset {
    PrimitiveAssetAttributes attributes;
    std::optional<std::vector<uint32_t>> indices;
    // material: Option<Arc<MaterialAsset>> — not on this path
    PrimitiveAssetMode mode = PrimitiveAssetMode::TriangleList;
    std::vector<PrimitiveAssetMorphTarget> targets;
};

struct MeshAsset {
    std::optional<std::string> name;
    std::vector<PrimitiveAsset> primitives;
    std::vector<float> weights;
};

struct NodeAsset {
    AssetIndex id;
    std::optional<std::string> name;
    // camera: Option<CameraAsset> — not on this path
    std::vector<NodeAsset> children;
    std::shared_ptr<const SkinAsset> skin;  // Option<Arc<SkinAsset>>
    std::optional<NodeTransform> transform; // None (PMX nodes, loader/pmx/mod.rs:253) = identity
    std::optional<MeshAsset> mesh;
    std::vector<float> weights;
};

struct SceneAsset {
    std::optional<std::string> name;
    std::vector<NodeAsset> nodes;
};

template <class T>
struct AnimationKeyFrame {
    float time;
    T value;
};

template <class T>
struct CubicValue { T in_tangent, value, out_tangent; };  // "in, val, out" (animation.rs:18-19)

template <class T>
struct AnimationKeyFrames {
    enum class Kind { Linear, Step, CubicSpline } kind = Kind::Linear;
    std::vector<AnimationKeyFrame<T>> frames;                    // Linear / Step
    std::vector<AnimationKeyFrame<CubicValue<T>>> cubic_frames;  // CubicSpline
};

struct AnimationSampler {
    // MorphTargetWeights: the variant the reference's loader has not written yet (Property::MorphTargetWeights =>
    // todo!(), loader/gltf/mod.rs:1312); here: M floats per key, M = the morphed mesh's target count
    enum class Kind { Rotation, Translation, Scale, MorphTargetWeights } kind = Kind::Translation;
    AnimationKeyFrames<std::array<float, 4>> rotation;
    AnimationKeyFrames<std::array<float, 3>> vec3;  // Translation / Scale
    AnimationKeyFrames<std::vector<float>> weights; // MorphTargetWeights
};

struct AnimationChannelAsset {
    AnimationSampler sampler;
    float length = 0;  // max key time of this channel (loader/gltf/mod.rs:1243-1247)
    AssetIndex target_id;
};

struct AnimationAsset {
    std::optional<std::string> name;
    std::vector<AnimationChannelAsset> channels;
};

}  // namespace renderer_asset
