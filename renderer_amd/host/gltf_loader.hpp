// gltf_loader.hpp — native .glb -> (Vec<SceneAsset>, Vec<AnimationAsset>) loader for the parts of
// a glTF file that feed the skinned-scene path, following the reference
// loader's conventions and quirks (renderer-asset/src/loader/gltf/mod.rs) so that the two produce
// the same asset trees from the same file:
//   * accessor without a bufferView -> `count` zero BYTES, sparse accessors not applied (:241-245)
//   * strided reads; the bounds check counts a full stride for the last element too (:265-272)
//   * normalised reads u8/255, u16/65535, u32/u32::MAX, f32 as is; signed types rejected (:362-370)
//   * JOINTS_n u8 widened to u16 (:1019-1031); indices u8/u16/u32 -> u32 (:931-948)
//   * missing TANGENT -> calculate_tangent() of the positions, w = 1 (tangent.rs:15-58, :1086-1088);
//     missing NORMAL -> xyz of the TANGENT (:1089)
//   * morph target TANGENT must be VEC4 (:906); missing -> calculate_tangent() of the DELTA positions
//   * CUBICSPLINE: times repeated three times, zipped, regrouped as (in, value, out) (:1226-1266)
//   * skin without inverseBindMatrices -> empty Vec (:1139-1145)
//   * node with `matrix` -> NodeTransform::Matrix, else Decomposed with glTF defaults (:1174-1187)
//   * weights animation channels: the reference hits todo!() (:1312) -> an error here
// Materials, textures, cameras and images are not on the path and are not read.
// Where the reference panics (expect / assert / todo! / unreachable!), this returns an error code.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "asset.hpp"

namespace skinrt_host {

using namespace renderer_asset;

// load_glb_from_buffer_with_id (loader/gltf/mod.rs:1367-1381).  Returns SKINRT_OK or SKINRT_E_*.
int load_glb_from_buffer_with_id(const uint8_t* buffer, size_t size, const BundleIndex& id,
                                 std::vector<SceneAsset>& scenes, std::vector<AnimationAsset>& animations,
                                 std::string& err);

// calculate_tangent (renderer-asset/src/tangent.rs:15-58): per-vertex sum of unit face normals,
// normalised (vertices no triangle touches come out NaN, as in the reference).  mode is the
// PrimitiveAssetMode; returns false for TriangleStrip (todo!() in the reference).
bool calculate_tangent(PrimitiveAssetMode mode, const std::vector<std::array<float, 3>>& positions,
                       const std::vector<uint32_t>* indices, std::vector<std::array<float, 3>>& out);

}  // namespace skinrt_host
