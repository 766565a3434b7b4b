// pmx_loader.hpp — PMX 2.0 (MikuMikuDance) models onto the flat rig.
//
// The reference parses the whole file (renderer-asset/src/loader/pmx/format.rs:159-780) and then
// keeps only positions / normals / UVs per material as static, un-skinned primitives: bones, skin
// weights and morphs are read and dropped (loader/pmx/mod.rs:200-262).  This loader reads the same
// binary layout, in the same field order and with the same checks (magic "PMX ", version 2.0,
// globals_count >= 8, 0..=4 additional vec4, index widths 1/2/4 with -1 = none and other negatives
// rejected, surface count a multiple of 3), and maps what the reference drops onto the rig the
// path evaluates.  Nothing in the reference defines that mapping, so it is BUILD-DEFINED:
//   * node k = bone k's position in a DFS pre-order over the bone forest (roots and children in
//     bone-index order, which keeps parent[n] < n); a PMX bone carries an absolute rest position
//     only, so local translation = position - parent position, rotation identity, scale 1;
//   * joint j = bone j (vertex bone indices are used unchanged); inverse bind = translation by
//     -position (the bind pose has no rotation);
//   * weights, one set: BDEF1 (b, 1); BDEF2 (b1, w), (b2, 1 - w); BDEF4 and QDEF (b1..b4, w1..w4 as
//     stored, not renormalised: linear blending of a QDEF vertex is an approximation); SDEF as
//     BDEF2 (its C / R0 / R1 correction is not applied); a "none" index becomes joint 0, weight 0;
//   * vertices stay indexed (the reference expands them per material); normal as stored; tangent =
//     calculate_tangent over the whole surface list, w = 1 — the reference calls the same function per
//     material on the expanded vertices (mod.rs:228-231), which differs at material seams;
//   * every Vertex morph becomes one sparse morph target (offsets sorted by vertex, repeated
//     vertices added up), default weight 0; Group / Bone / UV / Material / Flip / Impulse morphs are
//     parsed and dropped;
//   * no animation: the reference has no VMD reader, so the rig has no clips.
// Display frames, rigid bodies and joints (after the morphs) are not read.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "flatten.hpp"

namespace skinrt_host {

struct PmxInfo {
    uint32_t vertices = 0, triangles = 0, materials = 0, bones = 0, morphs = 0, vertex_morphs = 0;
    uint32_t deform_count[5] = {0, 0, 0, 0, 0};  // BDEF1, BDEF2, BDEF4, SDEF, QDEF
    std::vector<uint32_t> indices;                // the surface list (3 per triangle)
    std::vector<uint32_t> material_surface_count; // per material, in file order
};

// Returns SKINRT_OK or a negative SKINRT_E_* with `err` set; never throws for malformed input.
int load_pmx_from_buffer(const uint8_t* buffer, size_t size, const BundleIndex& id, FlatRig& rig, PmxInfo& info, std::string& err);

}  // namespace skinrt_host
