// host_capi.cpp — see host_capi.h.
#define SKINRT_BUILD
#include "host_capi.h"

#include <new>
#include <string>

#include "flatten.hpp"
#include "gltf_loader.hpp"
#include "pmx_loader.hpp"

using namespace skinrt_host;

struct skinrt_host_rig {
    FlatRig flat;
    PmxInfo pmx;  // filled by skinrt_host_load_pmx
};

namespace {
thread_local std::string g_error;
int fail(int code, const std::string& what) {
    g_error = what;
    return code;
}
}  // namespace

extern "C" {

int skinrt_host_load_glb(const uint8_t* data, size_t size, int scene, skinrt_host_rig** out) {
    if (!out) return fail(SKINRT_E_INVALID, "out is NULL");
    *out = nullptr;
    try {
        std::vector<SceneAsset> scenes;
        std::vector<AnimationAsset> animations;
        std::string err;
        BundleIndex id{};  // the digest only has to be consistent inside one load
        int rc = load_glb_from_buffer_with_id(data, size, id, scenes, animations, err);
        if (rc != SKINRT_OK) return fail(rc, err);
        auto* rig = new skinrt_host_rig();
        FlattenOptions opt;
        opt.scene = scene;
        rc = flatten(scenes, animations, opt, rig->flat, err);
        if (rc != SKINRT_OK) {
            delete rig;
            return fail(rc, err);
        }
        *out = rig;
        return SKINRT_OK;
    } catch (const std::bad_alloc&) {
        return fail(SKINRT_E_NOMEM, "out of memory");
    }
}

int skinrt_host_load_pmx(const uint8_t* data, size_t size, skinrt_host_rig** out) {
    if (!out) return fail(SKINRT_E_INVALID, "out is NULL");
    *out = nullptr;
    try {
        auto* rig = new skinrt_host_rig();
        std::string err;
        BundleIndex id{};
        const int rc = load_pmx_from_buffer(data, size, id, rig->flat, rig->pmx, err);
        if (rc != SKINRT_OK) {
            delete rig;
            return fail(rc, err);
        }
        *out = rig;
        return SKINRT_OK;
    } catch (const std::bad_alloc&) {
        return fail(SKINRT_E_NOMEM, "out of memory");
    }
}

const uint32_t* skinrt_host_rig_pmx_indices(const skinrt_host_rig* rig, uint32_t* count) {
    if (count) *count = rig ? (uint32_t)rig->pmx.indices.size() : 0;
    return (rig && !rig->pmx.indices.empty()) ? rig->pmx.indices.data() : nullptr;
}

void skinrt_host_rig_pmx_counts(const skinrt_host_rig* rig, uint32_t out[10]) {
    for (int k = 0; k < 10; ++k) out[k] = 0;
    if (!rig) return;
    const PmxInfo& p = rig->pmx;
    out[0] = p.vertices; out[1] = p.triangles; out[2] = p.materials; out[3] = p.bones; out[4] = p.morphs; out[5] = p.vertex_morphs;
    // out[6..9]: BDEF1, BDEF2 + SDEF, BDEF4 + QDEF, SDEF + QDEF (the ones blended linearly as an approximation)
    out[6] = p.deform_count[0]; out[7] = p.deform_count[1] + p.deform_count[3]; out[8] = p.deform_count[2] + p.deform_count[4];
    out[9] = p.deform_count[3] + p.deform_count[4];
}

const skinrt_rig_desc* skinrt_host_rig_desc(skinrt_host_rig* rig) { return rig ? rig->flat.desc() : nullptr; }

uint32_t skinrt_host_rig_node_gltf_index(const skinrt_host_rig* rig, uint32_t n) {
    return (rig && n < rig->flat.node_id.size()) ? (uint32_t)rig->flat.node_id[n].index : 0xFFFFFFFFu;
}

float skinrt_host_rig_clip_length(const skinrt_host_rig* rig, uint32_t clip) {
    return (rig && clip < rig->flat.clip_length.size()) ? rig->flat.clip_length[clip] : 0.0f;
}

uint64_t skinrt_host_rig_sanitized_tangent_deltas(const skinrt_host_rig* rig) { return rig ? rig->flat.sanitized_tangent_deltas : 0; }

void skinrt_host_free_rig(skinrt_host_rig* rig) { delete rig; }

const char* skinrt_host_last_error(void) { return g_error.c_str(); }

int skinrt_host_calculate_tangent(uint32_t mode, const float* positions, uint32_t count, const uint32_t* indices,
                                  uint32_t index_count, float* out) {
    if (mode > 4 || (!positions && count) || !out) return fail(SKINRT_E_INVALID, "bad argument");
    static const PrimitiveAssetMode modes[5] = {PrimitiveAssetMode::Points, PrimitiveAssetMode::LineStrip, PrimitiveAssetMode::LineList,
                                                PrimitiveAssetMode::TriangleStrip, PrimitiveAssetMode::TriangleList};
    std::vector<std::array<float, 3>> pos(count), res;
    for (uint32_t k = 0; k < count; ++k) pos[k] = {positions[3 * k], positions[3 * k + 1], positions[3 * k + 2]};
    std::vector<uint32_t> idx;
    if (indices) idx.assign(indices, indices + index_count);
    if (!calculate_tangent(modes[mode], pos, indices ? &idx : nullptr, res)) return fail(SKINRT_E_INVALID, "TriangleStrip: todo!() in the reference, or index out of range");
    for (uint32_t k = 0; k < count; ++k) { out[3 * k] = res[k][0]; out[3 * k + 1] = res[k][1]; out[3 * k + 2] = res[k][2]; }
    return SKINRT_OK;
}

}  // extern "C"
