// pmx_loader.cpp — see pmx_loader.hpp.  Compiled with -ffp-contract=off (bone-relative translations
// and calculate_tangent are plain fp32 operations in a fixed order).
#include "pmx_loader.hpp"

#include <algorithm>
#include <cstring>

#include "../../include/skinrt.h"
#include "gltf_loader.hpp"

namespace skinrt_host {
namespace {

struct Fail {
    int code;
    std::string what;
};
[[noreturn]] void fail(int code, const std::string& what) { throw Fail{code, what}; }

constexpr uint32_t kNone = 0xFFFFFFFFu;

// little-endian reader over the file; every read is bounds-checked
struct Reader {
    const uint8_t* p;
    size_t size, at = 0;
    void need(size_t n) const {
        if (n > size - at) fail(SKINRT_E_RANGE, "PMX file ends inside a record (offset " + std::to_string(at) + ")");
    }
    uint8_t u8() { need(1); return p[at++]; }
    int8_t i8() { return (int8_t)u8(); }
    uint16_t u16() { need(2); uint16_t v; std::memcpy(&v, p + at, 2); at += 2; return v; }
    int32_t i32() { need(4); int32_t v; std::memcpy(&v, p + at, 4); at += 4; return v; }
    float f32() { need(4); float v; std::memcpy(&v, p + at, 4); at += 4; return v; }
    void f32s(float* dst, size_t n) { need(n * 4); std::memcpy(dst, p + at, n * 4); at += n * 4; }
    void skip(size_t n) { need(n); at += n; }
    // PmxText (format.rs:41-46): u32 length + bytes; UTF-16 text must have an even length (:51-53)
    void text(bool utf16) {
        need(4);
        uint32_t len; std::memcpy(&len, p + at, 4); at += 4;
        if (utf16 && (len & 1u)) fail(SKINRT_E_INVALID, "BadUtf16Text: odd byte length");
        skip(len);
    }
    // PmxIndex (format.rs:91-156): signed, width 1 / 2 / 4, -1 = none, other negatives rejected
    uint32_t index(uint32_t width) {
        int64_t v;
        if (width == 1) v = i8();
        else if (width == 2) v = (int16_t)u16();
        else v = i32();
        if (v == -1) return kNone;
        if (v < 0) fail(SKINRT_E_INVALID, "Bad index: " + std::to_string(v));
        return (uint32_t)v;
    }
    // a count field asserted non-negative by the reference (format.rs:745-779) and bounded by what the file can hold
    uint32_t count(const char* what, size_t min_record_bytes) {
        const int32_t n = i32();
        if (n < 0) fail(SKINRT_E_INVALID, std::string(what) + " count is negative");
        if (min_record_bytes && (size_t)n > (size - at) / min_record_bytes) fail(SKINRT_E_RANGE, std::string(what) + " count exceeds the file");
        return (uint32_t)n;
    }
};

uint32_t index_width(uint8_t magic) {  // PmxIndexType (format.rs:79-86)
    if (magic == 1 || magic == 2 || magic == 4) return magic;
    fail(SKINRT_E_INVALID, "index type must be 1, 2 or 4 bytes");
}

struct Bone {
    float pos[3];
    uint32_t parent;
};

}  // namespace

int load_pmx_from_buffer(const uint8_t* buffer, size_t size, const BundleIndex& id, FlatRig& rig, PmxInfo& info, std::string& err) {
    rig = FlatRig();
    info = PmxInfo();
    try {
        if (!buffer || size < 9 || std::memcmp(buffer, "PMX ", 4) != 0) fail(SKINRT_E_INVALID, "not a PMX file (magic)");
        Reader r{buffer, size, 4};
        if (r.f32() != 2.0f) fail(SKINRT_E_INVALID, "unsupported PMX version (the reference reads 2.0 only, format.rs:743)");
        // ---- PmxGlobals (format.rs:159-172)
        const int8_t globals = r.i8();
        if (globals < 8) fail(SKINRT_E_INVALID, "globals_count must be at least 8");
        const uint8_t enc = r.u8();
        if (enc > 1) fail(SKINRT_E_INVALID, "text encoding must be 0 (UTF-16LE) or 1 (UTF-8)");
        const bool utf16 = enc == 0;
        const int8_t extra = r.i8();
        if (extra < 0 || extra > 4) fail(SKINRT_E_INVALID, "additional_vec4_count must be 0..=4");
        const uint32_t wv = index_width(r.u8()), wt = index_width(r.u8()), wm = index_width(r.u8());
        const uint32_t wb = index_width(r.u8()), wmo = index_width(r.u8()), wr = index_width(r.u8());
        r.skip((size_t)(globals - 8));
        for (int k = 0; k < 4; ++k) r.text(utf16);  // model name / comments, local and universal

        // ---- vertices (format.rs:246-258)
        const uint32_t V = r.count("vertex", 38 + 16 * (size_t)extra);
        info.vertices = V;
        rig.position.resize((size_t)V * 3);
        rig.normal.resize((size_t)V * 3);
        rig.joints.assign(1, std::vector<uint16_t>((size_t)V * 4, 0));
        rig.weights.assign(1, std::vector<float>((size_t)V * 4, 0.0f));
        std::vector<uint32_t> vertex_bones((size_t)V * 4, kNone);
        for (uint32_t v = 0; v < V; ++v) {
            r.f32s(&rig.position[(size_t)v * 3], 3);
            r.f32s(&rig.normal[(size_t)v * 3], 3);
            r.skip(8 + 16 * (size_t)extra);  // uv, additional vec4
            const uint8_t deform = r.u8();
            uint32_t* b = &vertex_bones[(size_t)v * 4];
            float* w = &rig.weights[0][(size_t)v * 4];
            switch (deform) {
                case 0: b[0] = r.index(wb); w[0] = 1.0f; break;                                  // BDEF1
                case 1: case 3:                                                                   // BDEF2, SDEF
                    b[0] = r.index(wb); b[1] = r.index(wb); w[0] = r.f32(); w[1] = 1.0f - w[0];
                    if (deform == 3) r.skip(36);  // C, R0, R1: not applied
                    break;
                case 2: case 4:                                                                   // BDEF4, QDEF
                    for (int k = 0; k < 4; ++k) b[k] = r.index(wb);
                    r.f32s(w, 4);
                    break;
                default: fail(SKINRT_E_INVALID, "unknown weight deform type " + std::to_string(deform));
            }
            info.deform_count[deform] += 1;
            r.skip(4);  // edge scale
        }
        // ---- surfaces (format.rs:749-752)
        const uint32_t S = r.count("surface", wv);
        if (S % 3 != 0) fail(SKINRT_E_INVALID, "surface count is not a multiple of 3");
        info.indices.resize(S);
        for (uint32_t k = 0; k < S; ++k) {
            const uint32_t i = r.index(wv);
            if (i == kNone || i >= V) fail(SKINRT_E_RANGE, "surface names a vertex that does not exist (reference: unwrap / index panic, mod.rs:217)");
            info.indices[k] = i;
        }
        info.triangles = S / 3;
        // ---- textures, materials (format.rs:261-331): read to keep in step, and for the surface split
        const uint32_t T = r.count("texture", 4);
        for (uint32_t k = 0; k < T; ++k) r.text(utf16);
        const uint32_t M = r.count("material", 4 + 4 + 16 + 12 + 4 + 12 + 1 + 16 + 4 + 2 * wt + 1 + 1 + 1 + 4 + 4);
        info.materials = M;
        uint64_t surfaces_used = 0;
        for (uint32_t k = 0; k < M; ++k) {
            r.text(utf16); r.text(utf16);
            r.skip(16 + 12 + 4 + 12 + 1 + 16 + 4);  // diffuse, specular, strength, ambient, flags, edge colour, edge scale
            (void)r.index(wt); (void)r.index(wt);
            if (r.u8() > 3) fail(SKINRT_E_INVALID, "unknown environment blend mode");
            const uint8_t toon = r.u8();
            if (toon == 0) (void)r.index(wt); else if (toon == 1) r.skip(1); else fail(SKINRT_E_INVALID, "unknown toon reference");
            r.text(utf16);
            const int32_t sc = r.i32();
            if (sc < 0) fail(SKINRT_E_INVALID, "material surface count is negative");
            if (sc % 3 != 0) fail(SKINRT_E_INVALID, "BadSurfacesCount (mod.rs:207-208)");
            surfaces_used += (uint64_t)sc;
            if (surfaces_used > S) fail(SKINRT_E_RANGE, "NoSurfaceLeft (mod.rs:202-206)");
            info.material_surface_count.push_back((uint32_t)sc);
        }
        // ---- bones (format.rs:405-430)
        const uint32_t B = r.count("bone", 4 + 4 + 12 + wb + 4 + 2);
        info.bones = B;
        std::vector<Bone> bones(B);
        for (uint32_t k = 0; k < B; ++k) {
            r.text(utf16); r.text(utf16);
            r.f32s(bones[k].pos, 3);
            bones[k].parent = r.index(wb);
            r.skip(4);  // layer
            const uint16_t flags = r.u16();
            if (flags & 0x0001) (void)r.index(wb); else r.skip(12);           // tail: bone index or offset
            if (flags & (0x0100 | 0x0200)) { (void)r.index(wb); r.skip(4); }  // inherit rotation / translation
            if (flags & 0x0400) r.skip(12);                                    // fixed axis
            if (flags & 0x0800) r.skip(24);                                    // local coordinate
            if (flags & 0x2000) (void)r.index(wb);                             // external parent
            if (flags & 0x0020) {                                              // IK
                (void)r.index(wb);
                if (r.i32() < 0) fail(SKINRT_E_INVALID, "IK loop count is negative");
                r.skip(4);
                const uint32_t links = r.count("IK link", wb + 1);
                for (uint32_t l = 0; l < links; ++l) {
                    (void)r.index(wb);
                    const uint8_t limited = r.u8();
                    if (limited == 1) r.skip(24); else if (limited != 0) fail(SKINRT_E_INVALID, "bad IK link limit flag");
                }
            }
        }
        // ---- morphs (format.rs:601-613): vertex morphs are kept, the rest is stepped over
        const uint32_t MO = r.count("morph", 4 + 4 + 1 + 1 + 4);
        info.morphs = MO;
        for (uint32_t k = 0; k < MO; ++k) {
            r.text(utf16); r.text(utf16);
            if (r.u8() > 4) fail(SKINRT_E_INVALID, "unknown morph panel");
            const uint8_t type = r.u8();
            if (type > 10) fail(SKINRT_E_INVALID, "unknown morph type");
            const int32_t n = r.i32();
            if (n <= 0) fail(SKINRT_E_INVALID, "morph offset_size must be positive (format.rs:610)");
            static const uint32_t fixed[11] = {4, 12, 24, 16, 16, 16, 16, 16, 1 + 16 + 12 + 4 + 12 + 16 + 4 + 16 + 16 + 16, 4, 1 + 12 + 12};
            const uint32_t iw = (type == 0 || type == 9) ? wmo : type == 2 ? wb : type == 8 ? wm : type == 10 ? wr : wv;
            if ((size_t)n > (r.size - r.at) / (fixed[type] + iw)) fail(SKINRT_E_RANGE, "morph offsets exceed the file");
            if (type != 1) { r.skip((size_t)n * (fixed[type] + iw)); continue; }
            std::vector<std::pair<uint32_t, std::array<float, 3>>> off((size_t)n);
            for (auto& o : off) {
                o.first = r.index(wv);
                if (o.first == kNone || o.first >= V) fail(SKINRT_E_RANGE, "vertex morph names a vertex that does not exist");
                r.f32s(o.second.data(), 3);
            }
            std::stable_sort(off.begin(), off.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
            FlatTarget t;
            t.dense = false;
            for (const auto& o : off) {
                if (!t.indices.empty() && t.indices.back() == o.first) {  // a vertex listed twice: offsets add up, file order
                    float* last = &t.position[t.position.size() - 3];
                    for (int e = 0; e < 3; ++e) last[e] += o.second[e];
                } else {
                    t.indices.push_back(o.first);
                    t.position.insert(t.position.end(), o.second.begin(), o.second.end());
                }
            }
            rig.targets.push_back(std::move(t));
            info.vertex_morphs += 1;
        }
        rig.default_weights.assign(rig.targets.size(), 0.0f);

        // ---- bones -> nodes in DFS pre-order (roots and children by bone index), joints = bones
        for (uint32_t k = 0; k < B; ++k)
            if (bones[k].parent != kNone && bones[k].parent >= B) fail(SKINRT_E_RANGE, "bone names a parent that does not exist");
        std::vector<std::vector<uint32_t>> children(B);
        std::vector<uint32_t> roots;
        for (uint32_t k = 0; k < B; ++k) {
            if (bones[k].parent == kNone) roots.push_back(k); else children[bones[k].parent].push_back(k);
        }
        std::vector<uint32_t> flat_of(B, kNone);
        std::vector<std::pair<uint32_t, uint32_t>> stack;  // (bone, next child)
        for (uint32_t root : roots) {
            stack.push_back({root, 0});
            while (!stack.empty()) {
                auto& [b, next] = stack.back();
                if (next == 0) {
                    flat_of[b] = (uint32_t)rig.parent.size();
                    const uint32_t pb = bones[b].parent;
                    rig.node_id.push_back(AssetIndex::type_index(id, BundleAssetType::Node, b));
                    rig.parent.push_back(pb == kNone ? -1 : (int32_t)flat_of[pb]);
                    for (int e = 0; e < 3; ++e) rig.translation.push_back(pb == kNone ? bones[b].pos[e] : bones[b].pos[e] - bones[pb].pos[e]);
                    rig.rotation.insert(rig.rotation.end(), {0.0f, 0.0f, 0.0f, 1.0f});
                    rig.scale.insert(rig.scale.end(), {1.0f, 1.0f, 1.0f});
                }
                if (next < children[b].size()) {
                    const uint32_t c = children[b][next++];
                    stack.push_back({c, 0});
                } else {
                    stack.pop_back();
                }
            }
        }
        if (rig.parent.size() != B) fail(SKINRT_E_INVALID, "bone hierarchy has a cycle");
        rig.joint_node.resize(B);
        rig.inverse_bind.assign((size_t)B * 16, 0.0f);
        for (uint32_t k = 0; k < B; ++k) {
            rig.joint_node[k] = flat_of[k];
            float* m = &rig.inverse_bind[(size_t)k * 16];  // column-major translation by -position
            m[0] = m[5] = m[10] = m[15] = 1.0f;
            m[12] = -bones[k].pos[0]; m[13] = -bones[k].pos[1]; m[14] = -bones[k].pos[2];
        }
        // vertex bone indices -> joints (u16, as the asset model's Joints = Vec<[u16;4]>, primitive.rs:19)
        if (B > 65536) fail(SKINRT_E_RANGE, "more than 65536 bones: joint indices are u16 in the asset model");
        for (size_t k = 0; k < vertex_bones.size(); ++k) {
            const uint32_t b = vertex_bones[k];
            if (b == kNone) { rig.weights[0][k] = 0.0f; continue; }  // "none": joint 0, no influence
            if (b >= B) fail(SKINRT_E_RANGE, "vertex names a bone that does not exist");
            rig.joints[0][k] = (uint16_t)b;
        }
        // tangents: calculate_tangent over the indexed surface list, padded with w = 1 (mod.rs:228-231)
        {
            std::vector<std::array<float, 3>> pos(V), tg;
            for (uint32_t v = 0; v < V; ++v) pos[v] = {rig.position[(size_t)v * 3], rig.position[(size_t)v * 3 + 1], rig.position[(size_t)v * 3 + 2]};
            if (!calculate_tangent(PrimitiveAssetMode::TriangleList, pos, &info.indices, tg)) fail(SKINRT_E_INVALID, "tangent generation failed");
            rig.tangent.resize((size_t)V * 4);
            for (uint32_t v = 0; v < V; ++v) {
                for (int e = 0; e < 3; ++e) rig.tangent[(size_t)v * 4 + e] = tg[v][e];
                rig.tangent[(size_t)v * 4 + 3] = 1.0f;
            }
        }
        return SKINRT_OK;
    } catch (const Fail& f) {
        err = f.what;
        rig = FlatRig();
        return f.code;
    } catch (const std::bad_alloc&) {
        err = "out of memory";
        rig = FlatRig();
        return SKINRT_E_NOMEM;
    }
}

}  // namespace skinrt_host
