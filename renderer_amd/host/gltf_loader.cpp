// gltf_loader.cpp — see gltf_loader.hpp.  Compiled with -ffp-contract=off: calculate_tangent's
// arithmetic follows glam's scalar Vec3 (cross, dot, length_recip, mul) operation by operation.
#include "gltf_loader.hpp"

#include <cmath>
#include <cstring>
#include <map>

#include "../../include/skinrt.h"
#include "json_min.hpp"

namespace skinrt_host {
namespace {

struct Fail {
    int code;
    std::string what;
};
[[noreturn]] void fail(int code, const std::string& what) { throw Fail{code, what}; }

enum DataType { I8 = 5120, U8 = 5121, I16 = 5122, U16 = 5123, U32 = 5125, F32 = 5126 };

struct Doc {
    json::Value root;
    std::vector<std::vector<uint8_t>> buffers;
    BundleIndex id;
};

struct Accessor {
    const json::Value* j;
    uint32_t data_type;
    std::string dims;
    size_t count;
};

Accessor accessor(const Doc& d, uint64_t index) {
    const json::Value& a = d.root["accessors"][(size_t)index];
    if (!a.is_object()) fail(SKINRT_E_RANGE, "accessor index out of range");
    Accessor r{&a, (uint32_t)a["componentType"].as_u64(), a["type"].str, (size_t)a["count"].as_u64()};
    switch (r.data_type) {
        case I8: case U8: case I16: case U16: case U32: case F32: break;
        default: fail(SKINRT_E_INVALID, "accessor componentType invalid");
    }
    return r;
}

size_t dims_items(const std::string& t) {
    if (t == "SCALAR") return 1;
    if (t == "VEC2") return 2;
    if (t == "VEC3") return 3;
    if (t == "VEC4") return 4;
    if (t == "MAT2") return 4;
    if (t == "MAT3") return 9;
    if (t == "MAT4") return 16;
    fail(SKINRT_E_INVALID, "accessor type invalid");
}

// load_accessor (loader/gltf/mod.rs:240-295)
std::vector<uint8_t> load_accessor(const Doc& d, const Accessor& a) {
    const json::Value& j = *a.j;
    if (!j.has("bufferView")) {  // `count` zero bytes, as the reference
        if (a.count > ((size_t)1 << 32)) fail(SKINRT_E_RANGE, "accessor count too large");
        return std::vector<uint8_t>(a.count, 0);
    }
    const size_t num_size = (a.data_type == I8 || a.data_type == U8) ? 1 : (a.data_type == I16 || a.data_type == U16) ? 2 : 4;
    const size_t item_length = num_size * dims_items(a.dims);
    const json::Value& view = d.root["bufferViews"][(size_t)j["bufferView"].as_u64()];
    if (!view.is_object()) fail(SKINRT_E_RANGE, "bufferView index out of range");
    const size_t bi = (size_t)view["buffer"].as_u64();
    if (bi >= d.buffers.size()) fail(SKINRT_E_RANGE, "buffer index out of range");
    const std::vector<uint8_t>& buffer = d.buffers[bi];
    const char* past = "accessor reads past its buffer (reference: assert, loader/gltf/mod.rs:272)";
    const uint64_t off_a = j["byteOffset"].as_u64(), off_v = view["byteOffset"].as_u64();
    if (off_a > buffer.size() || off_v > buffer.size() || off_a + off_v > buffer.size()) fail(SKINRT_E_RANGE, past);
    const size_t offset = (size_t)(off_a + off_v);
    const size_t stride = view.has("byteStride") ? (size_t)view["byteStride"].as_u64() : item_length;
    if (stride < item_length) fail(SKINRT_E_INVALID, "bufferView byteStride smaller than the element");
    // the reference checks offset + count * stride (a whole stride after the last element too); same
    // condition, written so that a hostile count or stride cannot wrap it
    if (a.count != 0 && a.count > (buffer.size() - offset) / stride) fail(SKINRT_E_RANGE, past);
    std::vector<uint8_t> out;
    out.reserve(a.count * item_length);
    for (size_t k = 0; k < a.count; ++k) {
        const uint8_t* src = buffer.data() + offset + k * stride;
        out.insert(out.end(), src, src + item_length);
    }
    return out;
}

std::vector<float> load_accessor_f32(const Doc& d, const Accessor& a) {
    if (a.data_type != F32) fail(SKINRT_E_INVALID, "accessor must be FLOAT (reference: assert_eq, loader/gltf/mod.rs:299)");
    const std::vector<uint8_t> raw = load_accessor(d, a);
    std::vector<float> out(raw.size() / 4);
    if (!out.empty()) std::memcpy(out.data(), raw.data(), out.size() * 4);  // little-endian host
    return out;
}
template <class T>
std::vector<T> load_accessor_uint(const Doc& d, const Accessor& a) {
    const std::vector<uint8_t> raw = load_accessor(d, a);
    std::vector<T> out(raw.size() / sizeof(T));
    if (!out.empty()) std::memcpy(out.data(), raw.data(), out.size() * sizeof(T));
    return out;
}
// load_accessor_normalized (:362-370)
std::vector<float> load_accessor_normalized(const Doc& d, const Accessor& a) {
    std::vector<float> out;
    switch (a.data_type) {
        case U8: for (uint8_t v : load_accessor_uint<uint8_t>(d, a)) out.push_back((float)v / 255.0f); break;
        case U16: for (uint16_t v : load_accessor_uint<uint16_t>(d, a)) out.push_back((float)v / 65535.0f); break;
        case U32: for (uint32_t v : load_accessor_uint<uint32_t>(d, a)) out.push_back((float)v / (float)UINT32_MAX); break;
        case F32: out = load_accessor_f32(d, a); break;
        default: fail(SKINRT_E_INVALID, "signed normalised data (reference: unreachable!, loader/gltf/mod.rs:368)");
    }
    return out;
}

void check_accessor(const Accessor& a, uint32_t type, const char* dims) {
    if (a.data_type != type) fail(SKINRT_E_INVALID, "BadAccessorDataType");
    if (a.dims != dims) fail(SKINRT_E_INVALID, "BadAccessorDimensions");
}

template <class T, size_t W>
std::vector<std::array<T, W>> chunk(const std::vector<T>& data) {  // chunks_exact(W)
    std::vector<std::array<T, W>> out(data.size() / W);
    for (size_t k = 0; k < out.size(); ++k)
        for (size_t e = 0; e < W; ++e) out[k][e] = data[k * W + e];
    return out;
}

std::vector<std::array<float, 4>> pad_vec3_to_vec4(const std::vector<std::array<float, 3>>& v, float pad) {
    std::vector<std::array<float, 4>> out;
    out.reserve(v.size());
    for (const auto& x : v) out.push_back({x[0], x[1], x[2], pad});
    return out;
}

PrimitiveAssetMorphTarget load_target(const Doc& d, const json::Value& t, PrimitiveAssetMode mode, const std::vector<uint32_t>* indices) {
    PrimitiveAssetMorphTarget out;
    bool has_position = false;
    if (t.has("POSITION")) {
        const Accessor a = accessor(d, t["POSITION"].as_u64());
        check_accessor(a, F32, "VEC3");
        out.position = chunk<float, 3>(load_accessor_f32(d, a));
        has_position = true;
    }
    if (t.has("NORMAL")) {
        const Accessor a = accessor(d, t["NORMAL"].as_u64());
        check_accessor(a, F32, "VEC3");
        out.normal = chunk<float, 3>(load_accessor_f32(d, a));
    }
    if (t.has("TANGENT")) {
        const Accessor a = accessor(d, t["TANGENT"].as_u64());
        check_accessor(a, F32, "VEC4");  // the reference expects Vec4 here (glTF says VEC3)
        out.tangent = chunk<float, 4>(load_accessor_f32(d, a));
    } else if (has_position) {
        std::vector<std::array<float, 3>> tg;
        if (!calculate_tangent(mode, out.position, indices, tg)) fail(SKINRT_E_INVALID, "tangent generation failed: TriangleStrip (reference: todo!(), tangent.rs:27) or vertex index out of range");
        out.tangent = pad_vec3_to_vec4(tg, 1.0f);
    }
    return out;
}

bool semantic_set(const std::string& name, const char* prefix, size_t& index) {
    const size_t n = std::strlen(prefix);
    if (name.compare(0, n, prefix) != 0 || name.size() == n) return false;
    index = 0;
    for (size_t k = n; k < name.size(); ++k) {
        if (name[k] < '0' || name[k] > '9') return false;
        index = index * 10 + (size_t)(name[k] - '0');
    }
    return index < 64;
}

template <class T>
std::vector<T> take_sets(std::vector<std::pair<bool, T>>& sets, const char* what) {
    std::vector<T> out;
    for (auto& s : sets) {
        if (!s.first) fail(SKINRT_E_INVALID, std::string("Missing ") + what + " set");
        out.push_back(std::move(s.second));
    }
    return out;
}

PrimitiveAsset load_primitive(const Doc& d, const json::Value& p) {
    PrimitiveAsset out;
    if (p.has("indices")) {
        const Accessor a = accessor(d, p["indices"].as_u64());
        std::vector<uint32_t> idx;
        switch (a.data_type) {
            case U8: for (uint8_t v : load_accessor_uint<uint8_t>(d, a)) idx.push_back(v); break;
            case U16: for (uint16_t v : load_accessor_uint<uint16_t>(d, a)) idx.push_back(v); break;
            case U32: idx = load_accessor_uint<uint32_t>(d, a); break;
            default: fail(SKINRT_E_INVALID, "BadAccessorDataType (indices)");
        }
        out.indices = std::move(idx);
    }
    bool has_position = false, has_normal = false, has_tangent = false;
    std::vector<std::pair<bool, std::vector<std::array<float, 2>>>> tex;
    std::vector<std::pair<bool, std::vector<std::array<float, 4>>>> col, wts;
    std::vector<std::pair<bool, std::vector<std::array<uint16_t, 4>>>> jnt;
    auto& at = out.attributes;
    for (const auto& kv : p["attributes"].obj) {
        const std::string& name = kv.first;
        const Accessor a = accessor(d, kv.second.as_u64());
        size_t set = 0;
        if (name == "POSITION") {
            check_accessor(a, F32, "VEC3");
            at.position = chunk<float, 3>(load_accessor_f32(d, a));
            has_position = true;
        } else if (name == "NORMAL") {
            check_accessor(a, F32, "VEC3");
            at.normal = chunk<float, 3>(load_accessor_f32(d, a));
            has_normal = true;
        } else if (name == "TANGENT") {
            check_accessor(a, F32, "VEC4");
            at.tangent = chunk<float, 4>(load_accessor_f32(d, a));
            has_tangent = true;
        } else if (semantic_set(name, "COLOR_", set)) {
            if (col.size() <= set) col.resize(set + 1);
            const std::vector<float> data = load_accessor_normalized(d, a);
            std::vector<std::array<float, 4>> c;
            if (a.dims == "VEC3") {  // NOT clamped: the reference discards the clamped copy (loader/mod.rs:52-53)
                for (const auto& x : chunk<float, 3>(data)) c.push_back({x[0], x[1], x[2], 1.0f});
            } else if (a.dims == "VEC4") {
                for (auto x : chunk<float, 4>(data)) {
                    for (float& v : x) v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
                    c.push_back(x);
                }
            } else fail(SKINRT_E_INVALID, "BadAccessorDimensions (color)");
            col[set] = {true, std::move(c)};
        } else if (semantic_set(name, "TEXCOORD_", set)) {
            if (tex.size() <= set) tex.resize(set + 1);
            if (a.dims != "VEC2") fail(SKINRT_E_INVALID, "BadAccessorDimensions (texcoord)");
            tex[set] = {true, chunk<float, 2>(load_accessor_normalized(d, a))};
        } else if (semantic_set(name, "JOINTS_", set)) {
            if (jnt.size() <= set) jnt.resize(set + 1);
            if (a.dims != "VEC4") fail(SKINRT_E_INVALID, "BadAccessorDimensions (joints)");
            std::vector<uint16_t> data;
            if (a.data_type == U8) { for (uint8_t v : load_accessor_uint<uint8_t>(d, a)) data.push_back(v); }
            else if (a.data_type == U16) data = load_accessor_uint<uint16_t>(d, a);
            else fail(SKINRT_E_INVALID, "BadAccessorDataType (joints)");
            jnt[set] = {true, chunk<uint16_t, 4>(data)};
        } else if (semantic_set(name, "WEIGHTS_", set)) {
            if (wts.size() <= set) wts.resize(set + 1);
            if (a.dims != "VEC4") fail(SKINRT_E_INVALID, "BadAccessorDimensions (weights)");
            wts[set] = {true, chunk<float, 4>(load_accessor_normalized(d, a))};
        }  // application-specific semantics are ignored
    }
    if (!has_position) fail(SKINRT_E_INVALID, "No positions in primitive (reference: expect, loader/gltf/mod.rs:1050)");
    at.tex_coord = take_sets(tex, "texture coordinates");
    at.color = take_sets(col, "vertex color");
    at.joints = take_sets(jnt, "skin joints");
    at.weights = take_sets(wts, "skin weights");
    if (at.joints.size() != at.weights.size()) fail(SKINRT_E_INVALID, "BadJointWeightSets");
    switch (p.has("mode") ? p["mode"].as_u64() : 4) {
        case 0: out.mode = PrimitiveAssetMode::Points; break;
        case 1: out.mode = PrimitiveAssetMode::LineList; break;
        case 3: out.mode = PrimitiveAssetMode::LineStrip; break;
        case 4: out.mode = PrimitiveAssetMode::TriangleList; break;
        case 5: out.mode = PrimitiveAssetMode::TriangleStrip; break;
        default: fail(SKINRT_E_INVALID, "UnsupportedPrimitiveMode");
    }
    const std::vector<uint32_t>* idx = out.indices ? &*out.indices : nullptr;
    if (!has_tangent) {
        std::vector<std::array<float, 3>> tg;
        if (!calculate_tangent(out.mode, at.position, idx, tg)) fail(SKINRT_E_INVALID, "tangent generation failed: TriangleStrip (reference: todo!(), tangent.rs:27) or vertex index out of range");
        at.tangent = pad_vec3_to_vec4(tg, 1.0f);
    }
    if (!has_normal) {  // clip_vec4_to_vec3(&tangent): the normal IS the tangent's xyz
        at.normal.clear();
        for (const auto& x : at.tangent) at.normal.push_back({x[0], x[1], x[2]});
    }
    for (const json::Value& t : p["targets"].arr) out.targets.push_back(load_target(d, t, out.mode, idx));
    return out;
}

MeshAsset load_mesh(const Doc& d, uint64_t index) {
    const json::Value& m = d.root["meshes"][(size_t)index];
    if (!m.is_object()) fail(SKINRT_E_RANGE, "mesh index out of range");
    MeshAsset out;
    if (m["name"].is_string()) out.name = m["name"].str;
    for (const json::Value& p : m["primitives"].arr) out.primitives.push_back(load_primitive(d, p));
    for (const json::Value& w : m["weights"].arr) out.weights.push_back((float)w.as_double());
    return out;
}

struct Loader {
    const Doc& d;
    std::map<uint64_t, std::shared_ptr<const SkinAsset>> skin_cache;

    AssetIndex node_id(uint64_t i) const { return AssetIndex::type_index(d.id, BundleAssetType::Node, (size_t)i); }

    std::shared_ptr<const SkinAsset> load_skin(uint64_t index) {
        auto it = skin_cache.find(index);
        if (it != skin_cache.end()) return it->second;
        const json::Value& s = d.root["skins"][(size_t)index];
        if (!s.is_object()) fail(SKINRT_E_RANGE, "skin index out of range");
        auto skin = std::make_shared<SkinAsset>();
        skin->id = AssetIndex::type_index(d.id, BundleAssetType::Skin, (size_t)index);
        for (const json::Value& j : s["joints"].arr) skin->joint_ids.push_back(node_id(j.as_u64()));
        if (s.has("inverseBindMatrices")) {
            const std::vector<float> data = load_accessor_f32(d, accessor(d, s["inverseBindMatrices"].as_u64()));
            for (const auto& m : chunk<float, 16>(data)) skin->inverse_bind_matrices.push_back(m);
        }
        if (s.has("skeleton")) skin->skeleton = node_id(s["skeleton"].as_u64());
        skin_cache[index] = skin;
        return skin;
    }

    // The reference recurses over `children` without any check (load_node, :1168-1207): a node that
    // is its own ancestor overflows its stack.  Here a cycle, an absurd depth, or a graph that
    // re-visits nodes until the walk explodes is an error.
    std::vector<uint8_t> on_path;
    size_t visited = 0;
    NodeAsset load_node(uint64_t index, int depth) {
        const json::Value& n = d.root["nodes"][(size_t)index];
        if (!n.is_object()) fail(SKINRT_E_RANGE, "node index out of range");
        if (on_path.size() < d.root["nodes"].size()) on_path.assign(d.root["nodes"].size(), 0);
        if (on_path[(size_t)index]) fail(SKINRT_E_INVALID, "node hierarchy has a cycle");
        if (depth > 16384 || ++visited > 4 * on_path.size() + 1024) fail(SKINRT_E_INVALID, "node hierarchy too deep or nodes shared between parents");
        struct Mark { uint8_t& f; explicit Mark(uint8_t& x) : f(x) { f = 1; } ~Mark() { f = 0; } } mark(on_path[(size_t)index]);
        NodeAsset out;
        out.id = node_id(index);
        if (n["name"].is_string()) out.name = n["name"].str;
        if (n.has("matrix")) {
            if (n["matrix"].size() != 16) fail(SKINRT_E_INVALID, "node matrix must have 16 numbers");
            MatrixNodeTransform m;
            for (size_t k = 0; k < 16; ++k) m.m[k] = (float)n["matrix"][k].as_double();
            out.transform = NodeTransform(m);
        } else {
            DecomposedTransform t;
            auto read = [&](const char* key, float* dst, size_t w) {
                if (!n.has(key)) return;
                if (n[key].size() != w) fail(SKINRT_E_INVALID, std::string("node ") + key + " has the wrong length");
                for (size_t k = 0; k < w; ++k) dst[k] = (float)n[key][k].as_double();
            };
            read("translation", t.translation.data(), 3);
            read("rotation", t.rotation.data(), 4);
            read("scale", t.scale.data(), 3);
            out.transform = NodeTransform(t);
        }
        if (n.has("mesh")) out.mesh = load_mesh(d, n["mesh"].as_u64());
        if (n.has("skin")) out.skin = load_skin(n["skin"].as_u64());
        for (const json::Value& c : n["children"].arr) out.children.push_back(load_node(c.as_u64(), depth + 1));
        for (const json::Value& w : n["weights"].arr) out.weights.push_back((float)w.as_double());
        return out;
    }

    // load_animation_sampler (:1219-1300)
    template <class V>
    static float read_keyframes(const std::vector<float>& time, const std::vector<V>& data, bool cubic,
                                AnimationKeyFrames<V>& out, const std::string& interpolation) {
        using KF = AnimationKeyFrames<V>;
        std::vector<AnimationKeyFrame<V>> frames;
        const size_t repeat = cubic ? 3 : 1;
        const size_t n = std::min(time.size() * repeat, data.size());  // zip stops at the shorter side
        for (size_t k = 0; k < n; ++k) frames.push_back({time[k / repeat], data[k]});
        float length = 0.0f;
        bool first = true;
        for (const auto& f : frames) {  // max_by(total_cmp)
            if (first || !(f.time < length)) length = f.time;
            first = false;
        }
        if (cubic) {
            out.kind = KF::Kind::CubicSpline;
            for (size_t k = 0; k + 3 <= frames.size(); k += 3)
                out.cubic_frames.push_back({frames[k].time, {frames[k].value, frames[k + 1].value, frames[k + 2].value}});
        } else {
            out.kind = interpolation == "STEP" ? KF::Kind::Step : KF::Kind::Linear;
            out.frames = std::move(frames);
        }
        return length;
    }

    AnimationAsset load_animation(const json::Value& a) {
        AnimationAsset out;
        if (a["name"].is_string()) out.name = a["name"].str;
        for (const json::Value& ch : a["channels"].arr) {
            const json::Value& target = ch["target"];
            const std::string& path = target["path"].str;
            const json::Value& sampler = a["samplers"][(size_t)ch["sampler"].as_u64()];
            if (!sampler.is_object()) fail(SKINRT_E_RANGE, "animation sampler index out of range");
            const std::string interpolation = sampler["interpolation"].is_string() ? sampler["interpolation"].str : "LINEAR";
            if (interpolation != "LINEAR" && interpolation != "STEP" && interpolation != "CUBICSPLINE")
                fail(SKINRT_E_INVALID, "sampler interpolation invalid");
            const bool cubic = interpolation == "CUBICSPLINE";
            AnimationChannelAsset c;
            const std::vector<float> time = load_accessor_f32(d, accessor(d, sampler["input"].as_u64()));
            const Accessor outa = accessor(d, sampler["output"].as_u64());
            if (path == "rotation") {
                c.sampler.kind = AnimationSampler::Kind::Rotation;
                c.length = read_keyframes(time, chunk<float, 4>(load_accessor_normalized(d, outa)), cubic, c.sampler.rotation, interpolation);
            } else if (path == "translation" || path == "scale") {
                c.sampler.kind = path == "scale" ? AnimationSampler::Kind::Scale : AnimationSampler::Kind::Translation;
                c.length = read_keyframes(time, chunk<float, 3>(load_accessor_f32(d, outa)), cubic, c.sampler.vec3, interpolation);
            } else if (path == "weights") {
                // Property::MorphTargetWeights => todo!() in the reference (loader/gltf/mod.rs:1312).  glTF: the output
                // accessor holds M scalars per key (3 M for CUBICSPLINE), float or normalised integer, M = the number of
                // morph targets of the node's mesh; read like the other samplers (times repeated for the spline triple).
                if (outa.dims != "SCALAR") fail(SKINRT_E_INVALID, "BadAccessorDimensions (animation weights)");
                const std::vector<float> flat = load_accessor_normalized(d, outa);
                const size_t rows = time.size() * (cubic ? 3 : 1);
                if (rows == 0 || flat.size() % rows != 0 || flat.empty()) fail(SKINRT_E_INVALID, "animation weights: output count is not keys x targets");
                const size_t M = flat.size() / rows;
                std::vector<std::vector<float>> data(rows);
                for (size_t k = 0; k < rows; ++k) data[k].assign(flat.begin() + k * M, flat.begin() + (k + 1) * M);
                c.sampler.kind = AnimationSampler::Kind::MorphTargetWeights;
                c.length = read_keyframes(time, data, cubic, c.sampler.weights, interpolation);
            } else {
                fail(SKINRT_E_INVALID, "animation target path invalid");
            }
            if (!target.has("node")) fail(SKINRT_E_INVALID, "animation channel without a target node");
            c.target_id = node_id(target["node"].as_u64());
            out.channels.push_back(std::move(c));
        }
        return out;
    }
};

uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

}  // namespace

bool calculate_tangent(PrimitiveAssetMode mode, const std::vector<std::array<float, 3>>& positions,
                       const std::vector<uint32_t>* indices, std::vector<std::array<float, 3>>& out) {
    out.assign(positions.size(), {0.0f, 0.0f, 0.0f});
    auto triangle = [&](size_t i0, size_t i1, size_t i2) {
        const auto& p0 = positions[i0];
        const auto& p1 = positions[i1];
        const auto& p2 = positions[i2];
        const float a[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};  // vec_a = pnt_2 - pnt_1
        const float b[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};  // vec_b = pnt_0 - pnt_1
        // glam Vec3::cross
        const float c[3] = {a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]};
        // normalize = self * (1 / sqrt(dot(self, self)))
        const float r = 1.0f / std::sqrt((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]);
        const float nrm[3] = {c[0] * r, c[1] * r, c[2] * r};
        for (size_t i : {i0, i1, i2})
            for (int e = 0; e < 3; ++e) out[i][e] += nrm[e];
    };
    switch (mode) {
        case PrimitiveAssetMode::Points: case PrimitiveAssetMode::LineStrip: case PrimitiveAssetMode::LineList: break;
        case PrimitiveAssetMode::TriangleStrip: return false;
        case PrimitiveAssetMode::TriangleList:
            if (indices) {
                for (size_t t = 0; t + 3 <= indices->size(); t += 3) {
                    const size_t i0 = (*indices)[t], i1 = (*indices)[t + 1], i2 = (*indices)[t + 2];
                    if (i0 >= positions.size() || i1 >= positions.size() || i2 >= positions.size()) return false;  // reference: index panic
                    triangle(i0, i1, i2);
                }
            } else {
                for (size_t t = 0; t + 3 <= positions.size(); t += 3) triangle(t, t + 1, t + 2);
            }
            break;
    }
    for (auto& v : out) {
        const float r = 1.0f / std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
        v = {v[0] * r, v[1] * r, v[2] * r};
    }
    return true;
}

int load_glb_from_buffer_with_id(const uint8_t* buffer, size_t size, const BundleIndex& id,
                                 std::vector<SceneAsset>& scenes, std::vector<AnimationAsset>& animations,
                                 std::string& err) {
    scenes.clear();
    animations.clear();
    try {
        if (!buffer || size < 20 || rd32(buffer) != 0x46546C67u) fail(SKINRT_E_INVALID, "not a GLB container (magic)");
        if (rd32(buffer + 4) != 2) fail(SKINRT_E_INVALID, "unsupported GLB version");
        const size_t total = rd32(buffer + 8);
        if (total > size) fail(SKINRT_E_RANGE, "GLB length exceeds the buffer");
        Doc d;
        d.id = id;
        size_t at = 12;
        bool have_json = false;
        std::vector<uint8_t> bin;
        bool have_bin = false;
        while (at + 8 <= total) {
            const size_t len = rd32(buffer + at);
            const uint32_t type = rd32(buffer + at + 4);
            at += 8;
            if (at + len > total) fail(SKINRT_E_RANGE, "GLB chunk exceeds the file");
            if (type == 0x4E4F534Au && !have_json) {  // JSON
                d.root = json::parse(reinterpret_cast<const char*>(buffer + at), len);
                have_json = true;
            } else if (type == 0x004E4942u && !have_bin) {  // BIN
                bin.assign(buffer + at, buffer + at + len);
                have_bin = true;
            }
            at += (len + 3) & ~(size_t)3;
        }
        if (!have_json || !d.root.is_object()) fail(SKINRT_E_INVALID, "GLB without a JSON chunk");
        for (size_t b = 0; b < d.root["buffers"].size(); ++b) {
            const json::Value& jb = d.root["buffers"][b];
            if (jb.has("uri")) fail(SKINRT_E_INVALID, "external / data-URI buffers are not supported by this loader (GLB BIN chunk only)");
            if (b != 0 || !have_bin) fail(SKINRT_E_INVALID, "buffer without data");
            if (jb["byteLength"].as_u64() > bin.size()) fail(SKINRT_E_RANGE, "buffer byteLength exceeds the BIN chunk");
            d.buffers.push_back(bin);
        }
        Loader L{d, {}, {}, 0};
        for (const json::Value& a : d.root["animations"].arr) animations.push_back(L.load_animation(a));
        for (const json::Value& s : d.root["scenes"].arr) {
            SceneAsset scene;
            if (s["name"].is_string()) scene.name = s["name"].str;
            for (const json::Value& n : s["nodes"].arr) scene.nodes.push_back(L.load_node(n.as_u64(), 0));
            scenes.push_back(std::move(scene));
        }
        return SKINRT_OK;
    } catch (const Fail& f) {
        err = f.what;
        return f.code;
    } catch (const json::ParseError& e) {
        err = e.what();
        return SKINRT_E_INVALID;
    } catch (const std::bad_alloc&) {
        err = "out of memory";
        return SKINRT_E_NOMEM;
    }
}

}  // namespace skinrt_host
