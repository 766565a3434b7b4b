// flatten.cpp — see flatten.hpp.
#include "flatten.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace skinrt_host {
namespace {

struct Walk {
    FlatRig& rig;
    std::map<AssetIndex, uint32_t> index_of;  // first occurrence wins (a node shared by two scenes is two copies)
    struct MeshRef { uint32_t node; const NodeAsset* src; };
    std::vector<MeshRef> skinned;
    std::shared_ptr<const SkinAsset> skin;
    const AssetIndex* wanted_skin;

    void node(const NodeAsset& n, int32_t parent) {
        const uint32_t me = rig.node_count();
        rig.node_id.push_back(n.id);
        rig.parent.push_back(parent);
        index_of.emplace(n.id, me);
        DecomposedTransform trs;  // transform: None = identity
        Mat4 m = mat4_identity();
        uint8_t is_matrix = 0;
        if (n.transform) {
            if (const auto* d = std::get_if<DecomposedTransform>(&*n.transform)) trs = *d;
            else { m = std::get<MatrixNodeTransform>(*n.transform).m; is_matrix = 1; }
        }
        rig.translation.insert(rig.translation.end(), trs.translation.begin(), trs.translation.end());
        rig.rotation.insert(rig.rotation.end(), trs.rotation.begin(), trs.rotation.end());
        rig.scale.insert(rig.scale.end(), trs.scale.begin(), trs.scale.end());
        rig.matrix.insert(rig.matrix.end(), m.begin(), m.end());
        rig.has_matrix.push_back(is_matrix);
        if (n.skin && n.mesh) {
            if (!skin && (!wanted_skin || n.skin->id == *wanted_skin)) skin = n.skin;
            if (skin && n.skin->id == skin->id) skinned.push_back({me, &n});
        }
        for (const NodeAsset& c : n.children) node(c, (int32_t)me);
    }
};

template <class V>
void push_keys(const AnimationKeyFrames<V>& k, FlatChannel& ch) {
    using Kind = typename AnimationKeyFrames<V>::Kind;
    if (k.kind == Kind::CubicSpline) {
        ch.interpolation = SKINRT_INTERP_CUBICSPLINE;
        for (const auto& f : k.cubic_frames) {
            ch.times.push_back(f.time);
            for (const auto* v : {&f.value.in_tangent, &f.value.value, &f.value.out_tangent})
                ch.values.insert(ch.values.end(), v->begin(), v->end());
        }
    } else {
        ch.interpolation = k.kind == Kind::Step ? SKINRT_INTERP_STEP : SKINRT_INTERP_LINEAR;
        for (const auto& f : k.frames) {
            ch.times.push_back(f.time);
            ch.values.insert(ch.values.end(), f.value.begin(), f.value.end());
        }
    }
}

int fail(std::string& err, int code, const std::string& what) {
    err = what;
    return code;
}

}  // namespace

int flatten(const std::vector<SceneAsset>& scenes, const std::vector<AnimationAsset>& animations,
            const FlattenOptions& opt, FlatRig& rig, std::string& err) {
    rig = FlatRig();
    if (opt.scene >= (int)scenes.size()) return fail(err, SKINRT_E_RANGE, "scene index out of range");
    Walk w{rig, {}, {}, nullptr, opt.skin};
    for (size_t s = 0; s < scenes.size(); ++s) {
        if (opt.scene >= 0 && (size_t)opt.scene != s) continue;
        for (const NodeAsset& n : scenes[s].nodes) w.node(n, -1);
    }
    if (rig.parent.empty()) return fail(err, SKINRT_E_INVALID, "no nodes");
    if (std::none_of(rig.has_matrix.begin(), rig.has_matrix.end(), [](uint8_t b) { return b != 0; })) {
        rig.matrix.clear();
        rig.has_matrix.clear();
    }

    // skin: joint -> node, inverse bind matrices (a8)
    if (w.skin) {
        for (const AssetIndex& j : w.skin->joint_ids) {
            auto it = w.index_of.find(j);
            if (it == w.index_of.end()) return fail(err, SKINRT_E_RANGE, "skin joint id is not a node of the flattened scenes");
            rig.joint_node.push_back(it->second);
        }
        const auto& ibm = w.skin->inverse_bind_matrices;
        if (!ibm.empty()) {
            if (ibm.size() != w.skin->joint_ids.size())
                return fail(err, SKINRT_E_INVALID, "inverse_bind_matrices length != joint_ids length");
            for (const Mat4& m : ibm) rig.inverse_bind.insert(rig.inverse_bind.end(), m.begin(), m.end());
        }
    }

    // clips (a1): one per AnimationAsset, channels in file order
    for (const AnimationAsset& a : animations) {
        std::vector<FlatChannel> clip;
        float length = 0;
        for (const AnimationChannelAsset& c : a.channels) {
            auto it = w.index_of.find(c.target_id);
            if (it == w.index_of.end()) return fail(err, SKINRT_E_RANGE, "animation channel targets an unknown node");
            FlatChannel ch;
            ch.node = it->second;
            switch (c.sampler.kind) {
                case AnimationSampler::Kind::Rotation: ch.path = SKINRT_PATH_ROTATION; push_keys(c.sampler.rotation, ch); break;
                case AnimationSampler::Kind::Translation: ch.path = SKINRT_PATH_TRANSLATION; push_keys(c.sampler.vec3, ch); break;
                case AnimationSampler::Kind::Scale: ch.path = SKINRT_PATH_SCALE; push_keys(c.sampler.vec3, ch); break;
                case AnimationSampler::Kind::MorphTargetWeights: ch.path = SKINRT_PATH_WEIGHTS; push_keys(c.sampler.weights, ch); break;
            }
            length = std::max(length, c.length);
            clip.push_back(std::move(ch));
        }
        rig.clips.push_back(std::move(clip));
        rig.clip_length.push_back(length);
    }

    // mesh (a11, a13): primitives of the nodes that use the skin, concatenated
    size_t K = 0, M = 0, prims = 0;
    for (const auto& mr : w.skinned)
        for (const PrimitiveAsset& p : mr.src->mesh->primitives) {
            K = std::max(K, std::min(p.attributes.joints.size(), p.attributes.weights.size()));
            M = std::max(M, p.targets.size());
            ++prims;
        }
    if (K > 4) return fail(err, SKINRT_E_LIMIT, "more than 4 joint/weight sets");
    rig.joints.assign(K, {});
    rig.weights.assign(K, {});
    rig.targets.assign(M, {});
    for (FlatTarget& t : rig.targets) t.dense = prims == 1;
    std::vector<bool> any_normal(M, false), any_tangent(M, false);
    for (const auto& mr : w.skinned)
        for (const PrimitiveAsset& p : mr.src->mesh->primitives)
            for (size_t m = 0; m < p.targets.size(); ++m) {
                any_normal[m] = any_normal[m] || !p.targets[m].normal.empty();
                any_tangent[m] = any_tangent[m] || !p.targets[m].tangent.empty();
            }
    for (const auto& mr : w.skinned) {
        const MeshAsset& mesh = *mr.src->mesh;
        for (size_t pi = 0; pi < mesh.primitives.size(); ++pi) {
            const PrimitiveAsset& p = mesh.primitives[pi];
            const auto& at = p.attributes;
            const size_t n = at.position.size();
            const uint32_t first = rig.vertex_count();
            if ((!at.normal.empty() && at.normal.size() != n) || (!at.tangent.empty() && at.tangent.size() != n))
                return fail(err, SKINRT_E_INVALID, "primitive attribute lengths differ");
            for (size_t v = 0; v < n; ++v) {
                rig.position.insert(rig.position.end(), at.position[v].begin(), at.position[v].end());
                const std::array<float, 3> nz{0, 0, 0};
                const std::array<float, 4> tz{0, 0, 0, 0};
                const auto& nn = at.normal.empty() ? nz : at.normal[v];
                const auto& tt = at.tangent.empty() ? tz : at.tangent[v];
                rig.normal.insert(rig.normal.end(), nn.begin(), nn.end());
                rig.tangent.insert(rig.tangent.end(), tt.begin(), tt.end());
            }
            for (size_t k = 0; k < K; ++k) {
                const bool have = k < at.joints.size() && k < at.weights.size();
                if (have && (at.joints[k].size() != n || at.weights[k].size() != n))
                    return fail(err, SKINRT_E_INVALID, "joint/weight set length differs from position length");
                for (size_t v = 0; v < n; ++v) {
                    const std::array<uint16_t, 4> jz{0, 0, 0, 0};
                    const std::array<float, 4> wz{0, 0, 0, 0};
                    const auto& jj = have ? at.joints[k][v] : jz;
                    const auto& ww = have ? at.weights[k][v] : wz;
                    rig.joints[k].insert(rig.joints[k].end(), jj.begin(), jj.end());
                    rig.weights[k].insert(rig.weights[k].end(), ww.begin(), ww.end());
                }
            }
            for (size_t m = 0; m < p.targets.size(); ++m) {
                const PrimitiveAssetMorphTarget& src = p.targets[m];
                FlatTarget& dst = rig.targets[m];
                if (src.position.size() != n || (!src.normal.empty() && src.normal.size() != n) ||
                    (!src.tangent.empty() && src.tangent.size() != n))
                    return fail(err, SKINRT_E_INVALID, "morph target length differs from position length");
                for (size_t v = 0; v < n; ++v) {
                    if (!dst.dense) dst.indices.push_back(first + (uint32_t)v);
                    dst.position.insert(dst.position.end(), src.position[v].begin(), src.position[v].end());
                    if (any_normal[m]) {
                        const std::array<float, 3> z{0, 0, 0};
                        const auto& x = src.normal.empty() ? z : src.normal[v];
                        dst.normal.insert(dst.normal.end(), x.begin(), x.end());
                    }
                    if (any_tangent[m]) {
                        // Deviation from the loader's data, on purpose: a target without a TANGENT accessor gets
                        // calculate_tangent(delta positions) from the loader (as the reference, loader/gltf/mod.rs:911-916),
                        // which is 0 * inf = NaN for every vertex whose triangles do not move.  A NaN delta would turn the
                        // vertex's tangent into NaN at ANY weight (0 * NaN), so non-finite tangent deltas become zero here.
                        const std::array<float, 4> z{0, 0, 0, 0};
                        std::array<float, 4> x = src.tangent.empty() ? z : src.tangent[v];
                        for (float& e : x)
                            if (!std::isfinite(e)) { e = 0.0f; ++rig.sanitized_tangent_deltas; }
                        dst.tangent.insert(dst.tangent.end(), x.begin(), x.end());
                    }
                }
            }
            rig.ranges.push_back({mr.node, (uint32_t)pi, first, (uint32_t)n});
        }
    }
    // default weights (a13): node.weights if non-empty, else mesh.weights, else zeros
    if (M && !w.skinned.empty()) {
        const NodeAsset& n0 = *w.skinned.front().src;
        const std::vector<float>& src = !n0.weights.empty() ? n0.weights : n0.mesh->weights;
        rig.default_weights.assign(M, 0.0f);
        std::copy_n(src.begin(), std::min(src.size(), M), rig.default_weights.begin());
    }
    // weights channels: the C ABI reads key_count x M (x 3) floats — check the width here, where the lengths are known.
    // A channel aimed at a mesh node that is not part of the flattened skin is dropped (its mesh is not in the rig).
    for (auto& clip : rig.clips) {
        for (size_t k = 0; k < clip.size();) {
            FlatChannel& ch = clip[k];
            if (ch.path != SKINRT_PATH_WEIGHTS) { ++k; continue; }
            const bool ours = std::any_of(w.skinned.begin(), w.skinned.end(), [&](const auto& sk) { return sk.node == ch.node; });
            if (!ours || M == 0) { clip.erase(clip.begin() + (long)k); continue; }
            const size_t per_key = (ch.interpolation == SKINRT_INTERP_CUBICSPLINE ? 3 : 1) * M;
            if (ch.values.size() != ch.times.size() * per_key)
                return fail(err, SKINRT_E_INVALID, "weights channel: values per key != morph target count of the mesh");
            ++k;
        }
    }
    return SKINRT_OK;
}

const skinrt_rig_desc* FlatRig::desc() {
    auto ptr = [](const auto& v) { return v.empty() ? nullptr : v.data(); };
    std::memset(&desc_, 0, sizeof desc_);
    desc_.struct_size = sizeof desc_;
    desc_.node_count = node_count();
    desc_.parent = ptr(parent);
    desc_.translation = ptr(translation);
    desc_.rotation = ptr(rotation);
    desc_.scale = ptr(scale);
    desc_.matrix = ptr(matrix);
    desc_.has_matrix = ptr(has_matrix);
    desc_.joint_count = (uint32_t)joint_node.size();
    desc_.joint_node = ptr(joint_node);
    desc_.inverse_bind = ptr(inverse_bind);
    clip_descs_.assign(clips.size(), {});
    channel_descs_.assign(clips.size(), {});
    for (size_t c = 0; c < clips.size(); ++c) {
        for (const FlatChannel& ch : clips[c]) {
            skinrt_channel_desc d{};
            d.target_node = ch.node;
            d.path = ch.path;
            d.interpolation = ch.interpolation;
            d.key_count = (uint32_t)ch.times.size();
            d.times = ptr(ch.times);
            d.values = ptr(ch.values);
            channel_descs_[c].push_back(d);
        }
        clip_descs_[c].channel_count = (uint32_t)channel_descs_[c].size();
        clip_descs_[c].channels = ptr(channel_descs_[c]);
    }
    desc_.clip_count = (uint32_t)clips.size();
    desc_.clips = ptr(clip_descs_);
    desc_.vertex_count = vertex_count();
    desc_.position = ptr(position);
    desc_.normal = ptr(normal);
    desc_.tangent = ptr(tangent);
    desc_.tangent_stride = 4;
    desc_.set_count = (uint32_t)joints.size();
    joint_ptrs_.clear();
    weight_ptrs_.clear();
    for (size_t k = 0; k < joints.size(); ++k) {
        joint_ptrs_.push_back(ptr(joints[k]));
        weight_ptrs_.push_back(ptr(weights[k]));
    }
    desc_.joints = ptr(joint_ptrs_);
    desc_.weights = ptr(weight_ptrs_);
    target_descs_.assign(targets.size(), {});
    for (size_t m = 0; m < targets.size(); ++m) {
        const FlatTarget& t = targets[m];
        target_descs_[m].count = (uint32_t)(t.position.size() / 3);
        target_descs_[m].indices = t.dense ? nullptr : ptr(t.indices);
        target_descs_[m].position = ptr(t.position);
        target_descs_[m].normal = ptr(t.normal);
        target_descs_[m].tangent = ptr(t.tangent);
    }
    desc_.target_count = (uint32_t)targets.size();
    desc_.targets = ptr(target_descs_);
    desc_.default_weights = ptr(default_weights);
    return &desc_;
}

}  // namespace skinrt_host
