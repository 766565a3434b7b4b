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
    struct MeshRef { uint32_t node; const NodeAsset* src; int32_t skin; };
    std::vector<MeshRef> meshes;                           // every node whose mesh becomes a section, DFS order
    std::vector<std::shared_ptr<const SkinAsset>> skins;   // distinct skins by id, in the order they are met
    const AssetIndex* wanted_skin;
    bool include_unskinned;

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
        if (n.mesh && n.skin) {  // node.rs:95-96: the node names its skin (shared by Arc, cached per glTF skin: loader/gltf/mod.rs:1124-1128)
            if (!wanted_skin || n.skin->id == *wanted_skin) {
                size_t k = 0;
                while (k < skins.size() && !(skins[k]->id == n.skin->id)) ++k;
                if (k == skins.size()) skins.push_back(n.skin);
                meshes.push_back({me, &n, (int32_t)k});
            }
        } else if (n.mesh && !wanted_skin && include_unskinned) {
            meshes.push_back({me, &n, -1});  // drawn with the node's world matrix as InstanceUniform (shader.wgsl:143-152)
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
    Walk w{rig, {}, {}, {}, opt.skin, opt.include_unskinned};
    for (size_t s = 0; s < scenes.size(); ++s) {
        if (opt.scene >= 0 && (size_t)opt.scene != s) continue;
        for (const NodeAsset& n : scenes[s].nodes) w.node(n, -1);
    }
    if (rig.parent.empty()) return fail(err, SKINRT_E_INVALID, "no nodes");
    if (std::none_of(rig.has_matrix.begin(), rig.has_matrix.end(), [](uint8_t b) { return b != 0; })) {
        rig.matrix.clear();
        rig.has_matrix.clear();
    }

    // skins: joint -> node, inverse bind matrices (a8)
    for (const auto& sk : w.skins) {
        FlatSkin fs;
        fs.id = sk->id;
        for (const AssetIndex& j : sk->joint_ids) {
            auto it = w.index_of.find(j);
            if (it == w.index_of.end()) return fail(err, SKINRT_E_RANGE, "skin joint id is not a node of the flattened scenes");
            fs.joint_node.push_back(it->second);
        }
        const auto& ibm = sk->inverse_bind_matrices;
        if (!ibm.empty()) {
            if (ibm.size() != sk->joint_ids.size())
                return fail(err, SKINRT_E_INVALID, "inverse_bind_matrices length != joint_ids length");
            for (const Mat4& m : ibm) fs.inverse_bind.insert(fs.inverse_bind.end(), m.begin(), m.end());
        }
        rig.skins.push_back(std::move(fs));
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

    // mesh (a11, a13): one section per mesh node, its primitives concatenated; the morph targets of a section are its
    // MeshAsset's (every primitive of a glTF mesh carries the same number of targets, weights live on the mesh: mesh.rs:4-8)
    size_t K = 0;
    for (const auto& mr : w.meshes)
        if (mr.skin >= 0)
            for (const PrimitiveAsset& p : mr.src->mesh->primitives)
                K = std::max(K, std::min(p.attributes.joints.size(), p.attributes.weights.size()));
    if (K > 4) return fail(err, SKINRT_E_LIMIT, "more than 4 joint/weight sets");
    if (K == 0 && !w.meshes.empty()) K = 1;  // un-skinned meshes only: one (unread) set keeps the descriptor regular
    rig.joints.assign(K, {});
    rig.weights.assign(K, {});
    for (const auto& mr : w.meshes) {
        const MeshAsset& mesh = *mr.src->mesh;
        size_t M = 0;
        bool all_have_all = true;
        for (const PrimitiveAsset& p : mesh.primitives) M = std::max(M, p.targets.size());
        for (const PrimitiveAsset& p : mesh.primitives) all_have_all = all_have_all && p.targets.size() == M;
        skinrt_section_desc sec{};
        sec.node = mr.node;
        sec.skin = mr.skin;
        sec.first_vertex = rig.vertex_count();
        sec.first_target = (uint32_t)rig.targets.size();
        sec.target_count = (uint32_t)M;
        const uint32_t section_index = (uint32_t)rig.sections.size();
        rig.targets.resize(rig.targets.size() + M);
        FlatTarget* tg = rig.targets.data() + sec.first_target;
        for (size_t m = 0; m < M; ++m) tg[m].dense = all_have_all;  // every primitive brings its rows: the concatenation is dense
        std::vector<bool> any_normal(M, false), any_tangent(M, false);
        for (const PrimitiveAsset& p : mesh.primitives)
            for (size_t m = 0; m < p.targets.size(); ++m) {
                any_normal[m] = any_normal[m] || !p.targets[m].normal.empty();
                any_tangent[m] = any_tangent[m] || !p.targets[m].tangent.empty();
            }
        for (size_t pi = 0; pi < mesh.primitives.size(); ++pi) {
            const PrimitiveAsset& p = mesh.primitives[pi];
            const auto& at = p.attributes;
            const size_t n = at.position.size();
            const uint32_t first = rig.vertex_count();
            const uint32_t local_first = first - sec.first_vertex;  // target rows and indices count from the section's first vertex
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
                // an un-skinned section's sets are not read by the library (skinrt.h): zeros
                const bool have = mr.skin >= 0 && k < at.joints.size() && k < at.weights.size();
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
                FlatTarget& dst = tg[m];
                if (src.position.size() != n || (!src.normal.empty() && src.normal.size() != n) ||
                    (!src.tangent.empty() && src.tangent.size() != n))
                    return fail(err, SKINRT_E_INVALID, "morph target length differs from position length");
                for (size_t v = 0; v < n; ++v) {
                    if (!dst.dense) dst.indices.push_back(local_first + (uint32_t)v);
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
            if (p.mode == PrimitiveAssetMode::TriangleList) {
                if (p.indices) {
                    const size_t whole = p.indices->size() / 3 * 3;
                    for (size_t c = 0; c < whole; ++c) {
                        if ((*p.indices)[c] >= n) return fail(err, SKINRT_E_RANGE, "primitive index beyond its positions (reference: index panic, tangent.rs:35)");
                        rig.triangles.push_back(first + (*p.indices)[c]);
                    }
                } else {
                    for (size_t c = 0; c < n / 3 * 3; ++c) rig.triangles.push_back(first + (uint32_t)c);
                }
            }
            rig.ranges.push_back({mr.node, (uint32_t)pi, first, (uint32_t)n, section_index});
        }
        sec.vertex_count = rig.vertex_count() - sec.first_vertex;
        // default weights (a13): node.weights if non-empty, else mesh.weights, else zeros
        const std::vector<float>& src = !mr.src->weights.empty() ? mr.src->weights : mesh.weights;
        rig.default_weights.resize(rig.targets.size(), 0.0f);
        std::copy_n(src.begin(), std::min(src.size(), M), rig.default_weights.begin() + sec.first_target);
        rig.sections.push_back(sec);
    }
    // weights channels: the C ABI reads key_count x M (x 3) floats, M = the targets of the section on the channel's node —
    // check the width here, where the lengths are known.  A channel aimed at a node whose mesh is not in the rig (or has
    // no targets) is dropped.
    for (auto& clip : rig.clips) {
        for (size_t k = 0; k < clip.size();) {
            FlatChannel& ch = clip[k];
            if (ch.path != SKINRT_PATH_WEIGHTS) { ++k; continue; }
            auto sec = std::find_if(rig.sections.begin(), rig.sections.end(),
                                    [&](const skinrt_section_desc& sc) { return sc.node == ch.node && sc.target_count; });
            if (sec == rig.sections.end()) { clip.erase(clip.begin() + (long)k); continue; }
            const size_t per_key = (ch.interpolation == SKINRT_INTERP_CUBICSPLINE ? 3 : 1) * (size_t)sec->target_count;
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
    if (skins.empty() && sections.empty()) {  // short form (one skin, one section)
        desc_.joint_count = (uint32_t)joint_node.size();
        desc_.joint_node = ptr(joint_node);
        desc_.inverse_bind = ptr(inverse_bind);
    } else {
        skin_descs_.assign(skins.size(), {});
        for (size_t k = 0; k < skins.size(); ++k) {
            skin_descs_[k].joint_count = (uint32_t)skins[k].joint_node.size();
            skin_descs_[k].joint_node = ptr(skins[k].joint_node);
            skin_descs_[k].inverse_bind = ptr(skins[k].inverse_bind);
        }
        desc_.skin_count = (uint32_t)skins.size();
        desc_.skins = ptr(skin_descs_);
        desc_.section_count = (uint32_t)sections.size();
        desc_.sections = ptr(sections);
    }
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
