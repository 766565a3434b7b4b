// selftest.cpp — checks of the C++ host side.  Without arguments: host-only (flattening order,
// channel / target layout, error codes, skinrt_validate_rig) — no GPU needed.  With --gpu: also
// runs the scene through the library and compares with a direct recursive evaluation of the
// owned tree (double precision, written here for the test only).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>

#include <cstring>

#include "gltf_loader.hpp"
#include "skinned_scene.hpp"

using namespace skinrt_host;

static int failures = 0;
#define EXPECT(cond)                                                              \
    do {                                                                          \
        if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

static const BundleIndex kBundle = [] { BundleIndex b{}; b[0] = 0xAB; return b; }();
static AssetIndex node_id(size_t gltf_index) { return AssetIndex::type_index(kBundle, BundleAssetType::Node, gltf_index); }

static NodeAsset make_node(size_t gltf_index, Vec3 t, Quat q = {0, 0, 0, 1}, Vec3 s = {1, 1, 1}) {
    NodeAsset n;
    n.id = node_id(gltf_index);
    n.transform = NodeTransform(DecomposedTransform{t, q, s});
    return n;
}

// scene 0:  A(5) -> [ B(2) -> [ D(7) ],  C(0, mesh+skin) ],  E(1, matrix)      (glTF indices out of DFS order)
static void build(std::vector<SceneAsset>& scenes, std::vector<AnimationAsset>& anims) {
    auto skin = std::make_shared<SkinAsset>();
    skin->id = AssetIndex::type_index(kBundle, BundleAssetType::Skin, 0);
    skin->joint_ids = {node_id(7), node_id(0), node_id(5)};
    // inverse bind: joints bound at their rest world position (translations only)
    auto translate = [](float x, float y, float z) { Mat4 m = mat4_identity(); m[12] = x; m[13] = y; m[14] = z; return m; };
    skin->inverse_bind_matrices = {translate(-1, -3, 0), translate(-1, 0, -4), translate(-1, 0, 0)};

    NodeAsset D = make_node(7, {0, 1, 0});
    NodeAsset B = make_node(2, {0, 2, 0}, {0, 0, 1, 0});  // half turn about Z (exact in fp32)
    B.children.push_back(D);
    NodeAsset C = make_node(0, {0, 0, 4});
    C.skin = skin;
    MeshAsset mesh;
    mesh.weights = {0.5f, 0.25f};
    PrimitiveAsset p0, p1;
    p0.attributes.position = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    p0.attributes.normal = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    p0.attributes.tangent = {{0, 1, 0, 1}, {0, 0, 1, 1}, {1, 0, 0, 1}};
    p0.attributes.joints = {{{0, 1, 0, 0}, {1, 0, 0, 0}, {2, 0, 0, 0}}};
    p0.attributes.weights = {{{0.5f, 0.5f, 0, 0}, {1, 0, 0, 0}, {1, 0, 0, 0}}};
    p0.targets.resize(1);
    p0.targets[0].position = {{1, 1, 1}, {0, 0, 0}, {2, 0, 0}};
    p1.attributes.position = {{2, 2, 2}, {3, 3, 3}};
    p1.attributes.normal = {{0, 0, 1}, {0, 0, 1}};
    p1.attributes.tangent = {{1, 0, 0, 1}, {1, 0, 0, 1}};
    p1.attributes.joints = {{{2, 0, 0, 0}, {0, 0, 0, 0}}};
    p1.attributes.weights = {{{1, 0, 0, 0}, {1, 0, 0, 0}}};
    p1.targets.resize(2);
    p1.targets[0].position = {{0, 0, 0}, {0, 4, 0}};
    p1.targets[1].position = {{8, 0, 0}, {0, 0, 0}};
    p1.targets[1].normal = {{0, 1, 0}, {0, 0, 0}};
    mesh.primitives = {p0, p1};
    C.mesh = mesh;
    NodeAsset A = make_node(5, {1, 0, 0});
    A.children = {B, C};
    NodeAsset E;
    E.id = node_id(1);
    Mat4 em = mat4_identity(); em[0] = 2; em[5] = 2; em[10] = 2; em[12] = 9;
    E.transform = NodeTransform(MatrixNodeTransform{em});
    SceneAsset s0;
    s0.nodes = {A, E};
    scenes = {s0};

    AnimationAsset walk;
    {
        AnimationChannelAsset c;  // D translates linearly from (0,1,0) to (0,3,0) over 2 s
        c.sampler.kind = AnimationSampler::Kind::Translation;
        c.sampler.vec3.kind = AnimationKeyFrames<std::array<float, 3>>::Kind::Linear;
        c.sampler.vec3.frames = {{0.0f, {0, 1, 0}}, {2.0f, {0, 3, 0}}};
        c.length = 2.0f;
        c.target_id = node_id(7);
        walk.channels.push_back(c);
    }
    {
        AnimationChannelAsset c;  // B: cubic-spline rotation, 2 keys
        c.sampler.kind = AnimationSampler::Kind::Rotation;
        c.sampler.rotation.kind = AnimationKeyFrames<std::array<float, 4>>::Kind::CubicSpline;
        c.sampler.rotation.cubic_frames = {{0.0f, {{9, 9, 9, 9}, {0, 0, 1, 0}, {0, 0, 0, 0}}},
                                           {1.0f, {{0, 0, 0, 0}, {0, 0, 1, 0}, {7, 7, 7, 7}}}};
        c.length = 1.0f;
        c.target_id = node_id(2);
        walk.channels.push_back(c);
    }
    anims = {walk};
}

static void host_checks() {
    std::vector<SceneAsset> scenes;
    std::vector<AnimationAsset> anims;
    build(scenes, anims);
    FlatRig rig;
    std::string err;
    EXPECT(flatten(scenes, anims, {}, rig, err) == SKINRT_OK);
    // node ordering: DFS pre-order A B D C E
    const size_t want_ids[5] = {5, 2, 7, 0, 1};
    const int32_t want_parent[5] = {-1, 0, 1, 0, -1};
    EXPECT(rig.node_count() == 5);
    for (int i = 0; i < 5; ++i) {
        EXPECT(rig.node_id[i].index == want_ids[i]);
        EXPECT(rig.parent[i] == want_parent[i]);
    }
    EXPECT(rig.has_matrix.size() == 5 && rig.has_matrix[4] == 1 && rig.has_matrix[0] == 0);
    EXPECT(rig.matrix[4 * 16 + 12] == 9.0f);
    EXPECT(rig.rotation[1 * 4 + 2] == 1.0f && rig.rotation[1 * 4 + 3] == 0.0f);
    // joints resolve through the id map, not by glTF index
    EXPECT(rig.skins.size() == 1 && rig.sections.size() == 1 && rig.sections[0].skin == 0 && rig.sections[0].node == 3);
    const FlatSkin& sk0 = rig.skins[0];
    EXPECT(sk0.joint_node.size() == 3 && sk0.joint_node[0] == 2 && sk0.joint_node[1] == 3 && sk0.joint_node[2] == 0);
    EXPECT(sk0.inverse_bind.size() == 48 && sk0.inverse_bind[13] == -3.0f);
    // clips
    EXPECT(rig.clips.size() == 1 && rig.clips[0].size() == 2 && rig.clip_length[0] == 2.0f);
    EXPECT(rig.clips[0][0].node == 2 && rig.clips[0][0].path == SKINRT_PATH_TRANSLATION && rig.clips[0][0].interpolation == SKINRT_INTERP_LINEAR);
    EXPECT(rig.clips[0][1].node == 1 && rig.clips[0][1].interpolation == SKINRT_INTERP_CUBICSPLINE);
    EXPECT(rig.clips[0][1].values.size() == 2 * 3 * 4 && rig.clips[0][1].values[0] == 9.0f && rig.clips[0][1].values[4 + 2] == 1.0f);
    // mesh: two primitives concatenated, targets become indexed because they differ per primitive
    EXPECT(rig.vertex_count() == 5 && rig.ranges.size() == 2 && rig.ranges[1].first_vertex == 3 && rig.ranges[1].node == 3);
    EXPECT(rig.joints.size() == 1 && rig.joints[0][4 * 4] == 0 && rig.joints[0][3 * 4] == 2);
    EXPECT(rig.targets.size() == 2 && !rig.targets[0].dense && rig.targets[0].indices.size() == 5);
    EXPECT(rig.targets[1].indices.size() == 2 && rig.targets[1].indices[0] == 3 && rig.targets[1].normal.size() == 6);
    EXPECT(rig.default_weights.size() == 2 && rig.default_weights[1] == 0.25f);
    // node.weights override mesh.weights
    {
        auto s2 = scenes;
        s2[0].nodes[0].children[1].weights = {1.0f};
        FlatRig r2;
        EXPECT(flatten(s2, anims, {}, r2, err) == SKINRT_OK);
        EXPECT(r2.default_weights[0] == 1.0f && r2.default_weights[1] == 0.0f);
    }
    // the library accepts the descriptor (host-side compile, no device)
    skinrt_rig_info info{};
    const int rc = skinrt_validate_rig(nullptr, rig.desc(), &info);
    if (rc != SKINRT_OK) std::printf("validate: %s\n", skinrt_last_error());
    EXPECT(rc == SKINRT_OK);
    EXPECT(info.node_count == 5 && info.joint_count == 3 && info.vertex_count == 5 && info.target_count == 2 && info.clip_count == 1);
    // errors instead of the reference's panics
    {
        auto a2 = anims;
        a2[0].channels[0].target_id = node_id(99);
        FlatRig r2;
        EXPECT(flatten(scenes, a2, {}, r2, err) == SKINRT_E_RANGE);
        EXPECT(err.find("unknown node") != std::string::npos);
    }
    {
        auto s2 = scenes;
        s2[0].nodes[0].children[1].mesh->primitives[0].attributes.normal.pop_back();
        FlatRig r2;
        EXPECT(flatten(s2, anims, {}, r2, err) == SKINRT_E_INVALID);
    }
    {
        FlattenOptions o;
        o.scene = 3;
        FlatRig r2;
        EXPECT(flatten(scenes, anims, o, r2, err) == SKINRT_E_RANGE);
    }
}

// scene of build() + a second skin on the same node tree (node F(4) under B, joints D and A, its own morph target and a
// weights channel) + an un-skinned mesh on the Matrix node E: what an avatar with several skins and a prop looks like
static void build_multi(std::vector<SceneAsset>& scenes, std::vector<AnimationAsset>& anims) {
    build(scenes, anims);
    auto skin1 = std::make_shared<SkinAsset>();
    skin1->id = AssetIndex::type_index(kBundle, BundleAssetType::Skin, 1);
    skin1->joint_ids = {node_id(5), node_id(7)};  // no inverse bind matrices: identity (loader/gltf/mod.rs:1139-1145)
    NodeAsset F = make_node(4, {0, 0, 0});
    F.skin = skin1;
    MeshAsset face;
    face.weights = {0.75f};
    PrimitiveAsset fp;
    fp.attributes.position = {{1, 1, 1}, {2, 0, 0}};
    fp.attributes.normal = {{0, 1, 0}, {1, 0, 0}};
    fp.attributes.tangent = {{1, 0, 0, 1}, {0, 1, 0, 1}};
    fp.attributes.joints = {{{1, 0, 0, 0}, {0, 1, 0, 0}}};
    fp.attributes.weights = {{{1, 0, 0, 0}, {0.25f, 0.75f, 0, 0}}};
    fp.targets.resize(1);
    fp.targets[0].position = {{0, 0, 4}, {0, 0, 0}};
    face.primitives = {fp};
    F.mesh = face;
    NodeAsset& A = scenes[0].nodes[0];
    A.children[0].children.push_back(F);  // A -> B -> [D, F]
    NodeAsset& E = scenes[0].nodes[1];
    MeshAsset prop;
    PrimitiveAsset pp;
    pp.attributes.position = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    pp.attributes.normal = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    pp.attributes.tangent = {{0, 1, 0, 1}, {0, 0, 1, 1}, {1, 0, 0, 1}};
    prop.primitives = {pp};
    E.mesh = prop;
    {
        AnimationChannelAsset c;  // F's mesh: its one morph weight goes 0 -> 1 over 2 s
        c.sampler.kind = AnimationSampler::Kind::MorphTargetWeights;
        c.sampler.weights.kind = AnimationKeyFrames<std::vector<float>>::Kind::Linear;
        c.sampler.weights.frames = {{0.0f, {0.0f}}, {2.0f, {1.0f}}};
        c.length = 2.0f;
        c.target_id = node_id(4);
        anims[0].channels.push_back(c);
    }
}

static void host_checks_multi() {
    std::vector<SceneAsset> scenes;
    std::vector<AnimationAsset> anims;
    build_multi(scenes, anims);
    FlatRig rig;
    std::string err;
    EXPECT(flatten(scenes, anims, {}, rig, err) == SKINRT_OK);
    // DFS pre-order A B D F C E; sections in that order: F (skin 1 is met first), C, E
    EXPECT(rig.node_count() == 6 && rig.node_id[3].index == 4 && rig.node_id[4].index == 0);
    EXPECT(rig.skins.size() == 2 && rig.skins[0].id.index == 1 && rig.skins[1].id.index == 0);
    EXPECT(rig.skins[0].joint_node.size() == 2 && rig.skins[0].joint_node[0] == 0 && rig.skins[0].joint_node[1] == 2 && rig.skins[0].inverse_bind.empty());
    EXPECT(rig.sections.size() == 3);
    if (rig.sections.size() == 3) {
        EXPECT(rig.sections[0].node == 3 && rig.sections[0].skin == 0 && rig.sections[0].first_vertex == 0 && rig.sections[0].vertex_count == 2);
        EXPECT(rig.sections[0].first_target == 0 && rig.sections[0].target_count == 1);
        EXPECT(rig.sections[1].node == 4 && rig.sections[1].skin == 1 && rig.sections[1].first_vertex == 2 && rig.sections[1].vertex_count == 5);
        EXPECT(rig.sections[1].first_target == 1 && rig.sections[1].target_count == 2);
        EXPECT(rig.sections[2].node == 5 && rig.sections[2].skin == -1 && rig.sections[2].first_vertex == 7 && rig.sections[2].vertex_count == 3);
        EXPECT(rig.sections[2].target_count == 0);
    }
    EXPECT(rig.vertex_count() == 10 && rig.targets.size() == 3 && rig.targets[0].dense && !rig.targets[1].dense);
    EXPECT(rig.targets[2].indices.size() == 2 && rig.targets[2].indices[0] == 3);  // counted from the section's first vertex
    EXPECT(rig.default_weights.size() == 3 && rig.default_weights[0] == 0.75f && rig.default_weights[2] == 0.25f);
    EXPECT(rig.clips[0].size() == 3 && rig.clips[0][2].path == SKINRT_PATH_WEIGHTS && rig.clips[0][2].node == 3);
    skinrt_rig_info info{};
    const int rc = skinrt_validate_rig(nullptr, rig.desc(), &info);
    if (rc != SKINRT_OK) std::printf("validate: %s\n", skinrt_last_error());
    EXPECT(rc == SKINRT_OK);
    EXPECT(info.skin_count == 2 && info.section_count == 3 && info.joint_count == 5 && info.palette_count == 6 && info.vertex_count == 10);
    EXPECT(info.weight_channel_clips == 1);
    // the one-skin view of the same scenes: only that skin's meshes
    FlattenOptions o;
    const AssetIndex want = AssetIndex::type_index(kBundle, BundleAssetType::Skin, 0);
    o.skin = &want;
    FlatRig r1;
    EXPECT(flatten(scenes, anims, o, r1, err) == SKINRT_OK);
    EXPECT(r1.skins.size() == 1 && r1.sections.size() == 1 && r1.vertex_count() == 5 && r1.clips[0].size() == 2);
    // un-skinned meshes can be left out
    FlattenOptions o2;
    o2.include_unskinned = false;
    FlatRig r2;
    EXPECT(flatten(scenes, anims, o2, r2, err) == SKINRT_OK);
    EXPECT(r2.sections.size() == 2 && r2.vertex_count() == 7);
}

// ---- direct evaluation of the owned tree (test-only) -------------------------------------------
struct M4 { double m[16]; };
static M4 mul(const M4& a, const M4& b) {
    M4 r{};
    for (int c = 0; c < 4; ++c)
        for (int rr = 0; rr < 4; ++rr)
            for (int k = 0; k < 4; ++k) r.m[c * 4 + rr] += a.m[k * 4 + rr] * b.m[c * 4 + k];
    return r;
}
static M4 trs(const Vec3& t, const Quat& q, const Vec3& s) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    M4 r{};
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y),
                         2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x),
                         2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)};
    for (int c = 0; c < 3; ++c)
        for (int rr = 0; rr < 3; ++rr) r.m[c * 4 + rr] = R[c * 3 + rr] * s[c];
    r.m[12] = t[0]; r.m[13] = t[1]; r.m[14] = t[2]; r.m[15] = 1;
    return r;
}

static void gpu_checks() {
    std::vector<SceneAsset> scenes;
    std::vector<AnimationAsset> anims;
    build(scenes, anims);
    SkinnedScene scene(scenes, anims, 2);
    const float tau[2] = {0.0f, 1.0f};
    const uint32_t clip[2] = {0, 0};
    const float mw[4] = {0, 0, 1.0f, 0.5f};  // instance 0: no morph; instance 1: both targets
    scene.set_clocks(tau, clip);
    scene.set_morph_weights(mw);
    scene.render();
    scene.synchronize();
    const auto world = scene.download<float>(skinrt_get_world);        // [2][5][12] rows
    const auto pos = scene.download<float>(skinrt_get_positions);      // [2][5][3]
    const auto keys = scene.download<uint32_t>(skinrt_get_keyframe_index);
    EXPECT(world.size() == 2 * 5 * 12 && pos.size() == 2 * 5 * 3);
    for (int i = 0; i < 2; ++i) {
        // expected worlds by recursion over the tree, with D's channel value substituted
        std::vector<M4> w;
        std::function<void(const NodeAsset&, const M4*)> rec = [&](const NodeAsset& n, const M4* parent) {
            M4 local;
            if (const auto* d = std::get_if<DecomposedTransform>(&*n.transform)) {
                Vec3 t = d->translation;
                if (n.id.index == 7) t = {0, 1.0f + tau[i], 0};  // linear 1 -> 3 over 2 s
                local = trs(t, d->rotation, d->scale);              // B's cubic channel holds (0,0,1,0) at both keys with zero tangents inside
            } else {
                for (int k = 0; k < 16; ++k) local.m[k] = std::get<MatrixNodeTransform>(*n.transform).m[k];
            }
            const M4 me = parent ? mul(*parent, local) : local;
            w.push_back(me);
            const size_t at = w.size() - 1;
            for (const auto& c : n.children) { const M4 keep = w[at]; rec(c, &keep); }
        };
        for (const auto& n : scenes[0].nodes) rec(n, nullptr);
        for (int n = 0; n < 5; ++n)
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 4; ++c) {
                    const double want = w[n].m[c * 4 + r];
                    const double got = world[(i * 5 + n) * 12 + r * 4 + c];
                    if (std::fabs(got - want) > 1e-6 + 1e-5 * std::fabs(want)) {
                        std::printf("world[%d][%d] row %d col %d: got %g want %g\n", i, n, r, c, got, want);
                        ++failures;
                    }
                }
    }
    EXPECT(keys[0] == 0 && keys[2 + 0] == 0);  // tau = 1.0 is inside [0, 2): key 0 of D's channel
    EXPECT(keys[2 + 1] == 1);                  // tau = 1.0 >= last key of B's channel: clamped to key 1
    // vertex 2 (joint 2 = node A, weight 1): joint = world(A) * ibm = identity -> position unchanged; morph adds (2,0,0) * w0
    EXPECT(std::fabs(pos[(0 * 5 + 2) * 3 + 0] - 0.0f) < 1e-6 && std::fabs(pos[(0 * 5 + 2) * 3 + 2] - 1.0f) < 1e-6);
    EXPECT(std::fabs(pos[(1 * 5 + 2) * 3 + 0] - 2.0f) < 1e-6);
    // vertex 3 (second primitive, joint 2): target 1 adds (8,0,0) * 0.5
    EXPECT(std::fabs(pos[(1 * 5 + 3) * 3 + 0] - (2.0f + 4.0f)) < 1e-6);
    // reference-layout exports
    const auto uni = scene.skin_uniform(1);
    EXPECT(uni.size() == MAX_JOINTS && uni[2].transform[0][0] == 1.0f && uni[2].transform[3][0] == 0.0f && uni[2].normal[1][1] == 1.0f);
    EXPECT(uni[3].transform[0][0] == 0.0f && uni[127].normal[2][2] == 0.0f);
    const auto verts = scene.texture_skin_vertices(1);
    EXPECT(verts.size() == 5 && verts[3].joint_index[0] == 2 && verts[3].joint_weight[0] == 1.0f);
    EXPECT(std::fabs(verts[3].position[0] - 6.0f) < 1e-6);
    // per-frame normals from the deformed positions: the flattener kept the scene's triangle list (primitive 0: three
    // vertices without indices = one triangle; primitive 1 has two vertices: none), the device runs calculate_tangent per
    // instance on the skinned positions — bit for bit the host restatement on the same positions
    EXPECT(scene.rig().triangles.size() == 3 && scene.rig().triangles[2] == 2);
    scene.recompute_normals(0, 2);
    scene.synchronize();
    const auto nrm = scene.download<float>(skinrt_get_normals);
    for (int i = 0; i < 2; ++i) {
        std::vector<std::array<float, 3>> p(5), want;
        for (int v = 0; v < 5; ++v) p[v] = {pos[(i * 5 + v) * 3], pos[(i * 5 + v) * 3 + 1], pos[(i * 5 + v) * 3 + 2]};
        const std::vector<uint32_t> tri = scene.rig().triangles;
        EXPECT(calculate_tangent(PrimitiveAssetMode::TriangleList, p, &tri, want));
        for (int v = 0; v < 3; ++v) EXPECT(std::memcmp(&nrm[(i * 5 + v) * 3], want[v].data(), 12) == 0 && std::isfinite(want[v][0]));
        for (int v = 3; v < 5; ++v) EXPECT(std::isnan(nrm[(i * 5 + v) * 3]) && std::isnan(want[v][0]));
    }
}

static void gpu_checks_multi() {
    std::vector<SceneAsset> scenes;
    std::vector<AnimationAsset> anims;
    build_multi(scenes, anims);
    SkinnedScene scene(scenes, anims, 2);
    const float tau[2] = {0.0f, 1.0f};
    const uint32_t clip[2] = {0, 0};
    scene.set_clocks(tau, clip);
    scene.render();
    scene.synchronize();
    const auto pos = scene.download<float>(skinrt_get_positions);  // [2][10][3]
    EXPECT(pos.size() == 2 * 10 * 3);
    // section F, vertex 0: joint 1 of skin 1 = node D, no inverse bind: world(D) * p.  D's world at tau: A(1,0,0) * B(0,2,0; half turn
    // about Z) * D(0,1+tau,0) -> origin (1, 2 - (1+tau), 0), axes x,y flipped.  p = (1,1,1) + w * (0,0,4), w = tau / 2 from the channel
    for (int i = 0; i < 2; ++i) {
        const float w = tau[i] * 0.5f, ty = 2.0f - (1.0f + tau[i]);
        EXPECT(std::fabs(pos[(i * 10 + 0) * 3 + 0] - (1.0f - 1.0f)) < 1e-6);
        EXPECT(std::fabs(pos[(i * 10 + 0) * 3 + 1] - (ty - 1.0f)) < 1e-6);
        EXPECT(std::fabs(pos[(i * 10 + 0) * 3 + 2] - (1.0f + 4.0f * w)) < 1e-6);
    }
    // the prop on E: world(E) = diag(2) + translation (9,0,0): vertex (0,1,0) -> (9,2,0); normal matrix = diag(0.5)
    const auto nrm = scene.download<float>(skinrt_get_normals);
    EXPECT(std::fabs(pos[(1 * 10 + 8) * 3 + 0] - 9.0f) < 1e-6 && std::fabs(pos[(1 * 10 + 8) * 3 + 1] - 2.0f) < 1e-6);
    EXPECT(std::fabs(nrm[(1 * 10 + 8) * 3 + 1] - 0.5f) < 1e-6);
    const InstanceUniform iu = scene.instance_uniform(1, 5);
    EXPECT(iu.transform[0][0] == 2.0f && iu.transform[3][0] == 9.0f && iu.transform[3][3] == 1.0f && iu.normal[1][1] == 0.5f && iu.normal[1][3] == 0.0f);
    // per-skin palettes: skin 0 (the face's) has 2 joints, skin 1 (the body's) 3
    const auto u0 = scene.skin_uniform(1, 0), u1 = scene.skin_uniform(1, 1);
    EXPECT(u0[1].transform[3][1] == 0.0f - 0.0f + (2.0f - 2.0f) && u0[2].transform[0][0] == 0.0f && u1[2].transform[0][0] == 1.0f && u1[3].transform[0][0] == 0.0f);
    const auto fv = scene.texture_skin_vertices_of(0, 1);
    EXPECT(fv.size() == 2 && fv[1].joint_index[1] == 1 && fv[1].joint_weight[1] == 0.75f);
    EXPECT(std::fabs(fv[0].position[2] - 3.0f) < 1e-6);
}

int main(int argc, char** argv) {
    host_checks();
    host_checks_multi();
    if (argc > 1 && std::string(argv[1]) == "--gpu") {
        try {
            gpu_checks();
            gpu_checks_multi();
        } catch (const Error& e) {
            std::printf("FAIL %s\n", e.what());
            ++failures;
        }
    }
    std::printf(failures ? "host selftest: %d failure(s)\n" : "host selftest ok\n", failures);
    return failures ? 1 : 0;
}
