// selftest.cpp — checks of the C++ host side.  Without arguments: host-only (flattening order,
// channel / target layout, error codes, skinrt_validate_rig) — no GPU needed.  With --gpu: also
// runs the scene through the library and compares with a direct recursive evaluation of the
// owned tree (double precision, written here for the test only).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>

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
    EXPECT(rig.joint_node.size() == 3 && rig.joint_node[0] == 2 && rig.joint_node[1] == 3 && rig.joint_node[2] == 0);
    EXPECT(rig.inverse_bind.size() == 48 && rig.inverse_bind[13] == -3.0f);
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
}

int main(int argc, char** argv) {
    host_checks();
    if (argc > 1 && std::string(argv[1]) == "--gpu") {
        try {
            gpu_checks();
        } catch (const Error& e) {
            std::printf("FAIL %s\n", e.what());
            ++failures;
        }
    }
    std::printf(failures ? "host selftest: %d failure(s)\n" : "host selftest ok\n", failures);
    return failures ? 1 : 0;
}
