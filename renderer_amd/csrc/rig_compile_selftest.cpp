// rig_compile_selftest.cpp — host-only checks of rig-compiler POLICIES that the C ABI does not report (test
// infrastructure: built by `make selftest`, run by tests/test_rig_compiler.py; links rig_compile.o directly).
//
//  * morph pairs: a (tile, target) pair touching fewer than kMorphDenseMin vertices is stored compacted, and ANY
//    compacted pair puts the whole rig on k_skin2's SPARSE instantiations — unless such pairs are a small minority
//    (at most an eighth of the pairs, at most 64 MB as dense blocks), in which case they are stored dense too
//    (CompiledRig::promoted_pairs).  The stored deltas must be the same either way.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rig_compile.hpp"

using namespace skinrt;

namespace {

int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

// One joint, `vertices` vertices bound to it, and compacted-form targets given as lists of touched vertices.
struct TinyRig {
    std::vector<int32_t> parent{-1};
    std::vector<float> t{0, 0, 0}, r{0, 0, 0, 1}, s{1, 1, 1};
    std::vector<uint32_t> joint_node{0};
    std::vector<float> pos, nrm, tan, w;
    std::vector<uint16_t> j;
    std::vector<std::vector<uint32_t>> touched;
    std::vector<std::vector<float>> delta;
    std::vector<skinrt_morph_target_desc> targets;
    const uint16_t* jp[1];
    const float* wp[1];
    skinrt_rig_desc desc{};

    TinyRig(uint32_t vertices, const std::vector<std::vector<uint32_t>>& lists) : touched(lists) {
        pos.assign((size_t)vertices * 3, 0.25f); nrm.assign((size_t)vertices * 3, 0.0f); tan.assign((size_t)vertices * 3, 0.0f);
        for (uint32_t v = 0; v < vertices; ++v) { nrm[v * 3 + 2] = 1.0f; tan[v * 3] = 1.0f; }
        j.assign((size_t)vertices * 4, 0); w.assign((size_t)vertices * 4, 0.0f);
        for (uint32_t v = 0; v < vertices; ++v) w[v * 4] = 1.0f;
        for (size_t m = 0; m < touched.size(); ++m) {
            delta.emplace_back(touched[m].size() * 3);
            for (size_t k = 0; k < touched[m].size(); ++k)
                for (int e = 0; e < 3; ++e) delta[m][k * 3 + e] = 0.5f + (float)m + 0.001f * (float)touched[m][k] + (float)e;
        }
        for (size_t m = 0; m < touched.size(); ++m)
            targets.push_back(skinrt_morph_target_desc{(uint32_t)touched[m].size(), touched[m].data(), delta[m].data(), nullptr, nullptr});
        jp[0] = j.data(); wp[0] = w.data();
        desc.struct_size = sizeof(desc);
        desc.node_count = 1; desc.parent = parent.data(); desc.translation = t.data(); desc.rotation = r.data(); desc.scale = s.data();
        desc.joint_count = 1; desc.joint_node = joint_node.data();
        desc.vertex_count = vertices; desc.position = pos.data(); desc.normal = nrm.data(); desc.tangent = tan.data(); desc.tangent_stride = 3;
        desc.set_count = 1; desc.joints = jp; desc.weights = wp;
        desc.target_count = (uint32_t)targets.size(); desc.targets = targets.data();
    }
};

std::vector<uint32_t> run(uint32_t first, uint32_t count) {
    std::vector<uint32_t> v(count);
    for (uint32_t k = 0; k < count; ++k) v[k] = first + k;
    return v;
}

// the x delta the compiled rig holds for (pair, tile lane), whatever the pair's layout
float stored_dx(const CompiledRig& r, const MorphPair& mp, uint32_t lane) {
    const float* blk = &r.mblock[mp.data_off];
    if (mp.dense) return blk[(lane / 64) * kMorphWaveFloats + 0 * 64 + (lane % 64)];
    const uint8_t slot = reinterpret_cast<const uint8_t*>(blk)[lane];
    return slot ? blk[kTileVerts / 4 + (size_t)(slot - 1) * 9] : 0.0f;
}

}  // namespace

int main() {
    std::string why;
    // 16 targets over whole tiles (dense pairs) + 1 target of 10 vertices: 1 of 17 pairs is small -> stored dense
    {
        std::vector<std::vector<uint32_t>> lists;
        for (uint32_t m = 0; m < 16; ++m) lists.push_back(run((m % 4) * kTileVerts, kTileVerts));
        lists.push_back(run(300, 10));
        TinyRig rig(4 * kTileVerts, lists);
        CompiledRig c;
        CHECK(compile_rig(&rig.desc, nullptr, c, why) == SKINRT_OK);
        CHECK(c.pairs.size() == 17);
        CHECK(c.promoted_pairs == 1);
        for (const MorphPair& mp : c.pairs) CHECK(mp.dense == 1);
        // the promoted pair's deltas: the 10 touched lanes of tile 1 carry their values, every other lane an exact zero
        const MorphPair* small = nullptr;
        for (const MorphPair& mp : c.pairs) if (mp.target == 16) small = &mp;
        CHECK(small != nullptr);
        if (small)
            for (uint32_t lane = 0; lane < kTileVerts; ++lane) {
                const uint32_t v = kTileVerts + lane;
                const float want = (v >= 300 && v < 310) ? rig.delta[16][(v - 300) * 3] : 0.0f;
                CHECK(stored_dx(c, *small, lane) == want);
            }
    }
    // 4 dense pairs + 4 small ones: half the pairs are small -> the compact layout stays
    {
        std::vector<std::vector<uint32_t>> lists;
        for (uint32_t m = 0; m < 4; ++m) lists.push_back(run(m * kTileVerts, kTileVerts));
        for (uint32_t m = 0; m < 4; ++m) lists.push_back(run(m * kTileVerts + 7, 12));
        TinyRig rig(4 * kTileVerts, lists);
        CompiledRig c;
        CHECK(compile_rig(&rig.desc, nullptr, c, why) == SKINRT_OK);
        CHECK(c.pairs.size() == 8);
        CHECK(c.promoted_pairs == 0);
        uint32_t compact = 0;
        for (const MorphPair& mp : c.pairs) {
            compact += mp.dense ? 0 : 1;
            if (!mp.dense) {
                const uint32_t tile = (uint32_t)(mp.target - 4);
                for (uint32_t lane = 0; lane < kTileVerts; ++lane) {
                    const float want = (lane >= 7 && lane < 19) ? rig.delta[mp.target][(lane - 7) * 3] : 0.0f;
                    CHECK(stored_dx(c, mp, lane) == want);
                }
                (void)tile;
            }
        }
        CHECK(compact == 4);
    }
    // a rig of small targets only: nothing to promote
    {
        std::vector<std::vector<uint32_t>> lists;
        for (uint32_t m = 0; m < 5; ++m) lists.push_back(run(40 * m, 8));
        TinyRig rig(kTileVerts, lists);
        CompiledRig c;
        CHECK(compile_rig(&rig.desc, nullptr, c, why) == SKINRT_OK);
        CHECK(c.promoted_pairs == 0);
        for (const MorphPair& mp : c.pairs) CHECK(mp.dense == 0);
    }
    // no targets at all
    {
        TinyRig rig(kTileVerts, {});
        CompiledRig c;
        CHECK(compile_rig(&rig.desc, nullptr, c, why) == SKINRT_OK);
        CHECK(c.promoted_pairs == 0 && c.pairs.empty());
    }
    // ring classes: 40 tiles, 34 of them on 4 joints, 6 on a window of 30 -> k_skin2 in two launches, cut between the two lengths;
    // all tiles alike -> one launch
    for (int mixed = 1; mixed >= 0; --mixed) {
        const uint32_t J = 40, tiles = 40, V = tiles * kTileVerts;
        std::vector<int32_t> parent(J, -1);
        std::vector<float> t(J * 3, 0.0f), rq(J * 4, 0.0f), sc(J * 3, 1.0f);
        for (uint32_t n = 0; n < J; ++n) rq[n * 4 + 3] = 1.0f;
        std::vector<uint32_t> jn(J);
        for (uint32_t n = 0; n < J; ++n) jn[n] = n;
        std::vector<float> pos(V * 3, 0.5f), nrm(V * 3, 0.0f), tan(V * 3, 0.0f), w(V * 4, 0.25f);
        std::vector<uint16_t> j(V * 4);
        for (uint32_t v = 0; v < V; ++v) {
            const uint32_t tile = v / kTileVerts;
            const bool lng = mixed && tile % 7 == 3;  // tiles 3, 10, 17, 24, 31, 38
            for (int a = 0; a < 4; ++a) j[v * 4 + a] = (uint16_t)(lng ? (v * 7 + a * 11) % 30 : (tile % 9) + a);
        }
        const uint16_t* jp[1] = {j.data()};
        const float* wp[1] = {w.data()};
        skinrt_rig_desc d{};
        d.struct_size = sizeof(d);
        d.node_count = J; d.parent = parent.data(); d.translation = t.data(); d.rotation = rq.data(); d.scale = sc.data();
        d.joint_count = J; d.joint_node = jn.data();
        d.vertex_count = V; d.position = pos.data(); d.normal = nrm.data(); d.tangent = tan.data(); d.tangent_stride = 3;
        d.set_count = 1; d.joints = jp; d.weights = wp;
        // (one morph target, so that the blocks carry morph slots: without them even 30-joint rings leave four blocks per CU
        // and the cut would not pay)
        std::vector<uint32_t> touched = run(0, kTileVerts);
        std::vector<float> dp(kTileVerts * 3, 0.125f);
        skinrt_morph_target_desc tg{kTileVerts, touched.data(), dp.data(), nullptr, nullptr};
        d.target_count = 1; d.targets = &tg;
        CompiledRig c;
        CHECK(compile_rig(&d, nullptr, c, why) == SKINRT_OK);
        if (mixed) {
            CHECK(c.max_tile_joints2 == 30);
            CHECK(c.split_joints >= 4 && c.split_joints < 30);
            CHECK(c.split_joints_short == 4);
            CHECK(c.tile_kind_count[2] == tiles);
        } else {
            CHECK(c.max_tile_joints2 == 4);
            CHECK(c.split_joints == 0);
        }
    }
    // every tile on a window of 50 joints (one or two weight sets): all of them the self-staging kernel's (kind 2, up to
    // kSkin2MaxJoints = 64 entries), none the loader-wave kernel's (kind 1); a window of 70: staged by nobody (kind 0)
    for (uint32_t window : {50u, 70u}) {
        const uint32_t J = 72, tiles = 8, V = tiles * kTileVerts;
        std::vector<int32_t> parent(J, -1);
        std::vector<float> t(J * 3, 0.0f), rq(J * 4, 0.0f), sc(J * 3, 1.0f);
        for (uint32_t n = 0; n < J; ++n) rq[n * 4 + 3] = 1.0f;
        std::vector<uint32_t> jn(J);
        for (uint32_t n = 0; n < J; ++n) jn[n] = n;
        std::vector<float> pos(V * 3, 0.5f), nrm(V * 3, 0.0f), tan(V * 3, 0.0f), w(V * 4, 0.25f);
        std::vector<uint16_t> j(V * 4);
        for (uint32_t v = 0; v < V; ++v)
            for (int a = 0; a < 4; ++a) j[v * 4 + a] = (uint16_t)((v * 3 + a * 11) % window);
        const uint16_t* jp[1] = {j.data()};
        const float* wp[1] = {w.data()};
        skinrt_rig_desc d{};
        d.struct_size = sizeof(d);
        d.node_count = J; d.parent = parent.data(); d.translation = t.data(); d.rotation = rq.data(); d.scale = sc.data();
        d.joint_count = J; d.joint_node = jn.data();
        d.vertex_count = V; d.position = pos.data(); d.normal = nrm.data(); d.tangent = tan.data(); d.tangent_stride = 3;
        d.set_count = 1; d.joints = jp; d.weights = wp;
        CompiledRig c;
        CHECK(compile_rig(&d, nullptr, c, why) == SKINRT_OK);
        CHECK(c.tile_kind_count[1] == 0);
        if (window <= kSkin2MaxJoints) {
            CHECK(c.max_tile_joints2 == window);
            CHECK(c.tile_kind_count[2] + c.tile_kind_count[3] == tiles && c.tile_kind_count[2] >= tiles - 1);
        } else {
            CHECK(c.tile_kind_count[0] == tiles);
        }
    }
    if (failures) { std::printf("rig compiler selftest: %d check(s) failed (%s)\n", failures, why.c_str()); return 1; }
    std::printf("rig compiler selftest ok\n");
    return 0;
}
