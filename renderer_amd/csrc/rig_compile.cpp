// rig_compile.cpp — validate the flat loader output and build the kernel tables.
// Host-only; compiled with -ffp-contract=off so the one piece of arithmetic here
// (Matrix -> TRS decomposition) rounds like glam's scalar code.
#include "rig_compile.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace skinrt {

namespace {

struct Fail {
    int code;
    std::string msg;
};

#define REQUIRE(cond, code, text)            \
    do {                                     \
        if (!(cond)) throw Fail{code, text}; \
    } while (0)

bool finite_all(const float* p, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (!std::isfinite(p[i])) return false;
    return true;
}

bool affine_last_row(const float* m) {  // column-major: row 3 = m[3], m[7], m[11], m[15]
    return m[3] == 0.0f && m[7] == 0.0f && m[11] == 0.0f && m[15] == 1.0f;
}

void rows_from_cols(const float* m16, float* r12) {  // first three rows, row-major
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) r12[r * 4 + c] = m16[c * 4 + r];
}

float det4(const float* m) {  // glam Mat4::determinant, scalar path
    float m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3];
    float m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
    float m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11];
    float m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];
    float a2323 = m22 * m33 - m23 * m32;
    float a1323 = m21 * m33 - m23 * m31;
    float a1223 = m21 * m32 - m22 * m31;
    float a0323 = m20 * m33 - m23 * m30;
    float a0223 = m20 * m32 - m22 * m30;
    float a0123 = m20 * m31 - m21 * m30;
    return m00 * (m11 * a2323 - m12 * a1323 + m13 * a1223) - m01 * (m10 * a2323 - m12 * a0323 + m13 * a0223) +
           m02 * (m10 * a1323 - m11 * a0323 + m13 * a0123) - m03 * (m10 * a1223 - m11 * a0223 + m12 * a0123);
}

float len3(const float* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

}  // namespace

// glam Mat4::to_scale_rotation_translation + Quat::from_rotation_axes (Mike Day's method),
// the conversion behind `impl From<NodeTransform> for DecomposedTransform` (node.rs:73-87).
void decompose_affine(const float* m, float* t, float* q, float* s) {
    float det = det4(m);
    float sg = std::signbit(det) ? -1.0f : 1.0f;
    s[0] = len3(m + 0) * sg;
    s[1] = len3(m + 4);
    s[2] = len3(m + 8);
    float is0 = 1.0f / s[0], is1 = 1.0f / s[1], is2 = 1.0f / s[2];
    float m00 = m[0] * is0, m01 = m[1] * is0, m02 = m[2] * is0;
    float m10 = m[4] * is1, m11 = m[5] * is1, m12 = m[6] * is1;
    float m20 = m[8] * is2, m21 = m[9] * is2, m22 = m[10] * is2;
    if (m22 <= 0.0f) {
        float dif10 = m11 - m00, omm22 = 1.0f - m22;
        if (dif10 <= 0.0f) {
            float f = omm22 - dif10, i4 = 0.5f / std::sqrt(f);
            q[0] = f * i4; q[1] = (m01 + m10) * i4; q[2] = (m02 + m20) * i4; q[3] = (m12 - m21) * i4;
        } else {
            float f = omm22 + dif10, i4 = 0.5f / std::sqrt(f);
            q[0] = (m01 + m10) * i4; q[1] = f * i4; q[2] = (m12 + m21) * i4; q[3] = (m20 - m02) * i4;
        }
    } else {
        float sum10 = m11 + m00, opm22 = 1.0f + m22;
        if (sum10 <= 0.0f) {
            float f = opm22 - sum10, i4 = 0.5f / std::sqrt(f);
            q[0] = (m02 + m20) * i4; q[1] = (m12 + m21) * i4; q[2] = f * i4; q[3] = (m01 - m10) * i4;
        } else {
            float f = opm22 + sum10, i4 = 0.5f / std::sqrt(f);
            q[0] = (m12 - m21) * i4; q[1] = (m20 - m02) * i4; q[2] = (m01 - m10) * i4; q[3] = f * i4;
        }
    }
    t[0] = m[12]; t[1] = m[13]; t[2] = m[14];
}

static void compile_impl(const skinrt_rig_desc* d, const skinrt_config* cfg, CompiledRig& r) {
    REQUIRE(d != nullptr, SKINRT_E_INVALID, "rig desc is NULL");
    REQUIRE(d->struct_size == sizeof(skinrt_rig_desc), SKINRT_E_INVALID, "skinrt_rig_desc.struct_size mismatch");
    const uint32_t N = d->node_count, V = d->vertex_count, K = d->set_count, M = d->target_count;
    r.N = N; r.V = V; r.K = K; r.M = M; r.clips = d->clip_count;
    REQUIRE(N >= 1, SKINRT_E_INVALID, "rig has no nodes");
    REQUIRE(d->parent && d->translation && d->rotation && d->scale, SKINRT_E_INVALID, "node arrays missing");
    REQUIRE(M <= 65535, SKINRT_E_INVALID, "more than 65535 morph targets");

    // ---- skins and sections (ABI 6): resolve the short forms ------------------------------------------
    std::vector<skinrt_skin_desc> skins;
    if (d->skin_count) {
        REQUIRE(d->skins != nullptr, SKINRT_E_INVALID, "skins is NULL");
        skins.assign(d->skins, d->skins + d->skin_count);
    } else if (d->joint_count) {
        skins.push_back(skinrt_skin_desc{d->joint_count, d->joint_node, d->inverse_bind});
    }
    const bool one_section = d->section_count == 0;  // short form: a weights channel's target_node is not looked at
    std::vector<skinrt_section_desc> secs;
    if (!one_section) {
        REQUIRE(d->sections != nullptr, SKINRT_E_INVALID, "sections is NULL");
        secs.assign(d->sections, d->sections + d->section_count);
    } else if (V || M) {
        if (V) REQUIRE(!skins.empty(), SKINRT_E_INVALID, "skinned vertices but no joints");
        secs.push_back(skinrt_section_desc{0u, skins.empty() ? -1 : 0, 0u, V, 0u, M});
    }
    uint32_t sum_joints = 0;
    for (const skinrt_skin_desc& sk : skins) {
        REQUIRE(sk.joint_count == 0 || sk.joint_node, SKINRT_E_INVALID, "joint_node is NULL");
        if (cfg && cfg->reference_limits)
            REQUIRE(sk.joint_count <= SKINRT_MAX_JOINTS_REFERENCE, SKINRT_E_LIMIT, "reference_limits: more than 128 joints (uniform/skin.rs:10)");
        r.skins.push_back(SkinSpan{sum_joints, sk.joint_count});
        REQUIRE(sk.joint_count <= 65536 && sum_joints + sk.joint_count <= 65536, SKINRT_E_INVALID, "joint indices are u16: at most 65536 palette entries");
        sum_joints += sk.joint_count;
    }
    if (cfg && cfg->reference_limits) REQUIRE(K <= 1, SKINRT_E_LIMIT, "reference_limits: the reference shader reads joint/weight set 0 only");
    {   // sections tile the vertices and the targets in order; un-skinned ones get a palette entry per distinct node
        uint32_t v_next = 0, m_next = 0;
        std::vector<uint32_t> rigid_nodes;
        for (const skinrt_section_desc& sc : secs) {
            REQUIRE(sc.node < N, SKINRT_E_RANGE, "section node out of range");
            REQUIRE(sc.skin >= -1 && sc.skin < (int32_t)skins.size(), SKINRT_E_RANGE, "section skin out of range");
            REQUIRE(sc.first_vertex == v_next && sc.vertex_count <= V - v_next, SKINRT_E_INVALID, "sections must tile the vertices in order without gaps");
            REQUIRE(sc.first_target == m_next && sc.target_count <= M - m_next, SKINRT_E_INVALID, "sections must tile the morph targets in order without gaps");
            v_next += sc.vertex_count;
            m_next += sc.target_count;
            skinrt_section_layout lay{};
            lay.node = sc.node; lay.skin = sc.skin;
            lay.first_vertex = sc.first_vertex; lay.vertex_count = sc.vertex_count;
            lay.first_target = sc.first_target; lay.target_count = sc.target_count;
            if (sc.skin >= 0) lay.first_palette_entry = r.skins[sc.skin].first;
            else if (sc.vertex_count) {
                size_t k = std::find(rigid_nodes.begin(), rigid_nodes.end(), sc.node) - rigid_nodes.begin();
                if (k == rigid_nodes.size()) rigid_nodes.push_back(sc.node);
                REQUIRE(sum_joints + rigid_nodes.size() <= 65536, SKINRT_E_INVALID, "joint indices are u16: at most 65536 palette entries");
                lay.first_palette_entry = sum_joints + (uint32_t)k;
            }
            r.sections.push_back(lay);
        }
        REQUIRE(v_next == V, SKINRT_E_INVALID, "sections do not cover every vertex");
        REQUIRE(m_next == M, SKINRT_E_INVALID, "sections do not cover every morph target");
        // ---- palette: joint -> node, inverse bind rows -----------------------------------------------
        const uint32_t J = sum_joints + (uint32_t)rigid_nodes.size();
        r.J = J;
        r.joint_node.resize(J);
        r.ibm.resize((size_t)J * 12);
        static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        uint32_t j = 0;
        for (const skinrt_skin_desc& sk : skins)
            for (uint32_t q = 0; q < sk.joint_count; ++q, ++j) {
                r.joint_node[j] = sk.joint_node[q];
                REQUIRE(r.joint_node[j] < N, SKINRT_E_RANGE, "joint_node out of range");
                const float* m = sk.inverse_bind ? sk.inverse_bind + (size_t)q * 16 : ident;
                REQUIRE(finite_all(m, 16), SKINRT_E_INVALID, "non-finite inverse bind matrix");
                REQUIRE(affine_last_row(m), SKINRT_E_NONAFFINE, "inverse bind matrix last row is not 0,0,0,1");
                rows_from_cols(m, &r.ibm[(size_t)j * 12]);
                for (int e = 12; e < 15; ++e) r.bind_extent = std::max(r.bind_extent, std::fabs(m[e]));
            }
        for (uint32_t n : rigid_nodes) {
            r.joint_node[j] = n;
            rows_from_cols(ident, &r.ibm[(size_t)j * 12]);
            ++j;
        }
    }
    const uint32_t J = r.J;
    {
        const uint32_t mode = cfg ? cfg->exact_arithmetic : 0;
        // chosen per rig (mode 0), a far-from-origin rig gets 2: positions contraction-free (their sums cancel), normals and
        // tangents fused (3x3 rows of magnitude ~1, no translation: nothing cancels; C5 skin stage -6 %)
        r.exact = mode == 1 ? 1u : (mode == 2 ? 0u : (r.bind_extent > kExactBindExtent ? 2u : 0u));
    }
    // target -> section
    std::vector<uint32_t> target_section(M, 0);
    for (uint32_t s = 0; s < r.sections.size(); ++s)
        for (uint32_t m = 0; m < r.sections[s].target_count; ++m) target_section[r.sections[s].first_target + m] = s;

    // ---- nodes ------------------------------------------------------------------------
    r.parent.assign(d->parent, d->parent + N);
    std::vector<uint32_t> depth(N, 0);
    for (uint32_t n = 0; n < N; ++n) {
        int32_t p = r.parent[n];
        REQUIRE(p >= -1 && p < (int32_t)n, SKINRT_E_INVALID, "parent[] is not a DFS pre-order forest (need -1 <= parent[n] < n)");
        depth[n] = p < 0 ? 0 : depth[p] + 1;
    }
    REQUIRE(finite_all(d->translation, (size_t)N * 3) && finite_all(d->rotation, (size_t)N * 4) &&
                finite_all(d->scale, (size_t)N * 3), SKINRT_E_INVALID, "non-finite node transform");
    r.rest_trs.resize((size_t)N * 10);
    r.rest_mat.assign((size_t)N * 12, 0.0f);
    r.node_matrix.assign(N, 0);
    for (uint32_t n = 0; n < N; ++n) {
        float* trs = &r.rest_trs[(size_t)n * 10];
        std::memcpy(trs, d->translation + n * 3, 12);
        std::memcpy(trs + 3, d->rotation + n * 4, 16);
        std::memcpy(trs + 7, d->scale + n * 3, 12);
        if (d->has_matrix && d->has_matrix[n]) {
            REQUIRE(d->matrix != nullptr, SKINRT_E_INVALID, "has_matrix set but matrix is NULL");
            const float* m = d->matrix + (size_t)n * 16;
            REQUIRE(finite_all(m, 16), SKINRT_E_INVALID, "non-finite node matrix");
            REQUIRE(affine_last_row(m), SKINRT_E_NONAFFINE, "node matrix last row is not 0,0,0,1");
            r.node_matrix[n] = 1;
            rows_from_cols(m, &r.rest_mat[(size_t)n * 12]);
            decompose_affine(m, trs, trs + 3, trs + 7);
        }
    }
    uint32_t levels = 0;
    for (uint32_t n = 0; n < N; ++n) levels = std::max(levels, depth[n] + 1);
    r.levels = levels;
    r.level_start.assign(levels + 1, 0);
    for (uint32_t n = 0; n < N; ++n) r.level_start[depth[n] + 1]++;
    for (uint32_t l = 0; l < levels; ++l) r.level_start[l + 1] += r.level_start[l];
    r.level_nodes.resize(N);
    {
        std::vector<uint32_t> cur(r.level_start.begin(), r.level_start.end() - 1);
        for (uint32_t n = 0; n < N; ++n) r.level_nodes[cur[depth[n]]++] = n;
    }

    // ---- clips --------------------------------------------------------------------------
    REQUIRE(r.clips == 0 || d->clips, SKINRT_E_INVALID, "clips is NULL");
    r.node_chan.assign((size_t)r.clips * N * 3, -1);
    r.clip_chan_count.resize(r.clips);
    r.weight_src.assign((size_t)r.clips * M, WeightSrc{-1, 0, 0});
    r.clip_length.assign(r.clips, 0.0f);
    r.Cmax = 0;
    for (uint32_t c = 0; c < r.clips; ++c) {
        const skinrt_clip_desc& cl = d->clips[c];
        REQUIRE(cl.channel_count == 0 || cl.channels, SKINRT_E_INVALID, "clip channels is NULL");
        REQUIRE(cl.channel_count <= 65535, SKINRT_E_INVALID, "more than 65535 channels in a clip");
        r.clip_chan_count[c] = cl.channel_count;
        r.Cmax = std::max(r.Cmax, cl.channel_count);
        for (uint32_t k = 0; k < cl.channel_count; ++k) {
            const skinrt_channel_desc& ch = cl.channels[k];
            REQUIRE(ch.target_node < N, SKINRT_E_RANGE, "channel target_node out of range");
            REQUIRE(ch.path <= SKINRT_PATH_WEIGHTS, SKINRT_E_INVALID, "channel path invalid");
            const bool wchan = ch.path == SKINRT_PATH_WEIGHTS;
            REQUIRE(!wchan || M > 0, SKINRT_E_INVALID, "weights channel on a rig without morph targets");
            REQUIRE(ch.interpolation <= SKINRT_INTERP_CUBICSPLINE, SKINRT_E_INVALID, "channel interpolation invalid");
            REQUIRE(ch.key_count >= 1 && ch.times && ch.values, SKINRT_E_INVALID, "channel without keyframes");
            uint32_t wfirst = 0, wwidth = M;  // a weights channel drives the targets of the section on its node
            if (wchan && !one_section) {
                wwidth = 0;
                for (const skinrt_section_layout& sc : r.sections)
                    if (sc.node == ch.target_node && sc.target_count) { wfirst = sc.first_target; wwidth = sc.target_count; break; }
                REQUIRE(wwidth > 0, SKINRT_E_INVALID, "weights channel targets a node without a morphed mesh section");
            }
            const uint32_t w = wchan ? wwidth : (ch.path == SKINRT_PATH_ROTATION ? 4 : 3);
            const uint32_t per_key = (ch.interpolation == SKINRT_INTERP_CUBICSPLINE ? 3 : 1) * w;
            REQUIRE(finite_all(ch.times, ch.key_count), SKINRT_E_INVALID, "non-finite key time");
            for (uint32_t i = 1; i < ch.key_count; ++i)
                REQUIRE(ch.times[i] >= ch.times[i - 1], SKINRT_E_INVALID, "key times are not sorted");
            r.clip_length[c] = std::max(r.clip_length[c], ch.times[ch.key_count - 1]);
            if (wchan) {
                REQUIRE(finite_all(ch.values, (size_t)ch.key_count * per_key), SKINRT_E_INVALID, "non-finite morph weight key");
                bool first_of_clip = true;
                for (uint32_t m = 0; m < M; ++m) first_of_clip = first_of_clip && r.weight_src[(size_t)c * M + m].chan < 0;
                for (uint32_t m = wfirst; m < wfirst + wwidth; ++m) {
                    WeightSrc& ws = r.weight_src[(size_t)c * M + m];
                    REQUIRE(ws.chan < 0, SKINRT_E_INVALID, "two weights channels of one clip drive the same mesh");
                    ws = WeightSrc{(int32_t)r.chans.size(), (uint16_t)wfirst, (uint16_t)wwidth};
                }
                if (first_of_clip) ++r.weight_channel_clips;
            } else {
                int32_t& slot = r.node_chan[((size_t)c * N + ch.target_node) * 3 + ch.path];
                REQUIRE(slot < 0, SKINRT_E_INVALID, "two channels of one clip target the same node property");
                slot = (int32_t)r.chans.size();
            }
            ChanRec rec;
            rec.key_count = ch.key_count;
            rec.time_off = (uint32_t)r.times.size();
            rec.value_off = (uint32_t)r.values.size();
            rec.local = (uint16_t)k;
            rec.interp = ch.interpolation;
            rec.path = ch.path;
            r.chans.push_back(rec);
            r.times.insert(r.times.end(), ch.times, ch.times + ch.key_count);
            r.values.insert(r.values.end(), ch.values, ch.values + (size_t)ch.key_count * per_key);
            REQUIRE(r.values.size() < (1ull << 32), SKINRT_E_INVALID, "keyframe pool exceeds 2^32 floats");
        }
    }

    // ---- mesh ------------------------------------------------------------------------------
    const uint32_t TV = kTileVerts;
    const uint32_t T = (V + TV - 1) / TV;
    r.T = T;
    r.tangent_stride = d->tangent_stride ? d->tangent_stride : 3;
    if (V) {
        REQUIRE(d->position && d->normal && d->tangent, SKINRT_E_INVALID, "vertex arrays missing (reference: expect(\"No positions in primitive\"), loader/gltf/mod.rs:1050)");
        REQUIRE(r.tangent_stride == 3 || r.tangent_stride == 4, SKINRT_E_INVALID, "tangent_stride must be 3 or 4");
        REQUIRE(K >= 1 && K <= 4, SKINRT_E_INVALID, "need 1..4 joint/weight sets");
        REQUIRE(J >= 1, SKINRT_E_INVALID, "skinned vertices but no joints");
        bool any_skinned = false;
        for (const skinrt_section_layout& sc : r.sections) any_skinned = any_skinned || (sc.skin >= 0 && sc.vertex_count);
        if (any_skinned) {
            REQUIRE(d->joints && d->weights, SKINRT_E_INVALID, "need 1..4 joint/weight sets");
            for (uint32_t k = 0; k < K; ++k) REQUIRE(d->joints[k] && d->weights[k], SKINRT_E_INVALID, "joint/weight set is NULL");
        }
    }
    // influence a = 4k + c of vertex v as (weight, palette entry): a skinned section's indices count from its skin's
    // first entry; an un-skinned section is weight 1 on its node's entry and nothing else — S = 1 * M (+ 0 * M ...),
    // i.e. instance.transform (shader.wgsl:148)
    std::vector<uint32_t> vertex_section(V, 0);
    for (uint32_t s = 0; s < r.sections.size(); ++s)
        std::fill_n(vertex_section.begin() + r.sections[s].first_vertex, r.sections[s].vertex_count, s);
    struct Infl { float w; uint32_t j; bool checked; };
    auto influence = [&](uint32_t v, uint32_t k, uint32_t c) -> Infl {
        const skinrt_section_layout& sc = r.sections[vertex_section[v]];
        if (sc.skin < 0) return Infl{(k == 0 && c == 0) ? 1.0f : 0.0f, sc.first_palette_entry, true};
        const float w = d->weights[k][(size_t)v * 4 + c];
        const uint32_t j = d->joints[k][(size_t)v * 4 + c];
        return Infl{w, sc.first_palette_entry + j, j < r.skins[sc.skin].count};
    };
    const uint32_t W = 4 * K;
    r.base.assign((size_t)T * 9 * TV, 0.0f);
    r.wts.assign((size_t)T * W * TV, 0.0f);
    r.lidx.assign((size_t)T * W * TV, 0);
    r.tile_joint_start.assign(T + 1, 0);
    r.tile_lds.assign(T, 1);
    r.max_tile_joints = r.max_tile_joints_any = r.max_tile_joints2 = 0;
    r.lds_tiles = 0;
    std::vector<uint32_t> stamp(J ? J : 1, UINT32_MAX), local_of(J ? J : 1, 0), list;
    for (uint32_t t = 0; t < T; ++t) {
        const uint32_t v0 = t * TV, v1 = std::min(V, v0 + TV);
        float* b = &r.base[(size_t)t * 9 * TV];
        for (uint32_t v = v0; v < v1; ++v) {
            const uint32_t l = v - v0;
            for (int e = 0; e < 3; ++e) {
                b[(0 + e) * TV + l] = d->position[(size_t)v * 3 + e];
                b[(3 + e) * TV + l] = d->normal[(size_t)v * 3 + e];
                b[(6 + e) * TV + l] = d->tangent[(size_t)v * r.tangent_stride + e];
            }
        }
        // tile joint list: joints referenced with a non-zero weight
        list.clear();
        for (uint32_t v = v0; v < v1; ++v)
            for (uint32_t k = 0; k < K; ++k)
                for (uint32_t c = 0; c < 4; ++c) {
                    const Infl in = influence(v, k, c);
                    const float w = in.w;
                    const uint32_t j = in.j;
                    REQUIRE(std::isfinite(w), SKINRT_E_INVALID, "non-finite joint weight");
                    if (w == 0.0f) continue;  // contributes exactly 0 whatever the index
                    REQUIRE(in.checked, SKINRT_E_RANGE, "vertex joint index out of range");
                    if (stamp[j] != t) { stamp[j] = t; list.push_back(j); }
                }
        if (list.empty() && J) {  // every weight of the tile is zero: any entry of its first vertex's own skin will do
            const uint32_t j0 = r.sections[vertex_section[v0]].first_palette_entry;
            list.push_back(j0); stamp[j0] = t;
        }
        std::sort(list.begin(), list.end());
        for (uint32_t i = 0; i < list.size(); ++i) local_of[list[i]] = i;
        const bool lds = list.size() <= kTileJointCap;
        // 2: short list (its 16-byte palette pieces fit two LDS-DMA instructions per wave, or up to six for 22 .. 64 joints:
        // the self-staging kernel), 1: fits the loader-wave kernel's LDS staging, 0: gathers from L2
        r.tile_lds[t] = lds ? (list.size() <= kSkin2MaxJoints ? 2 : 1) : 0;
        r.tile_kind_count[r.tile_lds[t]] += 1;
        r.lds_tiles += lds ? 1 : 0;
        r.max_tile_joints_any = std::max<uint32_t>(r.max_tile_joints_any, (uint32_t)list.size());
        if (lds) r.max_tile_joints = std::max<uint32_t>(r.max_tile_joints, (uint32_t)list.size());
        if (r.tile_lds[t] == 2) r.max_tile_joints2 = std::max<uint32_t>(r.max_tile_joints2, (uint32_t)list.size());
        r.tile_joint_start[t + 1] = r.tile_joint_start[t] + (uint32_t)list.size();
        r.tile_joints.insert(r.tile_joints.end(), list.begin(), list.end());
        float* wt = &r.wts[(size_t)t * W * TV];
        uint16_t* li = &r.lidx[(size_t)t * W * TV];
        for (uint32_t v = v0; v < v1; ++v)
            for (uint32_t k = 0; k < K; ++k)
                for (uint32_t c = 0; c < 4; ++c) {
                    const Infl in = influence(v, k, c);
                    const float w = in.w;
                    const uint32_t j = in.j;
                    const uint32_t a = k * 4 + c, l = v - v0;
                    wt[a * TV + l] = w;
                    // zero-weight influences read the tile's first joint: 0 * finite = 0 either way
                    const uint32_t jj = (w == 0.0f) ? list[0] : j;
                    li[a * TV + l] = (uint16_t)(lds ? local_of[jj] : jj);
                }
        // lanes past the last vertex repeat the tile's first vertex: they compute a position that
        // already exists, so the bounds reduction needs no lane mask (their stores are masked)
        for (uint32_t l = v1 - v0; l < TV; ++l) {
            for (int e = 0; e < 9; ++e) b[e * TV + l] = b[e * TV];
            for (uint32_t a = 0; a < W; ++a) { wt[a * TV + l] = wt[a * TV]; li[a * TV + l] = li[a * TV]; }
        }
    }
    // Waves (64 consecutive lanes of a tile) whose weight sets beyond the first are zero in every lane can stop after
    // the sets they use (wave_influences in kernels_skin.hip); the kernels that test for it are chosen when at least an
    // eighth of the (wave, extra set) units of a two-set rig would be skipped.
    r.skippable_set_waves = r.extra_set_waves = 0;
    if (K >= 2)
        for (uint32_t t = 0; t < T; ++t)
            for (uint32_t wv = 0; wv < TV / 64; ++wv) {
                uint32_t last = 0;  // last set with a non-zero weight in this wave
                const float* wt = &r.wts[(size_t)t * W * TV + wv * 64];
                for (uint32_t a = 4; a < W; ++a)
                    for (uint32_t l = 0; l < 64; ++l)
                        if (wt[a * TV + l] != 0.0f) { last = a / 4; break; }
                r.extra_set_waves += K - 1;
                r.skippable_set_waves += K - 1 - last;
            }
    r.skip_sets = K == 2 && r.skippable_set_waves * 8 >= r.extra_set_waves && r.skippable_set_waves > 0;
    if (V) {
        r.joints0.assign((size_t)V * 4, 0);
        r.weights0.assign((size_t)V * 4, 0.0f);
        for (const skinrt_section_layout& sc : r.sections)
            if (sc.skin >= 0 && sc.vertex_count) {
                std::copy_n(d->joints[0] + (size_t)sc.first_vertex * 4, (size_t)sc.vertex_count * 4, r.joints0.begin() + (size_t)sc.first_vertex * 4);
                std::copy_n(d->weights[0] + (size_t)sc.first_vertex * 4, (size_t)sc.vertex_count * 4, r.weights0.begin() + (size_t)sc.first_vertex * 4);
            }
    }

    // ---- morph targets -> per-vertex lists (targets ascending) ----------------------------
    REQUIRE(M == 0 || d->targets, SKINRT_E_INVALID, "targets is NULL");
    r.default_weights.assign(M, 0.0f);
    if (d->default_weights) r.default_weights.assign(d->default_weights, d->default_weights + M);
    // (tile, target) pairs: every target that moves at least one vertex of a tile gets one pair
    // there, targets ascending inside a tile.  >= kMorphDenseMin touched vertices: a dense
    // [9][256] SoA block (zeros elsewhere: w*0 adds exactly 0); fewer: a 256-byte slot map plus
    // the touched vertices' deltas.
    struct PairTmp {
        uint32_t target;
        std::vector<uint8_t> lv;
        std::vector<float> dl;
    };
    std::vector<std::vector<PairTmp>> per_tile(T);
    r.morph_entries = 0;
    for (uint32_t m = 0; m < M; ++m) {
        const skinrt_morph_target_desc& tg = d->targets[m];
        const skinrt_section_layout& sc = r.sections[target_section[m]];  // rows and indices count from the section's first vertex
        if (!tg.indices) REQUIRE(tg.count == sc.vertex_count, SKINRT_E_INVALID, "dense morph target count != vertex_count");
        uint32_t prev = 0;
        for (uint32_t row = 0; row < tg.count; ++row) {
            const uint32_t lv = tg.indices ? tg.indices[row] : row;
            if (tg.indices) {
                REQUIRE(lv < sc.vertex_count, SKINRT_E_RANGE, "morph target vertex index out of range");
                REQUIRE(row == 0 || lv > prev, SKINRT_E_INVALID, "morph target indices must be strictly increasing");
                prev = lv;
            }
            const uint32_t v = sc.first_vertex + lv;
            float dl[9];
            for (int e = 0; e < 3; ++e) {
                dl[e] = tg.position ? tg.position[(size_t)row * 3 + e] : 0.0f;
                dl[3 + e] = tg.normal ? tg.normal[(size_t)row * 3 + e] : 0.0f;
                dl[6 + e] = tg.tangent ? tg.tangent[(size_t)row * r.tangent_stride + e] : 0.0f;
            }
            bool nz = false;
            for (int e = 0; e < 9; ++e) {
                REQUIRE(std::isfinite(dl[e]), SKINRT_E_INVALID, "non-finite morph delta");
                nz |= dl[e] != 0.0f;
            }
            if (!nz) continue;  // an all-zero delta adds exactly 0: dropped
            std::vector<PairTmp>& tp = per_tile[v / TV];
            if (tp.empty() || tp.back().target != m) tp.push_back(PairTmp{m, {}, {}});
            tp.back().lv.push_back((uint8_t)(v % TV));
            tp.back().dl.insert(tp.back().dl.end(), dl, dl + 9);
            r.morph_entries++;
        }
    }
    // A rig with ANY compacted pair runs k_skin2's SPARSE instantiations, which carry a layout bit per pair through the
    // instance loop of every tile (C3 +2.5 %, C5 +6.6 %: profiles/r03_ubench.txt).  Where compacted pairs are a small
    // minority — a target cut in two by a mesh-section boundary inside a tile, the rim of a morphed region — storing
    // them dense as well (zeros elsewhere: w * 0 adds exactly 0, same sums) costs 9 KB each and keeps the whole crowd
    // on the dense-only kernels.  Rigs made of small targets keep the compact layout.
    bool promote = false;
    {
        uint64_t n_pairs = 0, n_compact = 0;
        for (uint32_t t = 0; t < T; ++t)
            for (const PairTmp& pt : per_tile[t]) { ++n_pairs; n_compact += pt.lv.size() < kMorphDenseMin ? 1 : 0; }
        promote = n_compact > 0 && n_compact * 8 <= n_pairs && n_compact * 9 * TV * sizeof(float) <= (64ull << 20);
        r.promoted_pairs = promote ? (uint32_t)n_compact : 0;
    }
    r.tile_pair_start.assign(T + 1, 0);
    r.max_tile_pairs = 0;
    for (uint32_t t = 0; t < T; ++t) {
        r.tile_pair_start[t + 1] = r.tile_pair_start[t] + (uint32_t)per_tile[t].size();
        r.max_tile_pairs = std::max<uint32_t>(r.max_tile_pairs, (uint32_t)per_tile[t].size());
        const uint32_t nvalid = std::min(V - t * TV, TV);
        for (const PairTmp& pt : per_tile[t]) {
            MorphPair mp{};
            mp.target = (uint16_t)pt.target;
            mp.dense = (promote || pt.lv.size() >= kMorphDenseMin) ? 1 : 0;
            mp.data_off = (uint32_t)r.mblock.size();
            if (mp.dense) {
                mp.floats = 9 * TV;
                r.mblock.resize(r.mblock.size() + mp.floats, 0.0f);
                float* blk = &r.mblock[mp.data_off];
                auto at = [&](uint32_t lane, int e) -> float& { return blk[(lane / 64) * kMorphWaveFloats + e * 64 + (lane % 64)]; };
                for (size_t k = 0; k < pt.lv.size(); ++k)
                    for (int e = 0; e < 9; ++e) at(pt.lv[k], e) = pt.dl[k * 9 + e];
                for (uint32_t l = nvalid; l < TV; ++l)  // padded lanes mirror lane 0
                    for (int e = 0; e < 9; ++e) at(l, e) = at(0, e);
                for (uint32_t l = 0; l < TV; ++l)
                    for (int e = 0; e < 9; ++e)
                        if (at(l, e) != 0.0f) mp.wave_mask |= 1u << (l / 64);
            } else {
                mp.floats = (TV / 4 + (uint32_t)pt.dl.size() + 3) & ~3u;
                r.mblock.resize(r.mblock.size() + mp.floats, 0.0f);
                uint8_t* map = reinterpret_cast<uint8_t*>(&r.mblock[mp.data_off]);
                for (size_t k = 0; k < pt.lv.size(); ++k) map[pt.lv[k]] = (uint8_t)(k + 1);
                for (uint32_t l = nvalid; l < TV; ++l) map[l] = map[0];
                for (uint32_t l = 0; l < TV; ++l)
                    if (map[l]) mp.wave_mask |= 1u << (l / 64);
                std::memcpy(&r.mblock[mp.data_off + TV / 4], pt.dl.data(), pt.dl.size() * 4);
            }
            REQUIRE(r.mblock.size() < (1ull << 32), SKINRT_E_INVALID, "morph delta pool exceeds 2^32 floats");
            r.pairs.push_back(mp);
        }
        std::vector<PairTmp>().swap(per_tile[t]);
    }
    r.mblock.resize(r.mblock.size() + 256, 0.0f);  // slack so a full-wave 16-byte piece never reads past the pool

    // k_skin2 sizes every block's palette ring by the LONGEST joint list among its tiles, and the ring decides how many
    // blocks a CU holds.  A few long lists — tiles that straddle two mesh sections with different skins, a hand among
    // torso tiles — must not cost every tile a resident block: if handing at most 2 % of the kernel's tiles over to
    // k_skin_fix (kind 3: every instance of the tile goes on its work list; that kernel stages any list of up to 64
    // joints and runs beside k_skin2 in what k_skin2 leaves free) buys one, do it.
    // (C3's avatar cut into three skins: lists of 8 except three straddling tiles with 15; 4 -> 5 blocks per CU.)
    if (K <= 2 && r.tile_kind_count[2] > 0) {
        uint32_t hist[kSkin2MaxJoints + 1] = {0};
        for (uint32_t t = 0; t < T; ++t)
            if (r.tile_lds[t] == 2) hist[r.tile_joint_start[t + 1] - r.tile_joint_start[t]] += 1;
        // what the plan (api.hip: plan_skin) will make of a ring sized for lists of jt joints: resident blocks first, and at
        // equal blocks two instances of palette lookahead rather than one
        auto blocks = [&](uint32_t jt) {
            const uint32_t b1 = skin2_blocks_per_cu(skin2_lds_bytes(jt, 1, M != 0, r.max_tile_pairs));
            const uint32_t b2 = skin2_blocks_per_cu(skin2_lds_bytes(jt, 2, M != 0, r.max_tile_pairs));
            return b1 > b2 ? 2 * b1 : 2 * b2 + 1;
        };
        const uint32_t full = r.max_tile_joints2, tiles2 = r.tile_kind_count[2];
        uint32_t cap = full, moved = 0;
        for (uint32_t c = full, gone = 0; c > 1; --c) {
            gone += hist[c];  // tiles with lists longer than c - 1
            if ((uint64_t)gone * 50 > tiles2) break;
            if (blocks(c - 1) > blocks(cap)) { cap = c - 1; moved = gone; }
        }
        if (cap < full) {
            r.max_tile_joints2 = 0;
            for (uint32_t t = 0; t < T; ++t) {
                if (r.tile_lds[t] != 2) continue;
                const uint32_t n = r.tile_joint_start[t + 1] - r.tile_joint_start[t];
                if (n > cap) r.tile_lds[t] = 3;
                else r.max_tile_joints2 = std::max(r.max_tile_joints2, n);
            }
            r.tile_kind_count[2] -= moved;
            r.tile_kind_count[3] += moved;
        }
        // What is left may still be a mixed lot — most tiles with short lists, more than 2 % with long ones (hands, hair, a
        // collar) — and one ring size for all of them costs the short tiles the long tiles' LDS (C3's avatar with a tenth of its
        // tiles on 36 joints: 4.70 ms against 4.03).  Then k_skin2 runs as two launches, each with the ring its class needs.
        // The cut is chosen with the measured cost of a ring size (tools/exp_spread.py: the stage against joints per tile,
        // i.e. against resident blocks and lookahead), and taken only where it promises 3 % of the stage.
        {
            auto cost = [&](uint32_t jt) {
                switch (blocks(jt)) {  // 2 * blocks per CU (+ 1 with two instances of lookahead)
                    case 15: case 14: case 13: case 12: case 11: return 1.00;
                    case 10: return 1.03;
                    case 9: return 1.04;
                    case 8: return 1.12;
                    case 7: return 1.20;
                    case 6: return 1.30;
                    default: return 1.50;
                }
            };
            uint32_t h2[kSkin2MaxJoints + 1] = {0};
            for (uint32_t t = 0; t < T; ++t)
                if (r.tile_lds[t] == 2) h2[r.tile_joint_start[t + 1] - r.tile_joint_start[t]] += 1;
            const uint32_t full2 = r.max_tile_joints2, n2 = r.tile_kind_count[2];
            const double whole = (double)n2 * cost(full2);
            double best = whole * 0.97;
            uint32_t below = 0, longest = 0;
            for (uint32_t c = 1; c < full2; ++c) {
                below += h2[c];
                if (h2[c]) longest = c;
                if (below == 0 || below == n2 || !h2[c]) continue;
                const double split = (double)below * cost(longest) + (double)(n2 - below) * cost(full2) + 0.005 * n2;  // (a second launch's tail)
                if (split < best) { best = split; r.split_joints = c; r.split_joints_short = longest; }
            }
        }
    }
}

int compile_rig(const skinrt_rig_desc* desc, const skinrt_config* cfg, CompiledRig& out, std::string& err) {
    try {
        compile_impl(desc, cfg, out);
    } catch (const Fail& f) {
        err = f.msg;
        return f.code;
    } catch (const std::bad_alloc&) {
        err = "out of host memory while compiling rig";
        return SKINRT_E_NOMEM;
    }
    return SKINRT_OK;
}

void fill_rig_info(const CompiledRig& r, uint64_t device_bytes, skinrt_rig_info* info) {
    std::memset(info, 0, sizeof *info);
    info->node_count = r.N;
    info->joint_count = 0;  // joints of the skins; palette_count adds the un-skinned sections' entries
    for (const SkinSpan& sk : r.skins) info->joint_count += sk.count;
    info->vertex_count = r.V; info->set_count = r.K;
    info->target_count = r.M; info->clip_count = r.clips; info->level_count = r.levels;
    info->tile_count = r.T; info->tile_vertices = kTileVerts; info->max_tile_joints = r.max_tile_joints_any;
    info->lds_tiles = r.lds_tiles; info->max_channels = r.Cmax; info->morph_entries = r.morph_entries;
    info->device_bytes = device_bytes;
    info->exact_arithmetic = r.exact;
    info->bind_extent = r.bind_extent;
    info->weight_channel_clips = r.weight_channel_clips;
    info->skip_weight_sets = r.skip_sets ? 1u : 0u;
    info->skin_count = (uint32_t)r.skins.size();
    info->section_count = (uint32_t)r.sections.size();
    info->palette_count = r.J;
    info->handed_over_tiles = r.tile_kind_count[3];
}

}  // namespace skinrt
