// api.hip — the C ABI of include/skinrt.h over the HIP kernels.  No torch types, no
// exceptions across the boundary; every HIP failure becomes SKINRT_E_DEVICE with text.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "device_types.hpp"

using namespace skinrt;
#ifdef SKINRT_STAMPS
namespace skinrt { void debug_dump_stamps(); }
#endif

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? SKINRT_E_NOMEM : SKINRT_E_DEVICE,         \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                   \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t alloc(size_t n) {
        bytes = n;
        if (n == 0) { p = nullptr; return hipSuccess; }
        return hipMalloc(&p, n);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace

// Scheduling knobs for measurements, read from the environment ONCE per skinrt_init (never per frame) and documented
// in include/skinrt.h; none of them changes a result.
struct EnvKnobs {
    int fix_stream = 1;          // SKINRT_FIX_STREAM=0: the skin stage's kernels one after the other
    int fix_prio = 0;            // SKINRT_FIX_PRIO: -1 urgent, 0 default, 1 least urgent side stream
    uint32_t fix_order = 0;      // SKINRT_FIX_ORDER=1: enqueue k_skin2 before k_skin_fix
    uint32_t fix_blocks = 32768; // SKINRT_FIX_BLOCKS: about this many k_skin_fix blocks
    int pose_pre = 1;            // SKINRT_POSE_PRE=0: sample channels inside the pose kernels (no local_t scratch)
    int tile_chunks = 0;         // SKINRT_TILE_CHUNKS: 1, 2, 4 or 8 tile chunks (0: chosen per rig)
    int range2 = 0;              // SKINRT_RANGE2: instances per k_skin2 block (0: chosen per crowd)
    int tile_run = 0;            // SKINRT_TILE_RUN: tiles per run dealt to an XCD (0: chosen per crowd)
    int skip_sets = -1;          // SKINRT_SKIP_SETS: force the kernels that stop a wave after its last used weight set on / off
    int lds_pad = 0;             // SKINRT_LDS_PAD: bytes of unused LDS added to every k_skin2 block (occupancy experiments)
    static int num(const char* name, int dflt) {
        const char* v = std::getenv(name);
        return v ? std::atoi(v) : dflt;
    }
    void read() {
        fix_stream = num("SKINRT_FIX_STREAM", 1);
        fix_prio = num("SKINRT_FIX_PRIO", 0);
        fix_order = (uint32_t)std::max(0, num("SKINRT_FIX_ORDER", 0));
        fix_blocks = (uint32_t)std::max(1, num("SKINRT_FIX_BLOCKS", 32768));
        pose_pre = num("SKINRT_POSE_PRE", 1);
        tile_chunks = num("SKINRT_TILE_CHUNKS", 0);
        range2 = num("SKINRT_RANGE2", 0);
        tile_run = num("SKINRT_TILE_RUN", 0);
        skip_sets = num("SKINRT_SKIP_SETS", -1);
        lds_pad = std::max(0, std::min(32 * 1024, num("SKINRT_LDS_PAD", 0)));
    }
};

struct skinrt_ctx {
    skinrt_config cfg{};
    EnvKnobs env;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;  // k_skin_fix next to k_skin2 (launch_skin)
};

struct skinrt_rig {
    skinrt_ctx* ctx = nullptr;
    CompiledRig host;          // tables kept on the host for exports / info (vectors stay small vs HBM)
    RigDev dev{};
    std::vector<DevBuf> bufs;
    uint64_t device_bytes = 0;
    DevBuf default_mw;
    const uint32_t* joints0_dev = nullptr;  // [V][2] (four u16), for the interleaved export
    const float* weights0_dev = nullptr;    // [V][4]
};

struct skinrt_crowd {
    skinrt_ctx* ctx = nullptr;
    skinrt_rig* rig = nullptr;
    uint32_t I = 0;
    DevBuf tau, clip, mw, mw_eff, local_t, root, world, palette, keyidx, pos, nrm, tan, aabb_part, aabb, fix, amask;
    bool has_root = false;
    // rigs with weights channels: the host set morph weights / clocks / clips since mw_eff was last sampled, so a skin
    // stage without a pose stage in front of it must refresh mw_eff first
    bool mw_stale = true;
    SkinPlan plan{};
    PosePath pose_path = kPoseWave;
    skinrt_placement_info place{};
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // 0-3 stage stamps, 4-5 fork / join of the side stream
    uint32_t last_mask = 0;
    CrowdDev dev() const {
        CrowdDev c{};
        c.I = I;
        c.tau = tau.as<float>(); c.clip = clip.as<uint32_t>();
        c.mw_in = mw.as<float>();
        c.mw = mw_eff.p ? mw_eff.as<float>() : mw.as<float>();  // rigs with weights channels: what the pose stage sampled
        c.root = has_root ? root.as<float>() : nullptr;
        c.world = world.as<float>(); c.palette = palette.as<float>(); c.keyidx = keyidx.as<uint32_t>();
        c.pos = pos.as<float>(); c.nrm = nrm.as<float>(); c.tan = tan.as<float>();
        c.aabb_part = aabb_part.as<float>(); c.aabb = aabb.as<float>();
        c.fix = fix.p ? fix.as<uint32_t>() : nullptr;
        c.fix_inst = fix.p ? fix.as<uint32_t>() + (rig->host.T + 3) / 4 * 4 : nullptr;
        c.amask = amask.p ? amask.as<uint64_t>() : nullptr;
        return c;
    }
};

namespace {

template <class T>
int upload(skinrt_rig* r, const std::vector<T>& v, const T** out) {
    DevBuf b;
    // never hand the kernels a null table: keep at least one element
    const size_t n = std::max<size_t>(v.size(), 1) * sizeof(T);
    HIP_TRY(b.alloc(n));
    r->bufs.push_back(b);
    r->device_bytes += n;
    if (!v.empty()) HIP_TRY(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    else HIP_TRY(hipMemset(b.p, 0, n));
    *out = b.as<T>();
    return SKINRT_OK;
}

uint32_t ceil_log2(uint32_t x) {
    uint32_t s = 0;
    while ((1u << s) < x) ++s;
    return s;
}

SkinPlan plan_skin(const CompiledRig& r, uint32_t I, const skinrt_config& cfg, const EnvKnobs& env) {
    SkinPlan p{};
    p.jt_shift = ceil_log2(std::max<uint32_t>(1, r.max_tile_joints));
    const uint32_t JTP = 1u << p.jt_shift;
    p.max_pairs = r.max_tile_pairs;
    p.pair_gshift = std::min<uint32_t>(6, ceil_log2(std::max<uint32_t>(1, r.max_tile_pairs)));
    p.lds_tiles = r.lds_tiles;
    // per instance: palette rows, morph weights, bounds partials, active-pair list (+ count)
    // (double-buffered: the loader wave fills one buffer while the compute waves read the other)
    const uint32_t per_inst = 2 * (JTP * 96 + 16 * 6 * 4 + (p.max_pairs + 1) * 16);
    // 4 blocks (20 waves) per CU share 160 KiB: keep a block under 40 KiB
    // block-lifetime tables: per-wave morph slots, the tile's pair records, instance order (mask + perm)
    const uint32_t fixed = (r.M ? 4 * kMorphSlots * kMorphWaveFloats * 4 : 0) + p.max_pairs * 16 + 256 * 4 + 256 * 2;
    const uint32_t budget = fixed < 30 * 1024 ? 40 * 1024 - fixed : 10 * 1024;
    p.batch = std::min<uint32_t>(16, std::max<uint32_t>(1, budget / per_inst));
    p.lds_bytes = p.batch * per_inst + fixed;
    const uint64_t want_blocks = 4096;
    // XCD split: as many tile chunks as keep the chunks balanced within 4 %, the rest of the 8
    // XCDs split the ranges (small meshes: every XCD takes all tiles of its own ranges)
    const uint32_t T = std::max<uint32_t>(1, r.T);
    p.tile_chunks = 1;
    for (uint32_t tcs = 8; tcs > 1; tcs >>= 1) {
        const uint32_t per = (T + tcs - 1) / tcs;
        if ((uint64_t)per * tcs * 100 <= (uint64_t)T * 104) { p.tile_chunks = tcs; break; }
    }
    if (env.tile_chunks == 1 || env.tile_chunks == 2 || env.tile_chunks == 4 || env.tile_chunks == 8)  // (the other XCDs split the instance ranges)
        p.tile_chunks = (uint32_t)env.tile_chunks;
    p.tiles_per_chunk = (T + p.tile_chunks - 1) / p.tile_chunks;
    const uint32_t rcs = 8 / p.tile_chunks;
    // ---- self-staging kernel (k_skin2): 4 waves per block, every wave stages its own palette ring
    // and morph slots, 5 blocks (20 compute waves) per CU -> a block must stay under 32 KiB of LDS
    // bit 0 / 1 / 2: tiles of kind 0 / 1 / 2-or-3 exist; bit 3: tiles of kind 3 (all instances handed to k_skin_fix)
    p.kinds = (r.tile_kind_count[0] ? 1u : 0u) | (r.tile_kind_count[1] ? 2u : 0u) | ((r.tile_kind_count[2] || r.tile_kind_count[3]) ? 4u : 0u) |
              (r.tile_kind_count[3] ? 8u : 0u);
    p.v2 = (cfg.skin_path == 0 && r.K <= 2 && r.tile_kind_count[2] > 0) ? 1 : 0;
    {
        p.jt2 = std::max<uint32_t>(1, std::min<uint32_t>(r.max_tile_joints2, kSkin2MaxJoints));
        p.fix_pieces = 128;
        for (uint32_t t = 0; t < r.T; ++t)
            if (r.tile_lds[t] >= 2) {
                const uint32_t pieces = (r.tile_joint_start[t + 1] - r.tile_joint_start[t]) * 6;
                p.fix_pieces = std::max<uint32_t>(p.fix_pieces, pieces > 256 ? 384 : (pieces > 128 ? 256 : 128));
            }
        auto lds_for = [&](uint32_t look) { return skin2_lds_bytes(p.jt2, look, r.M != 0, p.max_pairs); };
        // blocks per CU the LDS allows; two instances of lookahead unless one buys another resident block
        auto blocks_for = [&](uint32_t look) { return skin2_blocks_per_cu(lds_for(look)); };
        p.look2 = blocks_for(1) > blocks_for(2) ? 1 : 2;
        p.lds_bytes2 = lds_for(p.look2) + (uint32_t)env.lds_pad;
        if (p.lds_bytes2 > kSkin2MaxLds) p.v2 = 0;
        if (r.split_joints && p.v2) {  // the short class's ring
            p.split2 = 1;
            p.jt2s = std::max<uint32_t>(1, r.split_joints_short);
            auto lds_s = [&](uint32_t look) { return skin2_lds_bytes(p.jt2s, look, r.M != 0, p.max_pairs); };
            p.look2s = skin2_blocks_per_cu(lds_s(1)) > skin2_blocks_per_cu(lds_s(2)) ? 1 : 2;
            p.lds_bytes2s = lds_s(p.look2s) + (uint32_t)env.lds_pad;
        }
        uint64_t r2 = ((uint64_t)I * std::max<uint32_t>(1, r.T) + want_blocks - 1) / want_blocks;
        p.range2 = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(r2, 1), kSkin2Range);
        if (env.range2 > 0) p.range2 = std::min<uint32_t>(kSkin2Range, (uint32_t)env.range2);
        p.ranges2 = I ? (I + p.range2 - 1) / p.range2 : 0;
        p.ranges_per_chunk2 = (std::max<uint32_t>(1, p.ranges2) + rcs - 1) / rcs;
        // a block range's palettes fit an XCD's L2 twice over: interleave fully; else runs of 16 tiles (4096 vertices), which
        // keep most of the joint sharing between neighbouring tiles (C5: as fast as one contiguous chunk) and still spread
        // a mesh's busy regions over the XCDs
        p.run2 = (uint64_t)p.range2 * r.J * 96 <= (2u << 20) ? 1 : std::min<uint32_t>(16, p.tiles_per_chunk);
        if (env.tile_run > 0) p.run2 = std::min<uint32_t>(p.tiles_per_chunk, (uint32_t)env.tile_run);
        p.tiles_per_chunk2 = (p.tiles_per_chunk + p.run2 - 1) / p.run2 * p.run2;
    }
    {   // the loader-wave kernel's instance ranges: enough blocks OF ITS OWN TILES to fill 256 CUs several times over (where
        // k_skin2 takes most tiles, the few left to k_skin want short ranges: 256 instances in a row per block would be a
        // tail as long as a third of the stage), but ranges long enough to amortise the per-block vertex load
        const uint32_t own = r.tile_kind_count[0] + r.tile_kind_count[1] + (p.v2 ? 0u : r.tile_kind_count[2] + r.tile_kind_count[3]);
        uint64_t range = ((uint64_t)I * std::max<uint32_t>(1, own) + want_blocks - 1) / want_blocks;
        range = std::min<uint64_t>(std::max<uint64_t>(range, std::max<uint32_t>(p.batch, 32)), 256);  // (shorter ranges for few tiles: measured, no gain)
        range = (range + p.batch - 1) / p.batch * p.batch;
        // the kernel's per-block instance-order tables hold 256 entries: never more than that per block
        if (range > 256) range = 256 / p.batch * p.batch;
        p.range = (uint32_t)range;
        p.ranges = I ? (I + p.range - 1) / p.range : 0;
        p.ranges_per_chunk = (std::max<uint32_t>(1, p.ranges) + rcs - 1) / rcs;
    }
    p.exact = r.exact;  // skinrt_config.exact_arithmetic resolved per rig at compile time
    p.normalize = cfg.normalize_outputs ? 1 : 0;
    p.aabb = 0;
    p.debug = 0;
#ifdef SKINRT_STAMPS
    fprintf(stderr, "[skinrt plan] I=%u T=%u jt_shift=%u max_pairs=%u batch=%u range=%u ranges=%u lds=%u tile_chunks=%u\n", I, r.T,
            p.jt_shift, p.max_pairs, p.batch, p.range, p.ranges, p.lds_bytes, p.tile_chunks);
#endif
    return p;
}

bool affine_last_row(const float* m) { return m[3] == 0.0f && m[7] == 0.0f && m[11] == 0.0f && m[15] == 1.0f; }

// ---- do the context's two streams really run side by side? ------------------------------------------------------------
// HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and two streams on one queue run their
// kernels one after the other.  In a process that already holds streams of its own — a host running torch + RCCL beside
// the library: bench.py under torchrun — the side stream came to share the main stream's queue: k_skin_fix then ran
// BEHIND k_skin2 instead of beside it and the skin stage took 4.29 ms instead of 4.04 on C3 (round 3; identical to
// SKINRT_FIX_STREAM=0).  So skinrt_init measures it: a kernel that holds the main stream for 0.2 ms, an empty one on the
// side stream, and the side stream's event tells whether it had to wait.  A side stream that waits is replaced (the
// runtime hands the next stream another queue); the streams tried are kept alive until the choice is made.
__global__ void k_hold(uint64_t ticks) {  // leaves after `ticks` of the constant-rate clock: an exit every wave reaches
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
__global__ void k_nothing() {}

hipError_t side_by_side(int device, hipStream_t main, hipStream_t side, hipEvent_t e0, hipEvent_t e1, bool* concurrent) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) { (void)hipGetLastError(); khz = 100000; }
    const uint64_t hold = (uint64_t)khz / 5;  // 0.2 ms
    float ms = 0.0f;
    for (int pass = 0; pass < 2; ++pass) {  // the first pass creates queues and loads code; the second is the measurement
        hipError_t e = hipEventRecord(e0, main);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_hold, dim3(1), dim3(64), 0, main, pass ? hold : (uint64_t)1);
        hipLaunchKernelGGL(k_nothing, dim3(1), dim3(64), 0, side);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = hipEventRecord(e1, side)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(main)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(side)) != hipSuccess) return e;
        if ((e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) { (void)hipGetLastError(); ms = 0.0f; }  // (the side event before e0: it did not wait)
    }
    *concurrent = ms < 0.1f;
    return hipSuccess;
}

int check_crowd(skinrt_crowd* c) {
    if (!c || !c->ctx || !c->rig) return fail(SKINRT_E_INVALID, "crowd handle is NULL");
    return SKINRT_OK;
}

int get_buf(skinrt_crowd* c, const DevBuf& b, size_t elem, void** out_dev, uint64_t* out_count) {
    if (int rc = check_crowd(c)) return rc;
    if (!out_dev) return fail(SKINRT_E_INVALID, "out_dev is NULL");
    *out_dev = b.p;
    if (out_count) *out_count = b.bytes / elem;
    return SKINRT_OK;
}

}  // namespace

extern "C" {

int skinrt_abi_version(void) { return SKINRT_ABI_VERSION; }

uint32_t skinrt_build_flags(void) {
    uint32_t f = skin_build_flags();
#ifdef SKINRT_STAMPS
    f |= SKINRT_BUILD_STAMPS;
#endif
    return f;
}

const char* skinrt_last_error(void) { return g_err.c_str(); }

int skinrt_init(const skinrt_config* config, skinrt_ctx** out_ctx) {
    if (!config || !out_ctx) return fail(SKINRT_E_INVALID, "skinrt_init: NULL argument");
    if (config->struct_size != sizeof(skinrt_config)) return fail(SKINRT_E_INVALID, "skinrt_config.struct_size mismatch");
    if (config->pose_path > 2) return fail(SKINRT_E_INVALID, "skinrt_config.pose_path must be 0, 1 or 2");
    if (config->skin_path > 1) return fail(SKINRT_E_INVALID, "skinrt_config.skin_path must be 0 or 1");
    if (config->exact_arithmetic > 2) return fail(SKINRT_E_INVALID, "skinrt_config.exact_arithmetic must be 0, 1 or 2");
    if (config->output_placement > 1) return fail(SKINRT_E_INVALID, "skinrt_config.output_placement must be 0 or 1");
    if (skinrt_build_flags() != 0 && !config->allow_experimental_build)
        return fail(SKINRT_E_INVALID, "this libskinrt.so is a measurement build (skinrt_build_flags() = " + std::to_string(skinrt_build_flags()) +
                                          "): its results may be wrong by design; set skinrt_config.allow_experimental_build to use it");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (config->device < 0 || config->device >= count) return fail(SKINRT_E_DEVICE, "no such HIP device");
    HIP_TRY(hipSetDevice(config->device));
    // the static work area of skinrt_calculate_tangent_noalloc: resolve its address on this device now (once per
    // device); a failure is kept and reported by that entry point, it does not concern the others
    (void)tangent_static_setup(config->device);
    skinrt_ctx* ctx = new (std::nothrow) skinrt_ctx;
    if (!ctx) return fail(SKINRT_E_NOMEM, "out of host memory");
    ctx->cfg = *config;
    ctx->env.read();
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return fail(SKINRT_E_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    {
        if (ctx->env.fix_stream != 0) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = least urgent (largest number)
            const int want = ctx->env.fix_prio;  // -1 urgent, 0 default (the product), 1 least urgent — see profiles/r03_ubench.txt: k_skin_fix ends 0.18 ms after
                                                 // k_skin2 on C3 at the least urgent priority; at the default one the two end closer together
            const int prio = want < 0 ? hi : (want > 0 ? lo : (lo + hi) / 2);
            e = hipStreamCreateWithPriority(&ctx->side, hipStreamNonBlocking, prio);
            if (e != hipSuccess) {
                (void)hipStreamDestroy(ctx->stream);
                delete ctx;
                return fail(SKINRT_E_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
            }
            // a side stream that shares the main stream's hardware queue is no side stream: look for one that does not
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
                std::vector<hipStream_t> tried;
                for (int attempt = 0; attempt < 8; ++attempt) {
                    bool concurrent = true;
                    if (side_by_side(config->device, ctx->stream, ctx->side, e0, e1, &concurrent) != hipSuccess) { (void)hipGetLastError(); break; }
                    if (concurrent) break;
                    hipStream_t next = nullptr;
                    if (hipStreamCreateWithPriority(&next, hipStreamNonBlocking, prio) != hipSuccess) { (void)hipGetLastError(); break; }
                    tried.push_back(ctx->side);
                    ctx->side = next;
                }
                for (hipStream_t t : tried) (void)hipStreamDestroy(t);
            }
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            (void)hipGetLastError();
        }
    }
    *out_ctx = ctx;
    return SKINRT_OK;
}

int skinrt_destroy_ctx(skinrt_ctx* ctx) {
    if (!ctx) return SKINRT_OK;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    if (ctx->side) {
        (void)hipStreamSynchronize(ctx->side);
        (void)hipStreamDestroy(ctx->side);
    }
    delete ctx;
    return SKINRT_OK;
}

int skinrt_validate_rig(const skinrt_config* config, const skinrt_rig_desc* desc, skinrt_rig_info* out_info) {
    CompiledRig r;
    std::string err;
    const int rc = compile_rig(desc, config, r, err);
    if (rc != SKINRT_OK) return fail(rc, err);
    if (out_info) fill_rig_info(r, 0, out_info);
    return SKINRT_OK;
}

int skinrt_destroy_rig(skinrt_rig* rig) {
    if (!rig) return SKINRT_OK;
    if (rig->ctx) (void)hipSetDevice(rig->ctx->cfg.device);
    for (auto& b : rig->bufs) b.release();
    delete rig;
    return SKINRT_OK;
}

int skinrt_upload_rig(skinrt_ctx* ctx, const skinrt_rig_desc* desc, skinrt_rig** out_rig) {
    if (!ctx || !out_rig) return fail(SKINRT_E_INVALID, "skinrt_upload_rig: NULL argument");
    skinrt_rig* r = new (std::nothrow) skinrt_rig;
    if (!r) return fail(SKINRT_E_NOMEM, "out of host memory");
    r->ctx = ctx;
    std::string err;
    int rc = compile_rig(desc, &ctx->cfg, r->host, err);
    if (rc != SKINRT_OK) {
        delete r;
        return fail(rc, err);
    }
    {
        hipError_t se = hipSetDevice(ctx->cfg.device);
        if (se != hipSuccess) {
            delete r;
            return fail(SKINRT_E_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(se));
        }
    }
    const CompiledRig& h = r->host;
    RigDev& d = r->dev;
    d.N = h.N; d.J = h.J; d.V = h.V; d.K = h.K; d.M = h.M; d.clips = h.clips; d.Cmax = h.Cmax;
    d.levels = h.levels; d.T = h.T;
    d.weight_channel_clips = h.weight_channel_clips;
    d.skip_sets = h.skip_sets ? 1u : 0u;
    d.sparse_pairs = 0;
    if (SKINRT_SPARSE_SLOTS)
        for (const MorphPair& mp : h.pairs)
            if (!mp.dense) { d.sparse_pairs = 1; break; }
    if (ctx->env.skip_sets >= 0) d.skip_sets = (h.K == 2 && ctx->env.skip_sets != 0) ? 1u : 0u;  // tools/exp_skip_sets.py
    const float* defw = nullptr;
#define UP(field) if ((rc = upload(r, h.field, &d.field)) != SKINRT_OK) { skinrt_destroy_rig(r); return rc; }
    UP(parent) UP(rest_trs) UP(rest_mat) UP(node_matrix) UP(level_start) UP(level_nodes) UP(node_chan)
    UP(chans) UP(clip_chan_count) UP(weight_src) UP(clip_length) UP(times) UP(values) UP(joint_node) UP(ibm) UP(base) UP(wts) UP(lidx)
    UP(tile_joint_start) UP(tile_joints) UP(tile_lds) UP(tile_pair_start) UP(pairs) UP(mblock)
#undef UP
    {   // the tiles ordered by kind (RigDev::tile_order)
        std::vector<uint32_t> order;
        order.reserve(h.T);
        for (uint32_t kind = 0; kind < 3; ++kind) {
            for (uint32_t t = 0; t < h.T; ++t)
                if (kind < 2 ? h.tile_lds[t] == kind : h.tile_lds[t] >= 2) order.push_back(t);
            if (kind == 0) d.order_n0 = (uint32_t)order.size();
            if (kind == 1) d.order_n1 = (uint32_t)order.size() - d.order_n0;
        }
        if ((rc = upload(r, order, &d.tile_order)) != SKINRT_OK) { skinrt_destroy_rig(r); return rc; }
        std::vector<uint32_t> order2;
        if (h.split_joints) {
            for (int pass = 0; pass < 2; ++pass) {
                for (uint32_t t = 0; t < h.T; ++t)
                    if (h.tile_lds[t] == 2 && ((h.tile_joint_start[t + 1] - h.tile_joint_start[t] <= h.split_joints) == (pass == 0))) order2.push_back(t);
                if (pass == 0) d.order2_short = (uint32_t)order2.size();
            }
            d.order2_long = (uint32_t)order2.size() - d.order2_short;
        }
        if ((rc = upload(r, order2, &d.tile_order2)) != SKINRT_OK) { skinrt_destroy_rig(r); return rc; }
    }
    if ((rc = upload(r, h.default_weights, &defw)) != SKINRT_OK) { skinrt_destroy_rig(r); return rc; }
    r->default_mw = r->bufs.back();
    {
        const uint16_t* j0 = nullptr;
        if ((rc = upload(r, h.joints0, &j0)) != SKINRT_OK) { skinrt_destroy_rig(r); return rc; }
        r->joints0_dev = reinterpret_cast<const uint32_t*>(j0);
        if ((rc = upload(r, h.weights0, &r->weights0_dev)) != SKINRT_OK) { skinrt_destroy_rig(r); return rc; }
    }
    *out_rig = r;
    return SKINRT_OK;
}

int skinrt_get_rig_info(skinrt_rig* rig, skinrt_rig_info* out_info) {
    if (!rig || !out_info) return fail(SKINRT_E_INVALID, "skinrt_get_rig_info: NULL argument");
    fill_rig_info(rig->host, rig->device_bytes, out_info);
    return SKINRT_OK;
}

int skinrt_destroy_crowd(skinrt_crowd* c) {
    if (!c) return SKINRT_OK;
#ifdef SKINRT_STAMPS
    if (c->ctx) { (void)hipSetDevice(c->ctx->cfg.device); (void)hipStreamSynchronize(c->ctx->stream); skinrt::debug_dump_stamps(); }
#endif
    if (c->ctx) {
        (void)hipSetDevice(c->ctx->cfg.device);
        (void)hipStreamSynchronize(c->ctx->stream);
    }
    for (DevBuf* b : {&c->tau, &c->clip, &c->mw, &c->mw_eff, &c->local_t, &c->root, &c->world, &c->palette, &c->keyidx, &c->pos, &c->nrm,
                      &c->tan, &c->aabb_part, &c->aabb, &c->fix, &c->amask})
        b->release();
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    delete c;
    return SKINRT_OK;
}

int skinrt_create_crowd(skinrt_ctx* ctx, skinrt_rig* rig, uint32_t instances, skinrt_crowd** out_crowd) {
    if (!ctx || !rig || !out_crowd) return fail(SKINRT_E_INVALID, "skinrt_create_crowd: NULL argument");
    if (rig->ctx != ctx) return fail(SKINRT_E_INVALID, "rig belongs to another ctx");
    if (instances == 0) return fail(SKINRT_E_INVALID, "a crowd needs at least one instance");
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    skinrt_crowd* c = new (std::nothrow) skinrt_crowd;
    if (!c) return fail(SKINRT_E_NOMEM, "out of host memory");
    c->ctx = ctx; c->rig = rig; c->I = instances;
    const CompiledRig& h = rig->host;
    const size_t I = instances;
    struct { DevBuf* b; size_t n; } plan[] = {
        {&c->tau, I * 4}, {&c->clip, I * 4}, {&c->mw, I * h.M * 4}, {&c->root, I * 48},
        {&c->mw_eff, h.weight_channel_clips ? I * h.M * 4 : 0},  // morph weights after the weights channels (pose stage)
        {&c->world, I * h.N * 48}, {&c->palette, I * h.J * 96}, {&c->keyidx, I * h.Cmax * 4},
        {&c->aabb_part, I * h.T * 24}, {&c->aabb, I * 24},      // bounds per (instance, tile)
        {&c->fix, (h.M || h.tile_kind_count[3]) ? ((size_t)h.T + 3) / 4 * 16 + I * h.T * 4 : 0},  // per-tile work lists of k_skin_fix
        {&c->amask, h.M ? I * ((h.M + 63) / 64) * 8 : 0},               // non-zero morph weights as bit rows
    };
    for (auto& it : plan) {
        hipError_t e = it.b->alloc(it.n);
        if (e != hipSuccess) {
            skinrt_destroy_crowd(c);
            return fail(SKINRT_E_NOMEM, std::string("hipMalloc of crowd buffers: ") + hipGetErrorString(e));
        }
    }
    {   // the three output arrays: placed over the HBM's regions and measured (placement.hip)
        const size_t out_bytes = I * h.V * 12;
        void *p = nullptr, *n = nullptr, *t = nullptr;
        hipError_t e = allocate_outputs(ctx->cfg.device, out_bytes, h.V, instances, ctx->cfg.output_placement, ctx->cfg.placement_max_filler_gib, ctx->stream, &p, &n, &t,
                                        &c->place);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            skinrt_destroy_crowd(c);
            return fail(e == hipErrorOutOfMemory ? SKINRT_E_NOMEM : SKINRT_E_DEVICE, std::string("output arrays: ") + hipGetErrorString(e));
        }
        c->pos.p = p; c->nrm.p = n; c->tan.p = t;
        c->pos.bytes = c->nrm.bytes = c->tan.bytes = out_bytes;
    }
    // sampled local matrices [N][I][12] of the lane-per-instance channel sampling (k_sample_nodes); allocated after the
    // output arrays so that their steering sees the same free memory with and without it
    if (h.clips && ctx->env.pose_pre != 0) {
        if (c->local_t.alloc(I * h.N * 48) != hipSuccess) {
            (void)hipGetLastError();
            skinrt_destroy_crowd(c);
            return fail(SKINRT_E_NOMEM, "hipMalloc of the sampled local matrices failed");
        }
    }
    for (int k = 0; k < 6; ++k) {
        hipEvent_t& e = c->ev[k];
        if ((k < 4 ? hipEventCreate(&e) : hipEventCreateWithFlags(&e, hipEventDisableTiming)) != hipSuccess) {
            skinrt_destroy_crowd(c);
            return fail(SKINRT_E_DEVICE, "hipEventCreate failed");
        }
    }
    // defaults: clock 0, clip 0, rig default morph weights, no root transform
    hipError_t e = hipMemsetAsync(c->tau.p, 0, c->tau.bytes, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->clip.p, 0, c->clip.bytes, ctx->stream);
    if (e == hipSuccess && h.M) e = launch_fill_morph_weights(c->mw.as<float>(), rig->default_mw.as<float>(), instances, h.M, ctx->stream);
    if (e == hipSuccess && c->mw_eff.p) e = launch_fill_morph_weights(c->mw_eff.as<float>(), rig->default_mw.as<float>(), instances, h.M, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        skinrt_destroy_crowd(c);
        return fail(SKINRT_E_DEVICE, std::string("crowd initialisation: ") + hipGetErrorString(e));
    }
    c->plan = plan_skin(h, instances, ctx->cfg, ctx->env);
    // the loader-wave kernel's LDS need grows with the longest pair list of a tile: refuse here, with a
    // reason, what skinrt_evaluate could only report as a failed launch
    if ((c->plan.kinds & 3u) || !c->plan.v2) {
        if (c->plan.lds_bytes > 64 * 1024) {
            skinrt_destroy_crowd(c);
            return fail(SKINRT_E_LIMIT, "a vertex tile is touched by too many morph targets for the loader-wave skinning kernel's LDS (about 680 per 256 vertices)");
        }
    }
    const bool fits = (size_t)h.N * 48 <= 48 * 1024;
    if (ctx->cfg.pose_path == 1) {
        if (!fits) { skinrt_destroy_crowd(c); return fail(SKINRT_E_INVALID, "pose_path=1 (wavefront) needs node_count <= 1024"); }
        c->pose_path = kPoseWave;
    } else if (ctx->cfg.pose_path == 2) c->pose_path = kPoseSplit;
    else c->pose_path = (fits && h.levels <= 256) ? kPoseWave : kPoseSplit;
    *out_crowd = c;
    return SKINRT_OK;
}

int skinrt_shard_range(uint32_t instances, uint32_t rank, uint32_t world, uint32_t* out_first, uint32_t* out_count) {
    if (!out_first || !out_count) return fail(SKINRT_E_INVALID, "skinrt_shard_range: NULL argument");
    if (world == 0 || rank >= world) return fail(SKINRT_E_RANGE, "rank must be below world");
    const uint64_t per = ((uint64_t)instances + world - 1) / world;
    const uint64_t lo = std::min<uint64_t>(instances, (uint64_t)rank * per);
    const uint64_t hi = std::min<uint64_t>(instances, lo + per);
    *out_first = (uint32_t)lo;
    *out_count = (uint32_t)(hi - lo);
    return SKINRT_OK;
}

int skinrt_set_clocks(skinrt_crowd* c, const float* tau, const uint32_t* clip_id) {
    if (int rc = check_crowd(c)) return rc;
    if (!tau) return fail(SKINRT_E_INVALID, "tau is NULL");
    HIP_TRY(hipSetDevice(c->ctx->cfg.device));
    if (clip_id && c->rig->host.clips)  // a rig without animations is always at rest, whatever the id
        for (uint32_t i = 0; i < c->I; ++i)
            if (clip_id[i] != UINT32_MAX && clip_id[i] >= c->rig->host.clips)
                return fail(SKINRT_E_RANGE, "clip_id out of range");
    HIP_TRY(hipMemcpyAsync(c->tau.p, tau, c->tau.bytes, hipMemcpyHostToDevice, c->ctx->stream));
    if (clip_id) HIP_TRY(hipMemcpyAsync(c->clip.p, clip_id, c->clip.bytes, hipMemcpyHostToDevice, c->ctx->stream));
    else HIP_TRY(hipMemsetAsync(c->clip.p, 0, c->clip.bytes, c->ctx->stream));
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));  // host buffers are borrowed for the call only
    c->mw_stale = true;
    return SKINRT_OK;
}

int skinrt_set_clocks_async(skinrt_crowd* c, const float* tau, const uint32_t* clip_id) {
    if (int rc = check_crowd(c)) return rc;
    if (!tau) return fail(SKINRT_E_INVALID, "tau is NULL");
    HIP_TRY(hipSetDevice(c->ctx->cfg.device));
    // hipMemcpyDefault: device memory or GPU-visible host memory; stream-ordered, the host does not wait
    HIP_TRY(hipMemcpyAsync(c->tau.p, tau, c->tau.bytes, hipMemcpyDefault, c->ctx->stream));
    if (clip_id) HIP_TRY(hipMemcpyAsync(c->clip.p, clip_id, c->clip.bytes, hipMemcpyDefault, c->ctx->stream));
    c->mw_stale = true;
    return SKINRT_OK;
}

int skinrt_advance_clocks(skinrt_crowd* c, float dt, const float* dev_rate) {
    if (int rc = check_crowd(c)) return rc;
    if (!std::isfinite(dt)) return fail(SKINRT_E_INVALID, "dt is not finite");
    HIP_TRY(hipSetDevice(c->ctx->cfg.device));
    HIP_TRY(launch_advance_clocks(c->rig->dev, c->tau.as<float>(), c->clip.as<uint32_t>(), c->I, dt, dev_rate, c->ctx->stream));
    c->mw_stale = true;
    return SKINRT_OK;
}

int skinrt_set_morph_weights(skinrt_crowd* c, const float* weights) {
    if (int rc = check_crowd(c)) return rc;
    const uint32_t M = c->rig->host.M;
    if (M == 0) return SKINRT_OK;
    HIP_TRY(hipSetDevice(c->ctx->cfg.device));
    if (weights) HIP_TRY(hipMemcpyAsync(c->mw.p, weights, c->mw.bytes, hipMemcpyHostToDevice, c->ctx->stream));
    else HIP_TRY(launch_fill_morph_weights(c->mw.as<float>(), c->rig->default_mw.as<float>(), c->I, M, c->ctx->stream));
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    c->mw_stale = true;
    return SKINRT_OK;
}

int skinrt_set_root_transforms(skinrt_crowd* c, const float* matrices) {
    if (int rc = check_crowd(c)) return rc;
    if (!matrices) {
        HIP_TRY(hipSetDevice(c->ctx->cfg.device));
        HIP_TRY(hipStreamSynchronize(c->ctx->stream));  // an evaluate in flight was enqueued with the old setting
        c->has_root = false;
        return SKINRT_OK;
    }
    std::vector<float> rows;
    try { rows.resize((size_t)c->I * 12); } catch (const std::bad_alloc&) { return fail(SKINRT_E_NOMEM, "out of host memory"); }
    for (uint32_t i = 0; i < c->I; ++i) {
        const float* m = matrices + (size_t)i * 16;
        if (!affine_last_row(m)) return fail(SKINRT_E_NONAFFINE, "root matrix last row is not 0,0,0,1");
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 4; ++k) rows[(size_t)i * 12 + r * 4 + k] = m[k * 4 + r];
    }
    HIP_TRY(hipSetDevice(c->ctx->cfg.device));
    // stream-ordered like the other setters: an evaluate already enqueued reads root[] before this copy
    HIP_TRY(hipMemcpyAsync(c->root.p, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, c->ctx->stream));
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));  // host buffer is borrowed for the call only
    c->has_root = true;
    return SKINRT_OK;
}

int skinrt_evaluate(skinrt_crowd* c, uint32_t mask) {
    if (int rc = check_crowd(c)) return rc;
    if (mask == 0 || (mask & ~SKINRT_STAGE_ALL)) return fail(SKINRT_E_INVALID, "bad stages_mask");
    if ((mask & SKINRT_STAGE_AABB) && !(mask & SKINRT_STAGE_SKIN)) return fail(SKINRT_E_INVALID, "STAGE_AABB needs STAGE_SKIN");
    HIP_TRY(hipSetDevice(c->ctx->cfg.device));
    hipStream_t s = c->ctx->stream;
    const RigDev& rig = c->rig->dev;
    const CrowdDev cr = c->dev();
    HIP_TRY(hipEventRecord(c->ev[0], s));
    if (mask & SKINRT_STAGE_POSE) HIP_TRY(launch_pose(rig, cr, c->ctx->cfg.cubic_quirk ? 1u : 0u, c->pose_path, c->local_t.as<float>(), s));
    // the weights the skin stage reads on rigs with weights channels: sampled with the pose, and again in front of a
    // skin stage that runs without one after the host changed weights, clocks or clips
    if (c->mw_eff.p && ((mask & SKINRT_STAGE_POSE) || ((mask & SKINRT_STAGE_SKIN) && c->mw_stale))) {
        HIP_TRY(launch_sample_weights(rig, cr, c->mw_eff.as<float>(), c->ctx->cfg.cubic_quirk ? 1u : 0u, s));
        c->mw_stale = false;
    }
    HIP_TRY(hipEventRecord(c->ev[1], s));
    if (mask & SKINRT_STAGE_SKIN) {
        SkinPlan p = c->plan;
        // 2: k_skin2 writes its bounds records non-temporally — where the crowd's palettes compete for the memory-side cache
        // (see k_skin2's epilogue); small crowds keep them cached for k_aabb_reduce
        p.aabb = (mask & SKINRT_STAGE_AABB) ? ((uint64_t)cr.I * rig.J * 96 >= (64ull << 20) ? 2 : 1) : 0;
        SkinConcurrency conc{c->ctx->side, c->ev[4], c->ev[5]};
        conc.skin2_first = c->ctx->env.fix_order;
        conc.fix_blocks = c->ctx->env.fix_blocks;
        HIP_TRY(launch_skin(rig, cr, p, s, conc));
    }
    HIP_TRY(hipEventRecord(c->ev[2], s));
    if (mask & SKINRT_STAGE_AABB) HIP_TRY(launch_aabb_reduce(rig, cr, s));
    HIP_TRY(hipEventRecord(c->ev[3], s));
    c->last_mask = mask;
    return SKINRT_OK;
}

int skinrt_synchronize(skinrt_ctx* ctx) {
    if (!ctx) return fail(SKINRT_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SKINRT_OK;
}

int skinrt_last_stage_ms(skinrt_crowd* c, float ms[4]) {
    if (int rc = check_crowd(c)) return rc;
    if (!ms) return fail(SKINRT_E_INVALID, "ms is NULL");
    if (!c->last_mask) return fail(SKINRT_E_INVALID, "no evaluate has been enqueued on this crowd");
    HIP_TRY(hipSetDevice(c->ctx->cfg.device));
    HIP_TRY(hipEventSynchronize(c->ev[3]));
    for (int k = 0; k < 3; ++k) HIP_TRY(hipEventElapsedTime(&ms[k], c->ev[k], c->ev[k + 1]));
    HIP_TRY(hipEventElapsedTime(&ms[3], c->ev[0], c->ev[3]));
    return SKINRT_OK;
}

int skinrt_get_stream(skinrt_ctx* ctx, void** out_stream) {
    if (!ctx || !out_stream) return fail(SKINRT_E_INVALID, "skinrt_get_stream: NULL argument");
    *out_stream = ctx->stream;
    return SKINRT_OK;
}

int skinrt_get_output_placement(skinrt_crowd* c, skinrt_placement_info* out_info) {
    if (int rc = check_crowd(c)) return rc;
    if (!out_info) return fail(SKINRT_E_INVALID, "out_info is NULL");
    *out_info = c->place;
    return SKINRT_OK;
}

int skinrt_get_positions(skinrt_crowd* c, float** o, uint64_t* n) { return c ? get_buf(c, c->pos, 4, (void**)o, n) : check_crowd(c); }
int skinrt_get_normals(skinrt_crowd* c, float** o, uint64_t* n) { return c ? get_buf(c, c->nrm, 4, (void**)o, n) : check_crowd(c); }
int skinrt_get_tangents(skinrt_crowd* c, float** o, uint64_t* n) { return c ? get_buf(c, c->tan, 4, (void**)o, n) : check_crowd(c); }
int skinrt_get_aabbs(skinrt_crowd* c, float** o, uint64_t* n) { return c ? get_buf(c, c->aabb, 4, (void**)o, n) : check_crowd(c); }
int skinrt_get_palette(skinrt_crowd* c, float** o, uint64_t* n) { return c ? get_buf(c, c->palette, 4, (void**)o, n) : check_crowd(c); }
int skinrt_get_world(skinrt_crowd* c, float** o, uint64_t* n) { return c ? get_buf(c, c->world, 4, (void**)o, n) : check_crowd(c); }
int skinrt_get_keyframe_index(skinrt_crowd* c, uint32_t** o, uint64_t* n) { return c ? get_buf(c, c->keyidx, 4, (void**)o, n) : check_crowd(c); }
int skinrt_get_clocks(skinrt_crowd* c, float** o, uint64_t* n) { return c ? get_buf(c, c->tau, 4, (void**)o, n) : check_crowd(c); }

int skinrt_rig_skin_layout(skinrt_rig* rig, uint32_t skin, skinrt_skin_layout* out) {
    if (!rig || !out) return fail(SKINRT_E_INVALID, "skinrt_rig_skin_layout: NULL argument");
    if (skin >= rig->host.skins.size()) return fail(SKINRT_E_RANGE, "skin index out of range");
    out->first_palette_entry = rig->host.skins[skin].first;
    out->joint_count = rig->host.skins[skin].count;
    return SKINRT_OK;
}

int skinrt_rig_section_layout(skinrt_rig* rig, uint32_t section, skinrt_section_layout* out) {
    if (!rig || !out) return fail(SKINRT_E_INVALID, "skinrt_rig_section_layout: NULL argument");
    if (section >= rig->host.sections.size()) return fail(SKINRT_E_RANGE, "section index out of range");
    *out = rig->host.sections[section];
    return SKINRT_OK;
}

int skinrt_download(skinrt_ctx* ctx, const void* dev_src, void* host_dst, uint64_t bytes) {
    if (!ctx) return fail(SKINRT_E_INVALID, "ctx is NULL");
    if (bytes == 0) return SKINRT_OK;
    if (!dev_src || !host_dst) return fail(SKINRT_E_INVALID, "skinrt_download: NULL pointer");
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return SKINRT_OK;
}

namespace {

// the skin's span of the palette, or a failure code; a rig without skins has nothing to export
int skin_span(skinrt_crowd* c, uint32_t skin, SkinSpan* out, uint32_t slots = SKINRT_MAX_JOINTS_REFERENCE) {
    const CompiledRig& h = c->rig->host;
    if (skin >= h.skins.size()) {
        if (skin == 0) { *out = SkinSpan{0, 0}; return SKINRT_OK; }  // no skin at all: an all-zero palette, as J = 0 always gave
        return fail(SKINRT_E_RANGE, "skin index out of range");
    }
    *out = h.skins[skin];
    // assert!(items.len() <= MAX_JOINTS) — renderer/src/renderer/uniform/skin.rs:40,60
    if (slots == SKINRT_MAX_JOINTS_REFERENCE && out->count > slots) return fail(SKINRT_E_LIMIT, "the reference palette holds at most 128 joints");
    if (out->count > slots) return fail(SKINRT_E_LIMIT, "fewer slots per instance than the skin has joints");
    return SKINRT_OK;
}

struct VertexSpan { uint32_t first, count; };
int section_span(skinrt_crowd* c, int64_t section, VertexSpan* out) {
    const CompiledRig& h = c->rig->host;
    if (section < 0) { *out = VertexSpan{0, h.V}; return SKINRT_OK; }  // the whole mesh
    if ((uint64_t)section >= h.sections.size()) return fail(SKINRT_E_RANGE, "section index out of range");
    const skinrt_section_layout& sc = h.sections[(size_t)section];
    if (sc.skin < 0) return fail(SKINRT_E_INVALID, "an un-skinned section has no skin vertex layout (use the static mesh and skinrt_export_instance_uniform)");
    *out = VertexSpan{sc.first_vertex, sc.vertex_count};
    return SKINRT_OK;
}

int export_skin_uniform_host(skinrt_crowd* c, uint32_t skin, uint32_t instance, void* dst) {
    if (int rc = check_crowd(c)) return rc;
    if (!dst) return fail(SKINRT_E_INVALID, "dst is NULL");
    if (instance >= c->I) return fail(SKINRT_E_RANGE, "instance out of range");
    SkinSpan sp;
    if (int rc = skin_span(c, skin, &sp)) return rc;
    const uint32_t P = c->rig->host.J, J = sp.count;
    std::vector<float> rows((size_t)J * kPalFloats);
    if (int rc = skinrt_download(c->ctx, c->palette.as<float>() + ((size_t)instance * P + sp.first) * kPalFloats, rows.data(), rows.size() * 4)) return rc;
    float* out = static_cast<float*>(dst);
    std::memset(out, 0, (size_t)SKINRT_MAX_JOINTS_REFERENCE * SKINRT_SKIN_MATRIX_BYTES);
    for (uint32_t j = 0; j < J; ++j) {
        const float* e = &rows[(size_t)j * kPalFloats];
        float* o = out + (size_t)j * 28;
        for (int col = 0; col < 4; ++col) {  // transform: [[f32;4];4] column-major, last row 0,0,0,1
            for (int r = 0; r < 3; ++r) o[col * 4 + r] = e[r * 4 + col];
            o[col * 4 + 3] = col == 3 ? 1.0f : 0.0f;
        }
        for (int col = 0; col < 3; ++col) {  // normal: 3 columns padded with 0 (pad_vec3, :26-28)
            for (int r = 0; r < 3; ++r) o[16 + col * 4 + r] = e[12 + r * 3 + col];  // the nine entries are packed row by row
            o[16 + col * 4 + 3] = 0.0f;
        }
    }
    return SKINRT_OK;
}

int export_interleaved_host(skinrt_crowd* c, int64_t section, uint32_t instance, uint32_t layout, const float* attr, void* dst) {
    if (int rc = check_crowd(c)) return rc;
    if (!dst) return fail(SKINRT_E_INVALID, "dst is NULL");
    if (instance >= c->I) return fail(SKINRT_E_RANGE, "instance out of range");
    if (layout > SKINRT_LAYOUT_TEXTURE_SKIN) return fail(SKINRT_E_INVALID, "unknown vertex layout");
    VertexSpan vs;
    if (int rc = section_span(c, section, &vs)) return rc;
    const CompiledRig& h = c->rig->host;
    const size_t V = vs.count;
    std::vector<float> p(V * 3), n(V * 3), t(V * 3);
    const size_t off = ((size_t)instance * h.V + vs.first) * 3;
    if (int rc = skinrt_download(c->ctx, c->pos.as<float>() + off, p.data(), V * 12)) return rc;
    if (int rc = skinrt_download(c->ctx, c->nrm.as<float>() + off, n.data(), V * 12)) return rc;
    if (int rc = skinrt_download(c->ctx, c->tan.as<float>() + off, t.data(), V * 12)) return rc;
    const uint32_t na = layout == SKINRT_LAYOUT_COLOR_SKIN ? 4 : 2;
    const size_t stride = layout == SKINRT_LAYOUT_COLOR_SKIN ? 76 : 68;
    uint8_t* o = static_cast<uint8_t*>(dst);
    for (size_t v = 0; v < V; ++v, o += stride) {
        uint8_t* q = o;
        std::memcpy(q, &p[v * 3], 12); q += 12;
        if (attr) std::memcpy(q, attr + v * na, na * 4); else std::memset(q, 0, na * 4);
        q += na * 4;
        std::memcpy(q, &n[v * 3], 12); q += 12;
        std::memcpy(q, &t[v * 3], 12); q += 12;
        std::memcpy(q, &h.joints0[(vs.first + v) * 4], 8); q += 8;
        std::memcpy(q, &h.weights0[(vs.first + v) * 4], 16);
    }
    return SKINRT_OK;
}

int export_skin_uniform_dev(skinrt_crowd* c, uint32_t skin, uint32_t first, uint32_t count, void* dev_dst, uint32_t slots = SKINRT_MAX_JOINTS_REFERENCE) {
    if (int rc = check_crowd(c)) return rc;
    if (count == 0) return SKINRT_OK;
    if (!dev_dst) return fail(SKINRT_E_INVALID, "dev_dst is NULL");
    if (reinterpret_cast<uintptr_t>(dev_dst) & 15u) return fail(SKINRT_E_INVALID, "dev_dst must be 16-byte aligned");
    if ((uint64_t)first + count > c->I) return fail(SKINRT_E_RANGE, "instance range out of range");
    SkinSpan sp;
    if (int rc = skin_span(c, skin, &sp, slots)) return rc;
    if (slots == 0) return SKINRT_OK;
    HIP_TRY(hipSetDevice(c->ctx->cfg.device));
    HIP_TRY(launch_export_skin_uniform(c->dev(), c->rig->host.J, sp.first, sp.count, slots, first, count, dev_dst, c->ctx->stream));
    return SKINRT_OK;
}

int export_interleaved_dev(skinrt_crowd* c, int64_t section, uint32_t first, uint32_t count, uint32_t layout, const float* dev_attr, void* dev_dst) {
    if (int rc = check_crowd(c)) return rc;
    if (layout > SKINRT_LAYOUT_TEXTURE_SKIN) return fail(SKINRT_E_INVALID, "unknown vertex layout");
    VertexSpan vs;
    if (int rc = section_span(c, section, &vs)) return rc;
    if (count == 0 || vs.count == 0) return SKINRT_OK;
    if (!dev_dst) return fail(SKINRT_E_INVALID, "dev_dst is NULL");
    if (reinterpret_cast<uintptr_t>(dev_dst) & 15u) return fail(SKINRT_E_INVALID, "dev_dst must be 16-byte aligned");
    if ((uint64_t)first + count > c->I) return fail(SKINRT_E_RANGE, "instance range out of range");
    HIP_TRY(hipSetDevice(c->ctx->cfg.device));
    HIP_TRY(launch_export_interleaved(c->dev(), c->rig->host.V, vs.first, vs.count, c->rig->joints0_dev, c->rig->weights0_dev, dev_attr, layout,
                                      first, count, dev_dst, c->ctx->stream));
    return SKINRT_OK;
}

}  // namespace

int skinrt_export_skin_uniform(skinrt_crowd* c, uint32_t instance, void* dst) { return export_skin_uniform_host(c, 0, instance, dst); }
int skinrt_export_skin_uniform_of(skinrt_crowd* c, uint32_t skin, uint32_t instance, void* dst) { return export_skin_uniform_host(c, skin, instance, dst); }
int skinrt_export_interleaved(skinrt_crowd* c, uint32_t instance, uint32_t layout, const float* attr, void* dst) {
    return export_interleaved_host(c, -1, instance, layout, attr, dst);
}
int skinrt_export_interleaved_section(skinrt_crowd* c, uint32_t section, uint32_t instance, uint32_t layout, const float* attr, void* dst) {
    return export_interleaved_host(c, section, instance, layout, attr, dst);
}
int skinrt_export_skin_uniform_device(skinrt_crowd* c, uint32_t first, uint32_t count, void* dev_dst) {
    return export_skin_uniform_dev(c, 0, first, count, dev_dst);
}
int skinrt_export_skin_uniform_of_device(skinrt_crowd* c, uint32_t skin, uint32_t first, uint32_t count, void* dev_dst) {
    return export_skin_uniform_dev(c, skin, first, count, dev_dst);
}
int skinrt_export_skin_storage_device(skinrt_crowd* c, uint32_t skin, uint32_t first, uint32_t count, uint32_t slots, void* dev_dst) {
    return export_skin_uniform_dev(c, skin, first, count, dev_dst, slots);
}
int skinrt_export_interleaved_device(skinrt_crowd* c, uint32_t first, uint32_t count, uint32_t layout, const float* dev_attr, void* dev_dst) {
    return export_interleaved_dev(c, -1, first, count, layout, dev_attr, dev_dst);
}
int skinrt_export_interleaved_section_device(skinrt_crowd* c, uint32_t section, uint32_t first, uint32_t count, uint32_t layout,
                                             const float* dev_attr, void* dev_dst) {
    return export_interleaved_dev(c, section, first, count, layout, dev_attr, dev_dst);
}

int skinrt_export_instance_uniform_device(skinrt_crowd* c, uint32_t first, uint32_t count, const uint32_t* dev_nodes, uint32_t node_count,
                                          uint32_t stride, void* dev_dst) {
    if (int rc = check_crowd(c)) return rc;
    if (count == 0 || node_count == 0) return SKINRT_OK;
    if (!dev_nodes || !dev_dst) return fail(SKINRT_E_INVALID, "skinrt_export_instance_uniform_device: NULL pointer");
    if (stride < SKINRT_INSTANCE_UNIFORM_BYTES || (stride & 3u)) return fail(SKINRT_E_INVALID, "stride must be >= 124 and a multiple of 4");
    if (reinterpret_cast<uintptr_t>(dev_dst) & 3u) return fail(SKINRT_E_INVALID, "dev_dst must be 4-byte aligned");
    if ((uint64_t)first + count > c->I) return fail(SKINRT_E_RANGE, "instance range out of range");
    HIP_TRY(hipSetDevice(c->ctx->cfg.device));
    HIP_TRY(launch_export_instance_uniform(c->dev(), c->rig->host.N, first, count, dev_nodes, node_count, stride, dev_dst, c->ctx->stream));
    return SKINRT_OK;
}

int skinrt_export_instance_uniform(skinrt_crowd* c, uint32_t instance, uint32_t node, void* dst) {
    if (int rc = check_crowd(c)) return rc;
    if (!dst) return fail(SKINRT_E_INVALID, "dst is NULL");
    if (instance >= c->I) return fail(SKINRT_E_RANGE, "instance out of range");
    if (node >= c->rig->host.N) return fail(SKINRT_E_RANGE, "node out of range");
    // the device exporter on one record, through a small staging buffer: ONE implementation of the arithmetic
    HIP_TRY(hipSetDevice(c->ctx->cfg.device));
    DevBuf stage;
    HIP_TRY(stage.alloc(128 + 16));
    hipError_t e = hipMemcpyAsync(stage.p, &node, 4, hipMemcpyHostToDevice, c->ctx->stream);
    if (e == hipSuccess)
        e = launch_export_instance_uniform(c->dev(), c->rig->host.N, instance, 1, stage.as<uint32_t>(), 1, SKINRT_INSTANCE_UNIFORM_BYTES,
                                           stage.as<char>() + 16, c->ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dst, stage.as<char>() + 16, SKINRT_INSTANCE_UNIFORM_BYTES, hipMemcpyDeviceToHost, c->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->ctx->stream);
    stage.release();
    if (e != hipSuccess) return fail(SKINRT_E_DEVICE, std::string("skinrt_export_instance_uniform: ") + hipGetErrorString(e));
    return SKINRT_OK;
}

int skinrt_calculate_tangent(skinrt_ctx* ctx, uint32_t mode, const float* positions, uint32_t vertex_count,
                             const uint32_t* indices, uint32_t index_count, float* out, float* out_device_ms) {
    if (!ctx) return fail(SKINRT_E_INVALID, "ctx is NULL");
    if (mode > 4) return fail(SKINRT_E_INVALID, "mode must be a PrimitiveAssetMode ordinal (0..4)");
    if (mode == 3) return fail(SKINRT_E_INVALID, "TriangleStrip: todo!() in the reference (tangent.rs:27)");
    if (vertex_count == 0) return SKINRT_OK;
    if (!positions || !out) return fail(SKINRT_E_INVALID, "skinrt_calculate_tangent: NULL pointer");
    if (indices == nullptr && index_count != 0) return fail(SKINRT_E_INVALID, "index_count without indices");
    // only triangle lists have faces; points and lines leave every sum at zero (and normalise to NaN, as the reference)
    uint32_t corners = 0;
    if (mode == 4) corners = (indices ? index_count : vertex_count) / 3 * 3;
    if (indices)
        for (uint32_t k = 0; k < corners; ++k)
            if (indices[k] >= vertex_count) return fail(SKINRT_E_INVALID, "vertex index out of range (reference: index panic)");
    if (indices && corners > SKINRT_TANGENT_MAX_VALENCE) {  // a vertex named by more corners than the kernels sort: refuse here, where it can be seen
        std::vector<uint32_t> valence;
        try { valence.assign(vertex_count, 0u); } catch (const std::bad_alloc&) { return fail(SKINRT_E_NOMEM, "out of host memory"); }
        for (uint32_t k = 0; k < corners; ++k)
            if (++valence[indices[k]] > SKINRT_TANGENT_MAX_VALENCE)
                return fail(SKINRT_E_LIMIT, "a vertex is named by more than SKINRT_TANGENT_MAX_VALENCE triangle corners");
    }
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    const size_t work_bytes = tangent_work_bytes(vertex_count, corners);
    DevBuf dpos, didx, dwork, dout;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = dpos.alloc((size_t)vertex_count * 12);
    if (e == hipSuccess && indices && corners) e = didx.alloc((size_t)corners * 4);
    if (e == hipSuccess) e = dwork.alloc(work_bytes);
    if (e == hipSuccess) e = dout.alloc((size_t)vertex_count * 12);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    hipStream_t s = ctx->stream;
    if (e == hipSuccess) e = hipMemcpyAsync(dpos.p, positions, (size_t)vertex_count * 12, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && didx.p) e = hipMemcpyAsync(didx.p, indices, (size_t)corners * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipEventRecord(e0, s);
    if (e == hipSuccess) e = launch_tangents(dpos.as<float>(), didx.as<uint32_t>(), vertex_count, corners, dwork.p, work_bytes, dout.as<float>(), s);
    if (e == hipSuccess) e = hipEventRecord(e1, s);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout.p, (size_t)vertex_count * 12, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && out_device_ms) e = hipEventElapsedTime(out_device_ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    dpos.release(); didx.release(); dwork.release(); dout.release();
    if (e != hipSuccess) return fail(SKINRT_E_DEVICE, std::string("skinrt_calculate_tangent: ") + hipGetErrorString(e));
    return SKINRT_OK;
}

int skinrt_calculate_tangent_noalloc(skinrt_ctx* ctx, uint32_t mode, const float* dev_positions, uint32_t vertex_count,
                                     const uint32_t* dev_indices, uint32_t index_count, float* dev_out) {
    if (!ctx) return fail(SKINRT_E_INVALID, "ctx is NULL");
    if (mode > 4) return fail(SKINRT_E_INVALID, "mode must be a PrimitiveAssetMode ordinal (0..4)");
    if (mode == 3) return fail(SKINRT_E_INVALID, "TriangleStrip: todo!() in the reference (tangent.rs:27)");
    if (vertex_count == 0) return SKINRT_OK;
    if (!dev_positions || !dev_out) return fail(SKINRT_E_INVALID, "skinrt_calculate_tangent_noalloc: NULL pointer");
    if (dev_indices == nullptr && index_count != 0) return fail(SKINRT_E_INVALID, "index_count without indices");
    if (vertex_count > SKINRT_TANGENT_NOALLOC_MAX_VERTICES)
        return fail(SKINRT_E_LIMIT, "skinrt_calculate_tangent_noalloc: more than SKINRT_TANGENT_NOALLOC_MAX_VERTICES vertices (use skinrt_calculate_tangent)");
    uint32_t corners = 0;
    if (mode == 4) corners = (dev_indices ? index_count : vertex_count) / 3 * 3;
    if (corners > SKINRT_TANGENT_NOALLOC_MAX_INDICES)
        return fail(SKINRT_E_LIMIT, "skinrt_calculate_tangent_noalloc: more than SKINRT_TANGENT_NOALLOC_MAX_INDICES indices (use skinrt_calculate_tangent)");
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    void* work = nullptr;
    size_t work_bytes = 0;
    {
        const hipError_t e = tangent_static_area(ctx->cfg.device, &work, &work_bytes);
        if (e != hipSuccess) return fail(SKINRT_E_DEVICE, std::string("static work area not available on this device: ") + hipGetErrorString(e));
    }
    if (tangent_work_bytes(vertex_count, corners) > work_bytes) return fail(SKINRT_E_LIMIT, "skinrt_calculate_tangent_noalloc: input exceeds the static work area");
    struct Call { const float* pos; const uint32_t* idx; uint32_t v, c; void* work; size_t bytes; float* out; hipStream_t s; } call{
        dev_positions, dev_indices, vertex_count, corners, work, work_bytes, dev_out, ctx->stream};
    // several ctx can share one GPU and with it the work area: this call's kernels are ordered behind the previous user's
    HIP_TRY(tangent_static_use(ctx->cfg.device, ctx->stream, [](void* p) -> hipError_t {
        const Call& k = *static_cast<const Call*>(p);
        return launch_tangents(k.pos, k.idx, k.v, k.c, k.work, k.bytes, k.out, k.s);
    }, &call));
    return SKINRT_OK;
}

// ---- per-frame normals from the deformed positions (second half) -----------------------------------
struct skinrt_topology {
    skinrt_ctx* ctx = nullptr;
    uint32_t V = 0, corners = 0;
    DevBuf idx, work;                 // the index list (absent: corner c is vertex c) and the bucket tables
    const uint32_t* start = nullptr;  // [V + 1], inside `work`
    const uint32_t* adj = nullptr;    // [corners], each vertex's corner ids ascending, inside `work`
};

int skinrt_destroy_topology(skinrt_topology* t) {
    if (!t) return SKINRT_OK;
    if (t->ctx) {
        (void)hipSetDevice(t->ctx->cfg.device);
        (void)hipStreamSynchronize(t->ctx->stream);  // a recompute in flight reads the tables
    }
    t->idx.release();
    t->work.release();
    delete t;
    return SKINRT_OK;
}

int skinrt_create_topology(skinrt_ctx* ctx, uint32_t vertex_count, const uint32_t* indices, uint32_t index_count, skinrt_topology** out_topology) {
    if (!ctx || !out_topology) return fail(SKINRT_E_INVALID, "skinrt_create_topology: NULL argument");
    if (vertex_count == 0) return fail(SKINRT_E_INVALID, "a topology needs at least one vertex");
    if (indices == nullptr && index_count != 0) return fail(SKINRT_E_INVALID, "index_count without indices");
    const uint32_t corners = (indices ? index_count : vertex_count) / 3 * 3;  // whole triangles (tangent.rs:29, :42: len() / 3)
    if (indices) {
        for (uint32_t k = 0; k < corners; ++k)
            if (indices[k] >= vertex_count) return fail(SKINRT_E_INVALID, "vertex index out of range (reference: index panic)");
        if (corners > SKINRT_TANGENT_MAX_VALENCE) {
            std::vector<uint32_t> valence;
            try { valence.assign(vertex_count, 0u); } catch (const std::bad_alloc&) { return fail(SKINRT_E_NOMEM, "out of host memory"); }
            for (uint32_t k = 0; k < corners; ++k)
                if (++valence[indices[k]] > SKINRT_TANGENT_MAX_VALENCE)
                    return fail(SKINRT_E_LIMIT, "a vertex is named by more than SKINRT_TANGENT_MAX_VALENCE triangle corners");
        }
    }
    HIP_TRY(hipSetDevice(ctx->cfg.device));
    skinrt_topology* t = new (std::nothrow) skinrt_topology;
    if (!t) return fail(SKINRT_E_NOMEM, "out of host memory");
    t->ctx = ctx; t->V = vertex_count; t->corners = corners;
    const size_t work_bytes = tangent_work_bytes(vertex_count, corners);
    hipError_t e = t->work.alloc(work_bytes);
    if (e == hipSuccess && indices && corners) e = t->idx.alloc((size_t)corners * 4);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        skinrt_destroy_topology(t);
        return fail(SKINRT_E_NOMEM, std::string("hipMalloc of the topology tables: ") + hipGetErrorString(e));
    }
    hipStream_t s = ctx->stream;
    if (t->idx.p) e = hipMemcpyAsync(t->idx.p, indices, (size_t)corners * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = launch_topology(t->idx.as<uint32_t>(), vertex_count, corners, t->work.p, work_bytes, &t->start, &t->adj, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // the host index list is borrowed for the call only
    if (e != hipSuccess) {
        (void)hipGetLastError();
        skinrt_destroy_topology(t);
        return fail(SKINRT_E_DEVICE, std::string("skinrt_create_topology: ") + hipGetErrorString(e));
    }
    *out_topology = t;
    return SKINRT_OK;
}

int skinrt_recompute_normals(skinrt_crowd* c, skinrt_topology* t, uint32_t first, uint32_t count, uint32_t output) {
    if (int rc = check_crowd(c)) return rc;
    if (!t) return fail(SKINRT_E_INVALID, "topology is NULL");
    if (t->ctx != c->ctx) return fail(SKINRT_E_INVALID, "topology belongs to another ctx");
    if (t->V != c->rig->host.V) return fail(SKINRT_E_INVALID, "topology was made for another vertex count than the rig's");
    if (output != SKINRT_OUTPUT_NORMALS && output != SKINRT_OUTPUT_TANGENTS) return fail(SKINRT_E_INVALID, "output must be SKINRT_OUTPUT_NORMALS or SKINRT_OUTPUT_TANGENTS");
    if (first > c->I || count > c->I - first) return fail(SKINRT_E_RANGE, "instance range beyond the crowd");
    if (count == 0) return SKINRT_OK;
    HIP_TRY(hipSetDevice(c->ctx->cfg.device));
    float* out = output == SKINRT_OUTPUT_NORMALS ? c->nrm.as<float>() : c->tan.as<float>();
    HIP_TRY(launch_crowd_normals(c->pos.as<float>(), t->idx.as<uint32_t>(), t->start, t->adj, t->V, first, count, out, c->ctx->stream));
    return SKINRT_OK;
}

// ---- one crowd over several devices of this process (VERDICT r1 "multi-GPU behind the C ABI") ----------
// A node is nothing but one (ctx, rig, crowd) per listed device and the sharding rule of skinrt_shard_range: there is
// no exchange step on the data path, every entry point above is asynchronous on its ctx's stream, so one host thread
// keeps all devices busy — evaluate enqueues on each shard in turn and returns.  The setters take whole-crowd host
// arrays and hand each shard its rows.
struct skinrt_node {
    struct Shard {
        skinrt_ctx* ctx = nullptr;
        skinrt_rig* rig = nullptr;
        skinrt_crowd* crowd = nullptr;  // nullptr: the shard is empty (more devices than instances)
        uint32_t first = 0, count = 0;
    };
    std::vector<Shard> shards;
    uint32_t instances = 0, M = 0;
};

int skinrt_node_destroy(skinrt_node* node) {
    if (!node) return SKINRT_OK;
    int rc = SKINRT_OK;
    for (auto& sh : node->shards) {
        if (sh.crowd) { const int r = skinrt_destroy_crowd(sh.crowd); if (rc == SKINRT_OK) rc = r; }
        if (sh.rig) { const int r = skinrt_destroy_rig(sh.rig); if (rc == SKINRT_OK) rc = r; }
        if (sh.ctx) { const int r = skinrt_destroy_ctx(sh.ctx); if (rc == SKINRT_OK) rc = r; }
    }
    delete node;
    return rc;
}

int skinrt_node_create(const skinrt_config* config, const int32_t* devices, uint32_t device_count, const skinrt_rig_desc* desc,
                       uint32_t instances, skinrt_node** out_node) {
    if (!config || !devices || !desc || !out_node) return fail(SKINRT_E_INVALID, "skinrt_node_create: NULL argument");
    if (config->struct_size != sizeof(skinrt_config)) return fail(SKINRT_E_INVALID, "skinrt_config.struct_size mismatch");
    if (device_count == 0 || device_count > 64) return fail(SKINRT_E_INVALID, "skinrt_node_create: device_count must be 1..64");
    *out_node = nullptr;
    skinrt_node* node = new (std::nothrow) skinrt_node;
    if (!node) return fail(SKINRT_E_NOMEM, "out of host memory");
    node->instances = instances;
    node->M = desc->target_count;
    try { node->shards.resize(device_count); } catch (const std::bad_alloc&) { delete node; return fail(SKINRT_E_NOMEM, "out of host memory"); }
    for (uint32_t g = 0; g < device_count; ++g) {
        skinrt_node::Shard& sh = node->shards[g];
        skinrt_config cfg = *config;
        cfg.device = devices[g];  // the same device may be listed more than once: its shards then share it
        int rc = skinrt_shard_range(instances, g, device_count, &sh.first, &sh.count);
        if (rc == SKINRT_OK) rc = skinrt_init(&cfg, &sh.ctx);
        if (rc == SKINRT_OK) rc = skinrt_upload_rig(sh.ctx, desc, &sh.rig);
        if (rc == SKINRT_OK && sh.count) rc = skinrt_create_crowd(sh.ctx, sh.rig, sh.count, &sh.crowd);
        if (rc != SKINRT_OK) {
            const std::string why = g_err;  // destroy calls below may overwrite it
            (void)skinrt_node_destroy(node);
            return fail(rc, "skinrt_node_create, shard " + std::to_string(g) + ": " + why);
        }
    }
    *out_node = node;
    return SKINRT_OK;
}

int skinrt_node_shard_count(skinrt_node* node, uint32_t* out_count) {
    if (!node || !out_count) return fail(SKINRT_E_INVALID, "skinrt_node_shard_count: NULL argument");
    *out_count = (uint32_t)node->shards.size();
    return SKINRT_OK;
}

int skinrt_node_get_shard(skinrt_node* node, uint32_t shard, skinrt_ctx** out_ctx, skinrt_crowd** out_crowd, uint32_t* out_first,
                          uint32_t* out_count) {
    if (!node) return fail(SKINRT_E_INVALID, "node is NULL");
    if (shard >= node->shards.size()) return fail(SKINRT_E_RANGE, "shard index out of range");
    const skinrt_node::Shard& sh = node->shards[shard];
    if (out_ctx) *out_ctx = sh.ctx;
    if (out_crowd) *out_crowd = sh.crowd;
    if (out_first) *out_first = sh.first;
    if (out_count) *out_count = sh.count;
    return SKINRT_OK;
}

int skinrt_node_set_clocks(skinrt_node* node, const float* tau, const uint32_t* clip_id) {
    if (!node) return fail(SKINRT_E_INVALID, "node is NULL");
    if (!tau) return fail(SKINRT_E_INVALID, "tau is NULL");
    for (auto& sh : node->shards)
        if (sh.crowd)
            if (int rc = skinrt_set_clocks(sh.crowd, tau + sh.first, clip_id ? clip_id + sh.first : nullptr)) return rc;
    return SKINRT_OK;
}

int skinrt_node_set_morph_weights(skinrt_node* node, const float* weights) {
    if (!node) return fail(SKINRT_E_INVALID, "node is NULL");
    for (auto& sh : node->shards)
        if (sh.crowd)
            if (int rc = skinrt_set_morph_weights(sh.crowd, weights ? weights + (size_t)sh.first * node->M : nullptr)) return rc;
    return SKINRT_OK;
}

int skinrt_node_set_root_transforms(skinrt_node* node, const float* matrices) {
    if (!node) return fail(SKINRT_E_INVALID, "node is NULL");
    for (auto& sh : node->shards)
        if (sh.crowd)
            if (int rc = skinrt_set_root_transforms(sh.crowd, matrices ? matrices + (size_t)sh.first * 16 : nullptr)) return rc;
    return SKINRT_OK;
}

int skinrt_node_evaluate(skinrt_node* node, uint32_t stages_mask) {
    if (!node) return fail(SKINRT_E_INVALID, "node is NULL");
    for (auto& sh : node->shards)
        if (sh.crowd)
            if (int rc = skinrt_evaluate(sh.crowd, stages_mask)) return rc;  // enqueues and returns: the devices run side by side
    return SKINRT_OK;
}

int skinrt_node_synchronize(skinrt_node* node) {
    if (!node) return fail(SKINRT_E_INVALID, "node is NULL");
    int rc = SKINRT_OK;
    for (auto& sh : node->shards) {  // wait for every device even after a failure
        const int r = skinrt_synchronize(sh.ctx);
        if (rc == SKINRT_OK) rc = r;
    }
    return rc;
}

int skinrt_node_download_aabbs(skinrt_node* node, float* host_out) {
    if (!node || !host_out) return fail(SKINRT_E_INVALID, "skinrt_node_download_aabbs: NULL argument");
    for (auto& sh : node->shards) {
        if (!sh.crowd) continue;
        float* dev = nullptr;
        uint64_t n = 0;
        if (int rc = skinrt_get_aabbs(sh.crowd, &dev, &n)) return rc;
        if (int rc = skinrt_download(sh.ctx, dev, host_out + (size_t)sh.first * 6, n * 4)) return rc;
    }
    return SKINRT_OK;
}

}  // extern "C"

