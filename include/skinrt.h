/*
 * skinrt.h — C ABI of libskinrt: MI355X-native per-frame skinned-scene evaluation.
 *
 * Drop-in boundary for ONE path of fifth-light/renderer: animation
 * channel sampling -> node world-transform propagation -> joint palette -> morph
 * blending + linear-blend skinning (+ per-instance AABB) over a crowd of independent
 * instances of one rig.  The reference has no FFI for this path (it is plain Rust
 * types); every entry point below names the reference item it stands in for, as
 * `path:line` under /root/reference/crates/.
 *
 * Conventions
 *   - every call returns int: 0 = SKINRT_OK, negative = SKINRT_E_*; nothing aborts or
 *     throws across the boundary; skinrt_last_error() gives the text of the last failure
 *     on the calling thread.
 *   - host input pointers are borrowed for the duration of the call only.
 *   - handles are opaque, owned by the library until the matching destroy call.
 *   - all matrices are column-major float[16], as glam `Mat4::to_cols_array`
 *     (renderer-asset/src/loader/mod.rs:68-75).  Node/root matrices must be affine
 *     (last row 0,0,0,1): glTF requires it and the kernels store 3x4.
 *   - a ctx is bound to one HIP device and one stream; calls on one crowd must not
 *     overlap (the reference calls this path from one render thread, renderer/src/state.rs:236).
 *   - getters return DEVICE pointers valid until the crowd is destroyed; contents are
 *     defined after skinrt_evaluate + skinrt_synchronize.
 */
#ifndef SKINRT_H
#define SKINRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKINRT_ABI_VERSION 7
#define SKINRT_API __attribute__((visibility("default")))

/* ---- error codes ---------------------------------------------------------------- */
#define SKINRT_OK 0
#define SKINRT_E_INVALID (-1)     /* bad argument / malformed rig (reference: panics, loader/gltf/mod.rs:1050,:272) */
#define SKINRT_E_DEVICE (-2)      /* HIP runtime error (no GPU, launch failure, ...) */
#define SKINRT_E_NOMEM (-3)       /* host or device allocation failed */
#define SKINRT_E_NONAFFINE (-4)   /* a node / root matrix has a last row other than 0,0,0,1 */
#define SKINRT_E_RANGE (-5)       /* index out of range (joint, node, clip, instance) */
#define SKINRT_E_LIMIT (-6)       /* reference_limits=1 and joints > 128 (renderer/src/renderer/uniform/skin.rs:10,40,60) */

/* ---- enums ----------------------------------------------------------------------- */
/* AnimationSampler::{Translation,Rotation,Scale} — renderer-asset/src/animation.rs:23-27 */
#define SKINRT_PATH_TRANSLATION 0
#define SKINRT_PATH_ROTATION 1
#define SKINRT_PATH_SCALE 2
/* glTF "weights" (gltf Property::MorphTargetWeights): the reference's loader stops at todo!()
 * (renderer-asset/src/loader/gltf/mod.rs:1312); here the channel is sampled with the same rules as the others
 * (a2-a4, component by component) into the instance's morph weights — see skinrt_channel_desc. */
#define SKINRT_PATH_WEIGHTS 3
/* AnimationKeyFrames::{Linear,Step,CubicSpline} — renderer-asset/src/animation.rs:15-20 */
#define SKINRT_INTERP_LINEAR 0
#define SKINRT_INTERP_STEP 1
#define SKINRT_INTERP_CUBICSPLINE 2

/* stages of skinrt_evaluate (bit mask) */
#define SKINRT_STAGE_POSE 1u      /* sample + propagate + palette (a2-a9) */
#define SKINRT_STAGE_SKIN 2u      /* morph + linear-blend skinning (a10, a13) */
#define SKINRT_STAGE_AABB 4u      /* per-instance bounds of skinned positions (a14); needs SKIN */
#define SKINRT_STAGE_ALL 7u

/* vertex layouts of skinrt_export_interleaved — renderer/src/renderer/buffer/vertex.rs */
#define SKINRT_LAYOUT_COLOR_SKIN 0    /* ColorSkinVertex, 76 B  (:60-69) */
#define SKINRT_LAYOUT_TEXTURE_SKIN 1  /* TextureSkinVertex, 68 B (:82-91) */

#define SKINRT_MAX_JOINTS_REFERENCE 128 /* MAX_JOINTS, renderer/src/renderer/uniform/skin.rs:10 */
#define SKINRT_SKIN_MATRIX_BYTES 112    /* SkinMatrix{[[f32;4];4],[[f32;4];3]}, uniform/skin.rs:12-17 */
#define SKINRT_INSTANCE_UNIFORM_BYTES 124 /* InstanceUniform{[[f32;4];4],[[f32;4];3],[u8;12]}, uniform/transform.rs:8-14 */

/* skinrt_build_flags(): what the loaded library was compiled with.  0 = the product.  Any other bit marks a
 * measurement build whose RESULTS MAY BE WRONG BY DESIGN (kernels_skin.hip ablation switches); skinrt_init refuses
 * to create a context on such a library unless skinrt_config.allow_experimental_build is set. */
#define SKINRT_BUILD_ABLATE 1u   /* -DSKINRT_ABLATE=... : parts of the skin kernels removed */
#define SKINRT_BUILD_EXP 2u      /* -DSKINRT_EXP=...    : experimental variants of the skin kernels */
#define SKINRT_BUILD_STAMPS 4u   /* -DSKINRT_STAMPS     : in-kernel cycle stamps */
#define SKINRT_BUILD_TUNING 8u   /* non-default SKINRT_RANGE2 / SKINRT_MORPH_SLOTS / SKINRT_PALETTE_* compile-time tuning */

typedef struct skinrt_ctx skinrt_ctx;
typedef struct skinrt_rig skinrt_rig;
typedef struct skinrt_crowd skinrt_crowd;

/* ---- configuration ----------------------------------------------------------------
 * Stands in for the reference's env/feature configuration (SURVEY.md §5 "Config"). */
typedef struct skinrt_config {
    uint32_t struct_size;       /* = sizeof(skinrt_config) */
    int32_t device;             /* HIP device ordinal */
    /* 1: Interpolate::cubic_spline exactly as written, t2 = t.powi(3)
     *    (renderer-asset/src/animation.rs:57); 0: glTF-spec Hermite basis (t2 = t*t). */
    uint32_t cubic_quirk;
    /* 1: normal/tangent outputs are normalize()d as the outline shader variants do
     *    (renderer/src/shader/shader.wgsl:217-218); 0: un-normalised (:162-163,:186-187). */
    uint32_t normalize_outputs;
    /* 1: reject rigs the reference cannot render (joints > 128, weight sets > 1). */
    uint32_t reference_limits;
    /* Arithmetic of the skinning kernels.  1: no fused multiply-add, positions / normals / tangents are
     *    bit-equal to the fp32 CPU restatement (oracle/).  2: FMA in the same association — within
     *    1e-5 rel / 1e-6 abs of it on well-conditioned rigs, about 10 % faster.  0 (default): chosen per rig
     *    at upload: FMA unless a joint's inverse bind translation exceeds 4 units in some component —
     *    there the position sums cancel, fused and unfused rounding differ by more than the 1e-6
     *    absolute bar (BASELINE config 5: a 4096-joint chain 400 units long), and positions are computed
     *    contraction-free (bit-equal); normals and tangents, whose 3x3 sums carry no translation and do
     *    not cancel, stay fused and inside the bar.  skinrt_rig_info.exact_arithmetic reports the
     *    choice.  Pose stages are always contraction-free. */
    uint32_t exact_arithmetic;
    /* 0: choose per rig; 1: wavefront-per-instance pose kernel; 2: split pose kernels
     *    (sample / lane-per-instance propagate / palette) used for very deep or large rigs. */
    uint32_t pose_path;
    /* 0: choose per tile; 1: only the loader-wave skinning kernel (the self-staging kernel, used
     *    for tiles with at most 64 joints and rigs with at most 2 weight sets, is switched off). */
    uint32_t skin_path;
    /* Where skinrt_create_crowd puts the position / normal / tangent arrays of a large crowd
     * (3 arrays >= 1 GiB together).  0: one array per third of the device's physical HBM address
     * space where that can be had — on MI355X a write stream confined to one third runs at ~5.4 TB/s, streams
     * spread over the thirds at ~7 TB/s.  hipMalloc gives no say over physical placement and the VRAM manager's
     * habits differ between boxes, so the placement is SEARCHED BY MEASUREMENT: candidate arrays are allocated one
     * after the other and timed together with the arrays already chosen (the skin stage's store pattern, a few ms
     * each); the candidates not taken are freed before the call returns.  Nothing is assumed about the allocator, and
     * results never depend on where the arrays are; skinrt_get_output_placement reports what the final measurement
     * found.  While skinrt_create_crowd runs, the candidates are real allocated memory — up to all of the device's
     * free memory but 3 GiB, or placement_max_filler_gib — so a concurrent allocation of another thread or
     * process may fail, and the call takes up to about a second for a 20 GB crowd.
     * 1: three plain hipMalloc calls, no search, no measurement. */
    uint32_t output_placement;
    /* 1: skinrt_init accepts a library whose skinrt_build_flags() is not 0 (measurement builds). */
    uint32_t allow_experimental_build;
    /* output_placement = 0: the most the search may hold in candidates it did not take, in GiB (0: no limit but the
     * free memory).  When it runs out before a spread placement is found: strategy 3. */
    uint32_t placement_max_filler_gib;
} skinrt_config;

/* ---- flat rig ("loader output", SURVEY.md §8a rows a1,a5-a8,a11,a13) ---------------- */

/* One AnimationChannelAsset{sampler,length,target_id} — renderer-asset/src/animation.rs:29-34,
 * keyframes flattened to SoA (time[], value[]) from AnimationKeyFrame<T> (:8-12). */
typedef struct skinrt_channel_desc {
    uint32_t target_node;    /* flat node index of target_id's node */
    uint8_t path;            /* SKINRT_PATH_* */
    uint8_t interpolation;   /* SKINRT_INTERP_* */
    uint16_t reserved;
    uint32_t key_count;      /* >= 1 */
    const float* times;      /* [key_count], non-decreasing */
    /* width w = 4 for rotation (x,y,z,w), 3 for translation / scale, the rig's target_count for weights.
     * LINEAR/STEP: [key_count][w].  CUBICSPLINE: [key_count][3][w] = (in, value, out),
     * the order of AnimationKeyFrames::CubicSpline (animation.rs:18-19).
     * SKINRT_PATH_WEIGHTS: width = the target_count of the mesh section whose node is target_node (at most one such
     * channel per clip and section; with no sections[] the rig is one section and target_node is not looked at).
     * The channel does not turn a Matrix node into TRS form.  An instance playing a clip with a weights channel
     * takes that section's morph weights from the channel's sample at tau, in place of skinrt_set_morph_weights /
     * the rig's defaults, which keep applying to every other instance and section.  The sampled weights are
     * refreshed by SKINRT_STAGE_POSE — and by a SKINRT_STAGE_SKIN that follows skinrt_set_morph_weights /
     * skinrt_set_clocks without a pose stage in between — and consumed by SKINRT_STAGE_SKIN. */
    const float* values;
} skinrt_channel_desc;

/* One AnimationAsset{name,channels} — renderer-asset/src/animation.rs:36-40 */
typedef struct skinrt_clip_desc {
    uint32_t channel_count;
    const skinrt_channel_desc* channels;
} skinrt_clip_desc;

/* One PrimitiveAssetMorphTarget{position,normal,tangent} — renderer-asset/src/primitive.rs:33-38 (the same target
 * of every primitive of a mesh, concatenated).  A target belongs to the section whose [first_target,
 * first_target + target_count) holds it and moves that section's vertices only; rows and indices count from the
 * section's first_vertex.  indices == NULL: dense, count == the section's vertex_count (the loader's form).
 * indices != NULL: compacted form, arrays hold only the touched vertices (strictly increasing indices). */
typedef struct skinrt_morph_target_desc {
    uint32_t count;
    const uint32_t* indices;
    const float* position;   /* [count][3], NULL = zeros (loader: empty Vec) */
    const float* normal;     /* [count][3], NULL = zeros */
    const float* tangent;    /* [count][tangent_stride], NULL = zeros; only xyz is used */
} skinrt_morph_target_desc;

/* One SkinAsset{inverse_bind_matrices,joint_ids} — renderer-asset/src/skin.rs:5-11.  The loader caches skins per
 * glTF skin index and nodes share them by Arc (loader/gltf/mod.rs:1124-1166, node.rs:95): an avatar normally has
 * several skins (body / face / hair) over one node tree. */
typedef struct skinrt_skin_desc {
    uint32_t joint_count;
    const uint32_t* joint_node;  /* [joint_count] flat node index of each joint */
    const float* inverse_bind;   /* [joint_count][16] or NULL = identity (loader/gltf/mod.rs:1139-1145) */
} skinrt_skin_desc;

/* One mesh instance: a NodeAsset with mesh: Some(..) (renderer-asset/src/node.rs:89-99), all primitives of its
 * MeshAsset concatenated (mesh.rs:4-8).  skin >= 0: NodeAsset.skin, the section's JOINTS_n index that skin's joints
 * and its node transform is ignored, as the skin shaders do (shader.wgsl:154-165).  skin = -1: NodeAsset.skin == None,
 * an un-skinned mesh: its vertices are placed by the node's world matrix exactly as the non-skin shaders place them,
 * instance.transform * vec4(position, 1), instance.normal * normal / tangent (shader.wgsl:143-152, :167-176), with
 * instance.normal = Mat3(inverse(transform)^T) (uniform/transform.rs:27-34); joints[] / weights[] are not read for
 * its vertices.  Sections tile [0, vertex_count) and [0, target_count) in order, without gaps. */
typedef struct skinrt_section_desc {
    uint32_t node;            /* flat node index of the NodeAsset that carries the mesh */
    int32_t skin;             /* index into skins[], or -1 */
    uint32_t first_vertex, vertex_count;
    uint32_t first_target, target_count;  /* the mesh's morph targets (MeshAsset.weights.len()) */
} skinrt_section_desc;

typedef struct skinrt_rig_desc {
    uint32_t struct_size;    /* = sizeof(skinrt_rig_desc) */

    /* nodes: NodeAsset tree (renderer-asset/src/node.rs:89-99) flattened DFS pre-order over
     * SceneAsset.nodes then children (loader/gltf/mod.rs:1168-1217); parent[n] < n, -1 = root. */
    uint32_t node_count;
    const int32_t* parent;       /* [N] */
    const float* translation;    /* [N][3]  DecomposedTransform (node.rs:13-17) */
    const float* rotation;       /* [N][4]  x,y,z,w — used un-normalised, as glam does */
    const float* scale;          /* [N][3] */
    const float* matrix;         /* [N][16] or NULL; NodeTransform::Matrix (node.rs:29-33) */
    const uint8_t* has_matrix;   /* [N] or NULL; 1 = node uses matrix[n] */

    /* skin: SkinAsset{inverse_bind_matrices,joint_ids} — renderer-asset/src/skin.rs:5-11.  The single-skin
     * short form: read only when skin_count == 0 below (then this is skin 0). */
    uint32_t joint_count;
    const uint32_t* joint_node;  /* [J] flat node index of each joint */
    const float* inverse_bind;   /* [J][16] or NULL = identity (loader/gltf/mod.rs:1139-1145) */

    /* animations: Vec<AnimationAsset> */
    uint32_t clip_count;
    const skinrt_clip_desc* clips;

    /* mesh: PrimitiveAssetAttributes — renderer-asset/src/primitive.rs:22-31 (all primitives of
     * the skinned mesh concatenated). */
    uint32_t vertex_count;
    const float* position;       /* [V][3] */
    const float* normal;         /* [V][3] */
    const float* tangent;        /* [V][tangent_stride]; only xyz is used (shader.wgsl:90) */
    uint32_t tangent_stride;     /* 3 or 4 (asset side stores [f32;4], primitive.rs:26) */
    uint32_t set_count;          /* K >= 1 JOINTS_n/WEIGHTS_n sets; the reference shader reads set 0 */
    const uint16_t* const* joints;  /* [K] -> [V][4]  Joints = Vec<[u16;4]> (primitive.rs:19) */
    const float* const* weights;    /* [K] -> [V][4]  Weights = Vec<[f32;4]> (primitive.rs:20) */

    /* morph targets + MeshAsset.weights (renderer-asset/src/mesh.rs:7) */
    uint32_t target_count;
    const skinrt_morph_target_desc* targets;
    const float* default_weights;   /* [M] or NULL = zeros */

    /* several skins on one node tree, several meshes (ABI 6).  skin_count == 0: one skin, the short form above.
     * section_count == 0: one section = all vertices, all targets, skin 0 (or, without joints, nothing to skin).
     * With sections, joints[k][v] index the joints of the vertex's OWN skin, as the loader delivers them. */
    uint32_t skin_count;
    const skinrt_skin_desc* skins;
    uint32_t section_count;
    const skinrt_section_desc* sections;
} skinrt_rig_desc;

/* ---- rig introspection (what upload compiled; used by tests and the bench roofline) ------- */
typedef struct skinrt_rig_info {
    uint32_t node_count, joint_count, vertex_count, set_count, target_count, clip_count;
    uint32_t level_count;          /* depth levels of the node forest */
    uint32_t tile_count;           /* vertex tiles */
    uint32_t tile_vertices;        /* vertices per tile */
    uint32_t max_tile_joints;      /* largest tile-local joint list */
    uint32_t lds_tiles;            /* tiles whose joint list is LDS-staged (others gather from L2) */
    uint32_t max_channels;         /* Cmax */
    uint64_t morph_entries;        /* nonzero (vertex,target) pairs kept after compaction */
    uint64_t device_bytes;         /* bytes of device memory the rig occupies */
    uint32_t exact_arithmetic;     /* skin stage on this rig: 0 fused, 1 contraction-free, 2 positions contraction-free and normals / tangents fused (skinrt_config.exact_arithmetic) */
    float bind_extent;             /* largest |translation component| over the inverse bind matrices */
    uint32_t weight_channel_clips; /* clips that carry a SKINRT_PATH_WEIGHTS channel */
    /* 1: enough 64-vertex waves of the mesh have all-zero weights in the second weight set that the skin stage runs
     *    the kernels which let such a wave stop after the first set (same results: the skipped terms are 0 * M) */
    uint32_t skip_weight_sets;
    /* ABI 6.  The palette holds every skin's joints one skin after the other, then one entry per distinct node of an
     * un-skinned section (joint matrix = the node's world matrix): palette_count = sum of joint counts + those. */
    uint32_t skin_count, section_count, palette_count;
    /* tiles (of tile_count) whose joint list is longer than the lists the main skin kernel's LDS ring is sized for (at
     * most 2 % of the tiles: typically those that straddle two mesh sections with different skins): served whole by the kernel that otherwise takes the instances
     * with many active morph pairs */
    uint32_t handed_over_tiles;
} skinrt_rig_info;

/* Where a skin / a section lives in the crowd's arrays (skinrt_rig_skin_layout, skinrt_rig_section_layout). */
typedef struct skinrt_skin_layout {
    uint32_t first_palette_entry, joint_count;
} skinrt_skin_layout;
typedef struct skinrt_section_layout {
    uint32_t node;
    int32_t skin;
    uint32_t first_vertex, vertex_count, first_target, target_count;
    uint32_t first_palette_entry;   /* skinned: its skin's; un-skinned: the entry holding the node's world matrix */
    uint32_t reserved;
} skinrt_section_layout;

/* What skinrt_create_crowd did with the output arrays and what it measured (all zero for small crowds
 * and for output_placement=1). */
typedef struct skinrt_placement_info {
    /* 0: plain allocations (small crowd, output_placement = 1, not an MI355X);
     * 1: placed by the search and the final measurement confirms it (spread_gain >= 1.15);
     * 2: searched, but no spread placement was found (spread_gain < 1.15: one region reachable, or a device without
     *    such regions); the arrays are used where they are, the stage then writes at the one-region rate;
     * 3: the search ran out of room (free memory, placement_max_filler_gib) before it found one. */
    uint32_t strategy;
    uint32_t reserved;
    uint64_t filler_bytes;      /* transient reservation during skinrt_create_crowd (candidates not taken) */
    float probe_together_ms;    /* the skin stage's store pattern, no arithmetic, on all three arrays */
    float probe_alone_ms;       /* the same on the position array alone */
    float spread_gain;          /* 3 * alone / together: ~1.0 = one region, ~1.25-1.3 = spread */
    float probe_GBps;           /* bytes of the three arrays / probe_together_ms */
} skinrt_placement_info;

/* ---- lifecycle --------------------------------------------------------------------- */

/* Library/ABI version (SKINRT_ABI_VERSION). No reference counterpart. */
SKINRT_API int skinrt_abi_version(void);
/* SKINRT_BUILD_* bits the library was compiled with; 0 for the product. No reference counterpart. */
SKINRT_API uint32_t skinrt_build_flags(void);

/* Text of the last error raised on the calling thread ("" if none).
 * Stands in for GltfLoaderError's Display impl (loader/gltf/mod.rs:78-173). */
SKINRT_API const char* skinrt_last_error(void);

/* Create a context on config->device with its own HIP stream.
 * Replaces wgpu device/queue acquisition for this path (renderer/src/state.rs:94-127).
 *
 * Environment.  skinrt_init reads the variables below ONCE (nothing is read per frame); they move work between
 * streams / blocks / launch orders for measurements (tools/, profiles/) and never change a result.  Unset = the
 * product's choice, which is what every number in DESIGN.md is measured with.
 *   SKINRT_FIX_STREAM  0: the skin stage's kernels one after the other on the ctx stream (default 1: k_skin_fix beside k_skin2)
 *   SKINRT_FIX_PRIO    priority of that side stream: -1 most urgent, 0 default (default), 1 least urgent
 *   SKINRT_FIX_ORDER   1: enqueue k_skin2 before k_skin_fix (default 0)
 *   SKINRT_FIX_BLOCKS  about this many k_skin_fix blocks (default 32768)
 *   SKINRT_POSE_PRE    0: sample channels inside the pose kernels, no node-major scratch (default 1)
 *   SKINRT_TILE_CHUNKS 1 / 2 / 4 / 8 tile chunks over the XCDs (default: chosen per rig)
 *   SKINRT_RANGE2      instances per k_skin2 block, at most 80 (default: chosen per crowd)
 *   SKINRT_TILE_RUN    tiles per run dealt to an XCD (default: chosen per crowd)
 *   SKINRT_SKIP_SETS   1 / 0: force the kernels that stop a wave after its last used weight set on / off
 *   SKINRT_LDS_PAD     bytes of unused LDS added to every k_skin2 block, at most 32768 (occupancy experiments) */
SKINRT_API int skinrt_init(const skinrt_config* config, skinrt_ctx** out_ctx);
SKINRT_API int skinrt_destroy_ctx(skinrt_ctx* ctx);

/* Validate, compile (vertex tiles, tile-local joint lists, per-vertex morph lists,
 * depth levels) and upload a rig.  Replaces VertexBuffer::new (renderer/buffer/vertex.rs:110-121)
 * and the one-time walk of (Vec<SceneAsset>, Vec<AnimationAsset>) (loader/gltf/mod.rs:199). */
SKINRT_API int skinrt_upload_rig(skinrt_ctx* ctx, const skinrt_rig_desc* desc, skinrt_rig** out_rig);
/* Validate and compile a rig on the host only (no device needed); fills out_info (may be NULL).
 * Same error codes as skinrt_upload_rig.  config may be NULL (defaults). */
SKINRT_API int skinrt_validate_rig(const skinrt_config* config, const skinrt_rig_desc* desc, skinrt_rig_info* out_info);
SKINRT_API int skinrt_destroy_rig(skinrt_rig* rig);
/* What skinrt_upload_rig compiled for this rig. No reference counterpart. */
SKINRT_API int skinrt_get_rig_info(skinrt_rig* rig, skinrt_rig_info* out_info);
/* Where skin `skin` / section `section` of the rig lie in the palette, vertex and morph-weight arrays
 * (SkinAsset: skin.rs:5-11; NodeAsset.mesh: node.rs:96).  SKINRT_E_RANGE beyond the rig's counts. */
SKINRT_API int skinrt_rig_skin_layout(skinrt_rig* rig, uint32_t skin, skinrt_skin_layout* out);
SKINRT_API int skinrt_rig_section_layout(skinrt_rig* rig, uint32_t section, skinrt_section_layout* out);

/* Allocate per-instance state and output buffers for `instances` copies of `rig`.
 * Replaces one SkinUniformBuffer::new per entity (renderer/uniform/skin.rs:39-57). */
SKINRT_API int skinrt_create_crowd(skinrt_ctx* ctx, skinrt_rig* rig, uint32_t instances, skinrt_crowd** out_crowd);
SKINRT_API int skinrt_destroy_crowd(skinrt_crowd* crowd);

/* Output-array placement chosen and measured at skinrt_create_crowd. No reference counterpart. */
SKINRT_API int skinrt_get_output_placement(skinrt_crowd* crowd, skinrt_placement_info* out_info);

/* Instance range [*out_first, *out_first + *out_count) of rank `rank` of `world` when `instances` are sharded
 * over processes / GPUs (one ctx and one crowd per GPU, the rig replicated, no data-path collective: SURVEY.md
 * §8e): contiguous ranges of ceil(instances / world), the last ones shorter or empty.  Pure function, no device.
 * The same rule as renderer_amd/shard.py, for hosts that are not Python. No reference counterpart. */
SKINRT_API int skinrt_shard_range(uint32_t instances, uint32_t rank, uint32_t world, uint32_t* out_first, uint32_t* out_count);

/* ---- per-frame inputs ----------------------------------------------------------------- */

/* Per-instance animation clock tau[I] (seconds, already wrapped by the caller) and clip
 * index clip_id[I] (NULL = clip 0; UINT32_MAX = rest pose).  Keyframe search rule: SURVEY.md §8a a4. */
SKINRT_API int skinrt_set_clocks(skinrt_crowd* crowd, const float* tau, const uint32_t* clip_id);
/* The same without blocking the host (a frame loop that never drains the GPU; renderer/src/state.rs:236-330 is one
 * thread that would otherwise stall here every frame).  tau / clip_id are DEVICE pointers, or host memory the GPU
 * can read (hipHostMalloc), and must stay unchanged until the copy — enqueued on the ctx stream, in order with the
 * evaluates around it — has run.  clip_id NULL keeps the current clips.  Nothing is validated on the host: a clip id
 * >= clip_count plays the rest pose, like UINT32_MAX.  No reference counterpart. */
SKINRT_API int skinrt_set_clocks_async(skinrt_crowd* crowd, const float* tau, const uint32_t* clip_id);
/* Advance every instance's clock on the device, stream-ordered, no host data:
 *   tau[i] = fmodf(tau[i] + dt * rate[i], clip_length[clip[i]])      (two roundings, then the exact remainder)
 * — the wrap SURVEY.md §8a a4 leaves to the caller — with clip_length = the largest key time of the clip
 * (AnimationChannelAsset.length, loader/gltf/mod.rs:1243-1247); dev_rate is a DEVICE array [I] or NULL = 1.0;
 * instances at rest (clip id out of range) and clips of length 0 keep their clock.  No reference counterpart. */
SKINRT_API int skinrt_advance_clocks(skinrt_crowd* crowd, float dt, const float* dev_rate);
/* The clocks as they are now, [I] f32 device pointer (skinrt_download copies them out). */
SKINRT_API int skinrt_get_clocks(skinrt_crowd* crowd, float** out_dev, uint64_t* out_count);
/* Per-instance morph weights w[I][M] (NodeAsset.weights override, node.rs:98); NULL restores
 * the rig's default weights for every instance. */
SKINRT_API int skinrt_set_morph_weights(skinrt_crowd* crowd, const float* weights);
/* Per-instance root placement m[I][16] (pre-multiplies every scene root); NULL = none.
 * Replaces InstanceUniformBuffer.transform for crowd placement (renderer/uniform/transform.rs:17-20). */
SKINRT_API int skinrt_set_root_transforms(skinrt_crowd* crowd, const float* matrices);

/* ---- evaluation ------------------------------------------------------------------------ */

/* Enqueue the selected stages on the ctx stream (asynchronous).  This is the body the
 * reference leaves empty at Entity::render (renderer/src/client/entity/mod.rs:34). */
SKINRT_API int skinrt_evaluate(skinrt_crowd* crowd, uint32_t stages_mask);
/* Block until everything enqueued on the ctx stream has finished. */
SKINRT_API int skinrt_synchronize(skinrt_ctx* ctx);
/* Device time of the last evaluate's kernels in ms (HIP events on the ctx stream):
 * ms[0]=pose, ms[1]=skin+morph, ms[2]=aabb reduce, ms[3]=total span.  Synchronises. */
SKINRT_API int skinrt_last_stage_ms(skinrt_crowd* crowd, float ms[4]);
/* The ctx's hipStream_t, for callers that order their own work against it. */
SKINRT_API int skinrt_get_stream(skinrt_ctx* ctx, void** out_stream);

/* ---- outputs (device pointers) --------------------------------------------------------- */
/* Skinned world-space positions / normals / tangents, each [I][V][3] f32
 * (shader.wgsl:159-163,:223-226: skin_matrix * vec4(position,1), normal_matrix * normal/tangent). */
SKINRT_API int skinrt_get_positions(skinrt_crowd* crowd, float** out_dev, uint64_t* out_count);
SKINRT_API int skinrt_get_normals(skinrt_crowd* crowd, float** out_dev, uint64_t* out_count);
SKINRT_API int skinrt_get_tangents(skinrt_crowd* crowd, float** out_dev, uint64_t* out_count);
/* Per-instance bounds [I][2][3] f32 (min xyz, max xyz).  No reference counterpart (a14). */
SKINRT_API int skinrt_get_aabbs(skinrt_crowd* crowd, float** out_dev, uint64_t* out_count);
/* Joint palette [I][P][24] f32, P = skinrt_rig_info.palette_count (one skin: J): 3 rows x 4 of the joint matrix (row r = m[r][0..3] incl.
 * translation), then the 3 x 3 normal matrix row by row (9 floats, packed), then 3 floats of zero padding.
 * Same numbers as SkinMatrix (uniform/skin.rs:12-17), transposed to rows; see
 * skinrt_export_skin_uniform for the reference byte layout.
 * Sign of zero: the values compare equal (==) to glam's, but a zero in the normal rows may carry the other sign.
 * glam inverts the full 4x4 (uniform/skin.rs:33), whose cofactor sums include "+ 0 * x" terms that turn a -0.0 into
 * +0.0; the kernels invert the affine 3x4 and have no such terms.  (Seen on rigs whose rest rotations are all
 * identity.)  Nothing downstream of a zero normal-matrix entry depends on its sign except another zero's sign. */
SKINRT_API int skinrt_get_palette(skinrt_crowd* crowd, float** out_dev, uint64_t* out_count);
/* Node world matrices [I][N][12] f32, 3 rows x 4 (the Mat4's first three rows). */
SKINRT_API int skinrt_get_world(skinrt_crowd* crowd, float** out_dev, uint64_t* out_count);
/* Selected keyframe per (instance, channel of the instance's clip): [I][Cmax] u32, Cmax =
 * largest clip channel count; entries past the clip's channel count are UINT32_MAX. */
SKINRT_API int skinrt_get_keyframe_index(skinrt_crowd* crowd, uint32_t** out_dev, uint64_t* out_count);

/* Copy `bytes` from a device pointer returned above to host memory (synchronises the ctx). */
SKINRT_API int skinrt_download(skinrt_ctx* ctx, const void* dev_src, void* host_dst, uint64_t bytes);

/* ---- reference-layout exports (host destinations) ---------------------------------------- */
/* One instance's palette as the reference uploads it: 128 x SkinMatrix (112 B each: mat4
 * column-major, then 3 x vec4 normal columns padded with 0), entries >= J zero
 * (uniform/skin.rs:41-50; shader.wgsl:70-77).  Requires J <= 128. dst: 14336 bytes. */
SKINRT_API int skinrt_export_skin_uniform(skinrt_crowd* crowd, uint32_t instance, void* dst);
/* One instance's vertices in ColorSkinVertex (76 B) or TextureSkinVertex (68 B) layout
 * (buffer/vertex.rs:60-102): position/normal/tangent = skinned outputs; joint_index /
 * joint_weight = the rig's set-0 indices/weights passed through unchanged; color (4 f32) or
 * tex_coords (2 f32) are copied from `attr` ([V][4] or [V][2], NULL = zeros). dst: V*stride bytes. */
SKINRT_API int skinrt_export_interleaved(skinrt_crowd* crowd, uint32_t instance, uint32_t layout,
                              const float* attr, void* dst);

/* The two exports above for one skin / one section of a rig with several (ABI 6): `skin`'s joints as
 * 128 x SkinMatrix; `section`'s vertices (section.vertex_count records), joint_index / joint_weight = the
 * section's own set-0 indices (into its skin's joints) and weights.  SKINRT_E_INVALID for an un-skinned section:
 * the reference draws those from the static mesh and an InstanceUniform (below). */
SKINRT_API int skinrt_export_skin_uniform_of(skinrt_crowd* crowd, uint32_t skin, uint32_t instance, void* dst);
SKINRT_API int skinrt_export_interleaved_section(skinrt_crowd* crowd, uint32_t section, uint32_t instance, uint32_t layout,
                                                 const float* attr, void* dst);
/* InstanceUniform{transform: [[f32;4];4], normal: [[f32;4];3], padding: [u8;12]} of node `node` of instance
 * `instance` — what InstanceUniformBuffer::update writes (renderer/src/renderer/uniform/transform.rs:8-14, :51-59)
 * with transform = the node's world matrix, normal = Mat3::from_mat4(transform.inverse().transpose()) (:27-34), the
 * uniform the non-skin shaders read (shader.wgsl:56-61, :143-152).  dst: 124 bytes.  Zero signs as at
 * skinrt_get_palette. */
SKINRT_API int skinrt_export_instance_uniform(skinrt_crowd* crowd, uint32_t instance, uint32_t node, void* dst);

/* ---- reference-layout exports on the device (whole instance ranges, device destinations) ------ */
/* The two exports above for instances [first, first + count) in ONE kernel each, written straight from the
 * crowd's output arrays to device memory the caller owns (what a render core would bind as vertex / uniform
 * buffers: VertexBuffer::new, buffer/vertex.rs:110-121; SkinUniformBuffer::update, uniform/skin.rs:59-71).
 * Asynchronous on the ctx stream, ordered after earlier skinrt_evaluate calls; dev_dst must be 16-byte
 * aligned.  Bytes are identical to `count` calls of the host exports.
 *   skin uniform: dev_dst receives count * 128 SkinMatrix = count * 14336 bytes; requires J <= 128.
 *   interleaved : dev_dst receives count * V records of 76 / 68 bytes, instance after instance with no
 *                 padding; dev_attr is a DEVICE array [V][4] (color) or [V][2] (tex_coords), NULL = zeros. */
SKINRT_API int skinrt_export_skin_uniform_device(skinrt_crowd* crowd, uint32_t first, uint32_t count, void* dev_dst);
SKINRT_API int skinrt_export_interleaved_device(skinrt_crowd* crowd, uint32_t first, uint32_t count, uint32_t layout,
                                     const float* dev_attr, void* dev_dst);
/* The same for one skin / one section (see skinrt_export_skin_uniform_of, skinrt_export_interleaved_section);
 * dev_attr is [section.vertex_count][4 or 2]. */
SKINRT_API int skinrt_export_skin_uniform_of_device(skinrt_crowd* crowd, uint32_t skin, uint32_t first, uint32_t count, void* dev_dst);
SKINRT_API int skinrt_export_interleaved_section_device(skinrt_crowd* crowd, uint32_t section, uint32_t first, uint32_t count,
                                                        uint32_t layout, const float* dev_attr, void* dev_dst);
/* The palette of skin `skin` for a STORAGE-buffer binding: `slots` SkinMatrix records (112 B) per instance instead of
 * the uniform array's 128, slots >= the skin's joint count (SKINRT_E_LIMIT otherwise), records >= the joint count
 * zero; dev_dst receives count * slots * 112 bytes.  With slots = 128 the bytes are those of
 * skinrt_export_skin_uniform_of_device.  This is what lifts MAX_JOINTS = 128 (uniform/skin.rs:10, shader.wgsl:70:
 * `array<JointItem, 128>` in a uniform binding) once the render core declares the binding
 * `var<storage, read> joints: array<JointItem>` (renderer/mod.rs:236-248 -> BufferBindingType::Storage). */
SKINRT_API int skinrt_export_skin_storage_device(skinrt_crowd* crowd, uint32_t skin, uint32_t first, uint32_t count, uint32_t slots,
                                                 void* dev_dst);
/* InstanceUniform records for instances [first, first + count) x the listed nodes, one kernel:
 * record (k, n) at dev_dst + (k * node_count + n) * stride, stride >= 124 and a multiple of 4 (256 suits wgpu's
 * dynamic uniform offsets); bytes 124..stride of a record are not written.  dev_nodes: DEVICE array
 * [node_count] of flat node indices (indices >= the rig's node count write an all-zero record).  dev_dst 4-byte
 * aligned.  Same bytes as skinrt_export_instance_uniform. */
SKINRT_API int skinrt_export_instance_uniform_device(skinrt_crowd* crowd, uint32_t first, uint32_t count, const uint32_t* dev_nodes,
                                                     uint32_t node_count, uint32_t stride, void* dev_dst);

/* ---- load-time geometry ------------------------------------------------ */
/* calculate_tangent (renderer-asset/src/tangent.rs:15-58) on the device: per-vertex sum of the unit
 * face normals normalize((p2 - p1) x (p0 - p1)) of the triangles that touch the vertex, added in the
 * reference's order (triangles ascending, corners 0,1,2), then normalised — bit-equal to the
 * sequential loop, not an atomic scatter-add.  The reference calls it for primitives without TANGENT
 * (loader/gltf/mod.rs:1086-1088; its result also stands in for a missing NORMAL, :1089) and for morph
 * targets without one (:911-916, on the delta positions).
 * mode: PrimitiveAssetMode ordinal (0 Points, 1 LineStrip, 2 LineList, 3 TriangleStrip, 4 TriangleList);
 * 3 is SKINRT_E_INVALID (todo!() in the reference, tangent.rs:27); 0-2 have no faces, every vertex
 * normalises a zero sum (NaN, as the reference).  positions: host [V][3]; indices: host
 * [index_count] or NULL (then vertices 3f, 3f+1, 3f+2 form triangle f); an index >= V is
 * SKINRT_E_INVALID (reference: panic).  out: host [V][3].  out_device_ms (may be NULL): time of the
 * kernels alone (HIP events), without the transfers.  A vertex named by more than SKINRT_TANGENT_MAX_VALENCE
 * triangle corners is SKINRT_E_LIMIT (each vertex's corners are sorted on the device: a few by insertion, up to
 * that many by heapsort). */
#define SKINRT_TANGENT_MAX_VALENCE 49152u
SKINRT_API int skinrt_calculate_tangent(skinrt_ctx* ctx, uint32_t mode, const float* positions, uint32_t vertex_count,
                             const uint32_t* indices, uint32_t index_count, float* out, float* out_device_ms);

/* The same computation for callers that must not allocate device memory (code under HIP graph capture,
 * serving loops and harnesses that forbid hipMalloc): NO COUNTERPART IN THE REFERENCE.  positions,
 * indices and out are DEVICE pointers the caller owns; the work area (corner buckets, face normals,
 * scan sums) is a 16 MiB array inside the library's code object, one copy per GPU, held until the
 * process ends.  Limits: vertex_count <= SKINRT_TANGENT_NOALLOC_MAX_VERTICES and at most
 * SKINRT_TANGENT_NOALLOC_MAX_INDICES corners, else SKINRT_E_LIMIT.  The call enqueues kernels and one
 * hipMemsetAsync on the ctx stream and returns; it calls no hipMalloc / hipFree and does not
 * synchronise (out is defined after skinrt_synchronize).  Every call on a GPU shares that GPU's work
 * area: calls from several ctx on one GPU are ordered one behind the other on the device (an event made at
 * skinrt_init; the host does not wait), EXCEPT on a stream under capture, where no event from outside the graph
 * can be waited for — there, and for anything else that shares the GPU's copy, overlap is the caller's to avoid.
 * A vertex named by more than SKINRT_TANGENT_MAX_VALENCE corners (which cannot be seen from the host) gets NaN.  The indices
 * cannot be inspected on the host: a triangle with an index >= vertex_count contributes nothing
 * (skinrt_calculate_tangent rejects such input with SKINRT_E_INVALID instead).  Same results, bit for bit,
 * as skinrt_calculate_tangent on valid input. */
#define SKINRT_TANGENT_NOALLOC_MAX_VERTICES 262144u
#define SKINRT_TANGENT_NOALLOC_MAX_INDICES 1572864u
SKINRT_API int skinrt_calculate_tangent_noalloc(skinrt_ctx* ctx, uint32_t mode, const float* dev_positions, uint32_t vertex_count,
                                     const uint32_t* dev_indices, uint32_t index_count, float* dev_out);

/* ---- per-frame normals from the DEFORMED positions ("per-frame normal recompute after morphing") ----
 * The reference generates normals / "tangents" once, at load, from the bind-pose positions (calculate_tangent,
 * tangent.rs:15-58; loader/gltf/mod.rs:1086-1089) and only transforms them afterwards; under strong morphs (a face) or
 * where the asset had no normals at all, a renderer wants the same generator run on what the frame produced.  A topology
 * is a mesh's triangle list prepared for that, once: vertex_count = the RIG's vertex count (all sections; a section's
 * indices are offset by its first vertex, skinrt_rig_section_layout), indices host [index_count] (whole triangles are
 * used: index_count / 3 of them, tangent.rs:29) or NULL (vertices 3f, 3f+1, 3f+2 form triangle f).
 * An index >= vertex_count is SKINRT_E_INVALID, a vertex named by more than SKINRT_TANGENT_MAX_VALENCE corners
 * SKINRT_E_LIMIT.  The call blocks until the tables are built (the index list is borrowed for the call only). */
typedef struct skinrt_topology skinrt_topology;
SKINRT_API int skinrt_create_topology(skinrt_ctx* ctx, uint32_t vertex_count, const uint32_t* indices, uint32_t index_count,
                                      skinrt_topology** out_topology);
SKINRT_API int skinrt_destroy_topology(skinrt_topology* topology);  /* waits for the ctx stream; NULL is a no-op */
/* For instances [first, first + count): output[i][v] = calculate_tangent(TriangleList, positions[i], indices)[v] — bit for
 * bit what skinrt_calculate_tangent returns for that instance's skinned positions (vertices no triangle touches and
 * degenerate sums give NaN, as the reference) — written over the instance's normals (SKINRT_OUTPUT_NORMALS) or tangents
 * (SKINRT_OUTPUT_TANGENTS; the reference's loader uses the same vector for both when an asset has neither,
 * loader/gltf/mod.rs:1086-1089).  Enqueued on the ctx stream behind the evaluate that produced the positions; does not
 * synchronise.  One kernel, thread = one vertex of eight consecutive instances: the face normals a vertex needs are
 * recomputed from each instance's positions in the reference's order, no per-instance face table exists. */
#define SKINRT_OUTPUT_NORMALS 0u
#define SKINRT_OUTPUT_TANGENTS 1u
SKINRT_API int skinrt_recompute_normals(skinrt_crowd* crowd, skinrt_topology* topology, uint32_t first, uint32_t count, uint32_t output);

/* ---- one crowd over several GPUs of this process ------------------------------------
 * No reference counterpart (the reference renders on one adapter).  A node holds one ctx, one copy of the rig and
 * one crowd per listed device; instance i lives on the shard skinrt_shard_range assigns it to, at row i - first.
 * There is no exchange step on the data path.  All calls come from one host thread: skinrt_node_evaluate enqueues
 * the stages on every device's stream in turn and returns, so the devices run side by side;
 * skinrt_node_synchronize waits for all of them.  A device may be listed more than once (its shards then share
 * it).  Error behaviour as the single-device calls': the first failing shard's code, skinrt_last_error names it. */
typedef struct skinrt_node skinrt_node;

/* config->device is ignored (devices[] takes its place).  Fails as a whole — nothing is left allocated — when any
 * shard's skinrt_init / skinrt_upload_rig / skinrt_create_crowd fails. */
SKINRT_API int skinrt_node_create(const skinrt_config* config, const int32_t* devices, uint32_t device_count,
                                  const skinrt_rig_desc* desc, uint32_t instances, skinrt_node** out_node);
SKINRT_API int skinrt_node_destroy(skinrt_node* node);
SKINRT_API int skinrt_node_shard_count(skinrt_node* node, uint32_t* out_count);
/* The shard's handles, for every per-device call above (skinrt_get_positions, the exporters, skinrt_last_stage_ms,
 * skinrt_get_stream ...).  *out_crowd is NULL for an empty shard (more devices than instances).  Any out pointer
 * may be NULL.  The handles belong to the node. */
SKINRT_API int skinrt_node_get_shard(skinrt_node* node, uint32_t shard, skinrt_ctx** out_ctx, skinrt_crowd** out_crowd,
                                     uint32_t* out_first, uint32_t* out_count);
/* Whole-crowd host arrays, [instances] / [instances][target_count] / [instances][16]: each shard is handed its rows
 * (skinrt_set_clocks, skinrt_set_morph_weights, skinrt_set_root_transforms; NULL means what it means there). */
SKINRT_API int skinrt_node_set_clocks(skinrt_node* node, const float* tau, const uint32_t* clip_id);
SKINRT_API int skinrt_node_set_morph_weights(skinrt_node* node, const float* weights);
SKINRT_API int skinrt_node_set_root_transforms(skinrt_node* node, const float* matrices);
SKINRT_API int skinrt_node_evaluate(skinrt_node* node, uint32_t stages_mask);
SKINRT_API int skinrt_node_synchronize(skinrt_node* node);
/* The per-frame gather a renderer needs for culling: every shard's per-instance bounds into host_out[instances][6]
 * (min xyz, max xyz), in instance order.  Waits for the shards' streams. */
SKINRT_API int skinrt_node_download_aabbs(skinrt_node* node, float* host_out);

#ifdef __cplusplus
}
#endif
#endif /* SKINRT_H */
