/* host_capi.h — C entry points of libskinrt_host.so: the C++ host side (glb loader + scene
 * flattener) made callable from the Python tests and tools.  This library sits ABOVE the drop-in
 * boundary (include/skinrt.h): it produces the skinrt_rig_desc that skinrt_upload_rig takes. */
#ifndef SKINRT_HOST_CAPI_H
#define SKINRT_HOST_CAPI_H
#include <stddef.h>
#include <stdint.h>

#include "../../include/skinrt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct skinrt_host_rig skinrt_host_rig;

/* load_glb_from_buffer (renderer-asset/src/loader/gltf/mod.rs:1358-1381) followed by the flatten
 * walk (flatten.hpp).  scene = -1: all scenes.  Returns SKINRT_OK or SKINRT_E_*. */
SKINRT_API int skinrt_host_load_glb(const uint8_t* data, size_t size, int scene, skinrt_host_rig** out);
/* PMX 2.0 model -> flat rig (pmx_loader.hpp: the reference parses bones, weights and morphs and drops
 * them, loader/pmx/mod.rs:200-262; the mapping onto the rig is build-defined).  Node n of the rig is
 * PMX bone skinrt_host_rig_node_gltf_index(rig, n). */
SKINRT_API int skinrt_host_load_pmx(const uint8_t* data, size_t size, skinrt_host_rig** out);
/* the model's surface list (3 vertex indices per triangle); NULL / 0 for rigs that are not PMX */
SKINRT_API const uint32_t* skinrt_host_rig_pmx_indices(const skinrt_host_rig* rig, uint32_t* count);
/* vertices, triangles, materials, bones, morphs, vertex morphs, then vertices by deform class:
 * BDEF1, BDEF2+SDEF, BDEF4+QDEF, SDEF+QDEF */
SKINRT_API void skinrt_host_rig_pmx_counts(const skinrt_host_rig* rig, uint32_t out[10]);
/* descriptor for skinrt_upload_rig / skinrt_validate_rig; valid until the rig is freed */
SKINRT_API const skinrt_rig_desc* skinrt_host_rig_desc(skinrt_host_rig* rig);
/* glTF node index of flat node n (the flattening order under test) */
SKINRT_API uint32_t skinrt_host_rig_node_gltf_index(const skinrt_host_rig* rig, uint32_t n);
/* max over channels of AnimationChannelAsset.length */
SKINRT_API float skinrt_host_rig_clip_length(const skinrt_host_rig* rig, uint32_t clip);
/* non-finite morph-target tangent deltas the flattener replaced by zero (flatten.cpp: targets without a TANGENT
 * accessor get generated tangents, NaN wherever the delta mesh does not move) */
SKINRT_API uint64_t skinrt_host_rig_sanitized_tangent_deltas(const skinrt_host_rig* rig);
SKINRT_API void skinrt_host_free_rig(skinrt_host_rig* rig);
SKINRT_API const char* skinrt_host_last_error(void);
/* calculate_tangent (renderer-asset/src/tangent.rs:15-58); mode: 0 Points, 1 LineStrip,
 * 2 LineList, 3 TriangleStrip, 4 TriangleList (PrimitiveAssetMode order); indices may be NULL.
 * out: [count][3].  Returns SKINRT_OK or SKINRT_E_INVALID. */
SKINRT_API int skinrt_host_calculate_tangent(uint32_t mode, const float* positions, uint32_t count,
                                             const uint32_t* indices, uint32_t index_count, float* out);

#ifdef __cplusplus
}
#endif
#endif
