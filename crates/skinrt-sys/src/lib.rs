//! Raw bindings of `include/skinrt.h` (ABI version 7), field for field and function for function.
//!
//! `tests/test_rust_bindings.py` (CPU) parses the `#[repr(C)]` structs and the `extern "C"` block below, lays
//! the structs out by the C rules and compares sizes, field offsets and the function list with what `gcc`
//! reports for the header — this file cannot drift from the header unnoticed even though the build image
//! has no Rust toolchain.
//!
//! Reference types each struct stands in for (fifth-light/renderer, under `crates/`):
//!   skinrt_channel_desc       AnimationChannelAsset / AnimationSampler / AnimationKeyFrames  renderer-asset/src/animation.rs:8-34
//!   skinrt_clip_desc          AnimationAsset                                                  renderer-asset/src/animation.rs:36-40
//!   skinrt_morph_target_desc  PrimitiveAssetMorphTarget                                       renderer-asset/src/primitive.rs:33-38
//!   skinrt_rig_desc           NodeAsset tree + SkinAsset + PrimitiveAssetAttributes           node.rs:89-99, skin.rs:5-11, primitive.rs:22-31
//!   skinrt_skin_desc          SkinAsset (one per glTF skin, shared by Arc)                    renderer-asset/src/skin.rs:5-11
//!   skinrt_section_desc       a NodeAsset with mesh: Some(..), its skin or none               renderer-asset/src/node.rs:89-99
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const SKINRT_ABI_VERSION: c_int = 7;

pub const SKINRT_OK: c_int = 0;
pub const SKINRT_E_INVALID: c_int = -1;
pub const SKINRT_E_DEVICE: c_int = -2;
pub const SKINRT_E_NOMEM: c_int = -3;
pub const SKINRT_E_NONAFFINE: c_int = -4;
pub const SKINRT_E_RANGE: c_int = -5;
pub const SKINRT_E_LIMIT: c_int = -6;

pub const SKINRT_PATH_TRANSLATION: u8 = 0;
pub const SKINRT_PATH_ROTATION: u8 = 1;
pub const SKINRT_PATH_SCALE: u8 = 2;
pub const SKINRT_PATH_WEIGHTS: u8 = 3;
pub const SKINRT_INTERP_LINEAR: u8 = 0;
pub const SKINRT_INTERP_STEP: u8 = 1;
pub const SKINRT_INTERP_CUBICSPLINE: u8 = 2;

pub const SKINRT_STAGE_POSE: u32 = 1;
pub const SKINRT_STAGE_SKIN: u32 = 2;
pub const SKINRT_STAGE_AABB: u32 = 4;
pub const SKINRT_STAGE_ALL: u32 = 7;

pub const SKINRT_LAYOUT_COLOR_SKIN: u32 = 0;
pub const SKINRT_LAYOUT_TEXTURE_SKIN: u32 = 1;
pub const SKINRT_MAX_JOINTS_REFERENCE: u32 = 128;
pub const SKINRT_SKIN_MATRIX_BYTES: usize = 112;
pub const SKINRT_INSTANCE_UNIFORM_BYTES: usize = 124;
pub const SKINRT_BUILD_ABLATE: u32 = 1;
pub const SKINRT_BUILD_EXP: u32 = 2;
pub const SKINRT_BUILD_STAMPS: u32 = 4;
pub const SKINRT_BUILD_TUNING: u32 = 8;
pub const SKINRT_TANGENT_NOALLOC_MAX_VERTICES: u32 = 262_144;
pub const SKINRT_TANGENT_NOALLOC_MAX_INDICES: u32 = 1_572_864;
pub const SKINRT_TANGENT_MAX_VALENCE: u32 = 49_152;
pub const SKINRT_OUTPUT_NORMALS: u32 = 0;
pub const SKINRT_OUTPUT_TANGENTS: u32 = 1;

#[repr(C)]
pub struct skinrt_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct skinrt_rig {
    _private: [u8; 0],
}
#[repr(C)]
pub struct skinrt_crowd {
    _private: [u8; 0],
}
#[repr(C)]
pub struct skinrt_node {
    _private: [u8; 0],
}
#[repr(C)]
pub struct skinrt_topology {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct skinrt_config {
    pub struct_size: u32,
    pub device: i32,
    pub cubic_quirk: u32,
    pub normalize_outputs: u32,
    pub reference_limits: u32,
    pub exact_arithmetic: u32,
    pub pose_path: u32,
    pub skin_path: u32,
    pub output_placement: u32,
    pub allow_experimental_build: u32,
    pub placement_max_filler_gib: u32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct skinrt_channel_desc {
    pub target_node: u32,
    pub path: u8,
    pub interpolation: u8,
    pub reserved: u16,
    pub key_count: u32,
    pub times: *const f32,
    pub values: *const f32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct skinrt_clip_desc {
    pub channel_count: u32,
    pub channels: *const skinrt_channel_desc,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct skinrt_morph_target_desc {
    pub count: u32,
    pub indices: *const u32,
    pub position: *const f32,
    pub normal: *const f32,
    pub tangent: *const f32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct skinrt_skin_desc {
    pub joint_count: u32,
    pub joint_node: *const u32,
    pub inverse_bind: *const f32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct skinrt_section_desc {
    pub node: u32,
    pub skin: i32,
    pub first_vertex: u32,
    pub vertex_count: u32,
    pub first_target: u32,
    pub target_count: u32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct skinrt_rig_desc {
    pub struct_size: u32,
    pub node_count: u32,
    pub parent: *const i32,
    pub translation: *const f32,
    pub rotation: *const f32,
    pub scale: *const f32,
    pub matrix: *const f32,
    pub has_matrix: *const u8,
    pub joint_count: u32,
    pub joint_node: *const u32,
    pub inverse_bind: *const f32,
    pub clip_count: u32,
    pub clips: *const skinrt_clip_desc,
    pub vertex_count: u32,
    pub position: *const f32,
    pub normal: *const f32,
    pub tangent: *const f32,
    pub tangent_stride: u32,
    pub set_count: u32,
    pub joints: *const *const u16,
    pub weights: *const *const f32,
    pub target_count: u32,
    pub targets: *const skinrt_morph_target_desc,
    pub default_weights: *const f32,
    pub skin_count: u32,
    pub skins: *const skinrt_skin_desc,
    pub section_count: u32,
    pub sections: *const skinrt_section_desc,
}

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct skinrt_rig_info {
    pub node_count: u32,
    pub joint_count: u32,
    pub vertex_count: u32,
    pub set_count: u32,
    pub target_count: u32,
    pub clip_count: u32,
    pub level_count: u32,
    pub tile_count: u32,
    pub tile_vertices: u32,
    pub max_tile_joints: u32,
    pub lds_tiles: u32,
    pub max_channels: u32,
    pub morph_entries: u64,
    pub device_bytes: u64,
    pub exact_arithmetic: u32,
    pub bind_extent: f32,
    pub weight_channel_clips: u32,
    pub skip_weight_sets: u32,
    pub skin_count: u32,
    pub section_count: u32,
    pub palette_count: u32,
    pub handed_over_tiles: u32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct skinrt_skin_layout {
    pub first_palette_entry: u32,
    pub joint_count: u32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct skinrt_section_layout {
    pub node: u32,
    pub skin: i32,
    pub first_vertex: u32,
    pub vertex_count: u32,
    pub first_target: u32,
    pub target_count: u32,
    pub first_palette_entry: u32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct skinrt_placement_info {
    pub strategy: u32,
    pub reserved: u32,
    pub filler_bytes: u64,
    pub probe_together_ms: f32,
    pub probe_alone_ms: f32,
    pub spread_gain: f32,
    pub probe_GBps: f32,
}

extern "C" {
    pub fn skinrt_abi_version() -> c_int;
    pub fn skinrt_build_flags() -> u32;
    pub fn skinrt_last_error() -> *const c_char;

    pub fn skinrt_init(config: *const skinrt_config, out_ctx: *mut *mut skinrt_ctx) -> c_int;
    pub fn skinrt_destroy_ctx(ctx: *mut skinrt_ctx) -> c_int;
    pub fn skinrt_upload_rig(ctx: *mut skinrt_ctx, desc: *const skinrt_rig_desc, out_rig: *mut *mut skinrt_rig) -> c_int;
    pub fn skinrt_validate_rig(config: *const skinrt_config, desc: *const skinrt_rig_desc, out_info: *mut skinrt_rig_info) -> c_int;
    pub fn skinrt_destroy_rig(rig: *mut skinrt_rig) -> c_int;
    pub fn skinrt_get_rig_info(rig: *mut skinrt_rig, out_info: *mut skinrt_rig_info) -> c_int;
    pub fn skinrt_rig_skin_layout(rig: *mut skinrt_rig, skin: u32, out: *mut skinrt_skin_layout) -> c_int;
    pub fn skinrt_rig_section_layout(rig: *mut skinrt_rig, section: u32, out: *mut skinrt_section_layout) -> c_int;
    pub fn skinrt_create_crowd(ctx: *mut skinrt_ctx, rig: *mut skinrt_rig, instances: u32, out_crowd: *mut *mut skinrt_crowd) -> c_int;
    pub fn skinrt_destroy_crowd(crowd: *mut skinrt_crowd) -> c_int;
    pub fn skinrt_get_output_placement(crowd: *mut skinrt_crowd, out_info: *mut skinrt_placement_info) -> c_int;

    pub fn skinrt_shard_range(instances: u32, rank: u32, world: u32, out_first: *mut u32, out_count: *mut u32) -> c_int;

    pub fn skinrt_set_clocks(crowd: *mut skinrt_crowd, tau: *const f32, clip_id: *const u32) -> c_int;
    pub fn skinrt_set_clocks_async(crowd: *mut skinrt_crowd, tau: *const f32, clip_id: *const u32) -> c_int;
    pub fn skinrt_advance_clocks(crowd: *mut skinrt_crowd, dt: f32, dev_rate: *const f32) -> c_int;
    pub fn skinrt_get_clocks(crowd: *mut skinrt_crowd, out_dev: *mut *mut f32, out_count: *mut u64) -> c_int;
    pub fn skinrt_set_morph_weights(crowd: *mut skinrt_crowd, weights: *const f32) -> c_int;
    pub fn skinrt_set_root_transforms(crowd: *mut skinrt_crowd, matrices: *const f32) -> c_int;

    pub fn skinrt_evaluate(crowd: *mut skinrt_crowd, stages_mask: u32) -> c_int;
    pub fn skinrt_synchronize(ctx: *mut skinrt_ctx) -> c_int;
    pub fn skinrt_last_stage_ms(crowd: *mut skinrt_crowd, ms: *mut f32) -> c_int;
    pub fn skinrt_get_stream(ctx: *mut skinrt_ctx, out_stream: *mut *mut c_void) -> c_int;

    pub fn skinrt_get_positions(crowd: *mut skinrt_crowd, out_dev: *mut *mut f32, out_count: *mut u64) -> c_int;
    pub fn skinrt_get_normals(crowd: *mut skinrt_crowd, out_dev: *mut *mut f32, out_count: *mut u64) -> c_int;
    pub fn skinrt_get_tangents(crowd: *mut skinrt_crowd, out_dev: *mut *mut f32, out_count: *mut u64) -> c_int;
    pub fn skinrt_get_aabbs(crowd: *mut skinrt_crowd, out_dev: *mut *mut f32, out_count: *mut u64) -> c_int;
    pub fn skinrt_get_palette(crowd: *mut skinrt_crowd, out_dev: *mut *mut f32, out_count: *mut u64) -> c_int;
    pub fn skinrt_get_world(crowd: *mut skinrt_crowd, out_dev: *mut *mut f32, out_count: *mut u64) -> c_int;
    pub fn skinrt_get_keyframe_index(crowd: *mut skinrt_crowd, out_dev: *mut *mut u32, out_count: *mut u64) -> c_int;
    pub fn skinrt_download(ctx: *mut skinrt_ctx, dev_src: *const c_void, host_dst: *mut c_void, bytes: u64) -> c_int;

    pub fn skinrt_export_skin_uniform(crowd: *mut skinrt_crowd, instance: u32, dst: *mut c_void) -> c_int;
    pub fn skinrt_export_interleaved(crowd: *mut skinrt_crowd, instance: u32, layout: u32, attr: *const f32, dst: *mut c_void) -> c_int;
    pub fn skinrt_export_skin_uniform_of(crowd: *mut skinrt_crowd, skin: u32, instance: u32, dst: *mut c_void) -> c_int;
    pub fn skinrt_export_interleaved_section(crowd: *mut skinrt_crowd, section: u32, instance: u32, layout: u32, attr: *const f32,
                                             dst: *mut c_void) -> c_int;
    pub fn skinrt_export_instance_uniform(crowd: *mut skinrt_crowd, instance: u32, node: u32, dst: *mut c_void) -> c_int;
    pub fn skinrt_export_skin_uniform_of_device(crowd: *mut skinrt_crowd, skin: u32, first: u32, count: u32, dev_dst: *mut c_void) -> c_int;
    pub fn skinrt_export_interleaved_section_device(crowd: *mut skinrt_crowd, section: u32, first: u32, count: u32, layout: u32,
                                                    dev_attr: *const f32, dev_dst: *mut c_void) -> c_int;
    pub fn skinrt_export_skin_storage_device(crowd: *mut skinrt_crowd, skin: u32, first: u32, count: u32, slots: u32, dev_dst: *mut c_void) -> c_int;
    pub fn skinrt_export_instance_uniform_device(crowd: *mut skinrt_crowd, first: u32, count: u32, dev_nodes: *const u32, node_count: u32,
                                                 stride: u32, dev_dst: *mut c_void) -> c_int;
    pub fn skinrt_export_skin_uniform_device(crowd: *mut skinrt_crowd, first: u32, count: u32, dev_dst: *mut c_void) -> c_int;
    pub fn skinrt_export_interleaved_device(crowd: *mut skinrt_crowd, first: u32, count: u32, layout: u32, dev_attr: *const f32,
                                            dev_dst: *mut c_void) -> c_int;

    pub fn skinrt_calculate_tangent(ctx: *mut skinrt_ctx, mode: u32, positions: *const f32, vertex_count: u32, indices: *const u32,
                                    index_count: u32, out: *mut f32, out_device_ms: *mut f32) -> c_int;
    pub fn skinrt_calculate_tangent_noalloc(ctx: *mut skinrt_ctx, mode: u32, dev_positions: *const f32, vertex_count: u32,
                                            dev_indices: *const u32, index_count: u32, dev_out: *mut f32) -> c_int;
    pub fn skinrt_create_topology(ctx: *mut skinrt_ctx, vertex_count: u32, indices: *const u32, index_count: u32,
                                  out_topology: *mut *mut skinrt_topology) -> c_int;
    pub fn skinrt_destroy_topology(topology: *mut skinrt_topology) -> c_int;
    pub fn skinrt_recompute_normals(crowd: *mut skinrt_crowd, topology: *mut skinrt_topology, first: u32, count: u32, output: u32) -> c_int;

    pub fn skinrt_node_create(config: *const skinrt_config, devices: *const i32, device_count: u32, desc: *const skinrt_rig_desc,
                              instances: u32, out_node: *mut *mut skinrt_node) -> c_int;
    pub fn skinrt_node_destroy(node: *mut skinrt_node) -> c_int;
    pub fn skinrt_node_shard_count(node: *mut skinrt_node, out_count: *mut u32) -> c_int;
    pub fn skinrt_node_get_shard(node: *mut skinrt_node, shard: u32, out_ctx: *mut *mut skinrt_ctx, out_crowd: *mut *mut skinrt_crowd,
                                 out_first: *mut u32, out_count: *mut u32) -> c_int;
    pub fn skinrt_node_set_clocks(node: *mut skinrt_node, tau: *const f32, clip_id: *const u32) -> c_int;
    pub fn skinrt_node_set_morph_weights(node: *mut skinrt_node, weights: *const f32) -> c_int;
    pub fn skinrt_node_set_root_transforms(node: *mut skinrt_node, matrices: *const f32) -> c_int;
    pub fn skinrt_node_evaluate(node: *mut skinrt_node, stages_mask: u32) -> c_int;
    pub fn skinrt_node_synchronize(node: *mut skinrt_node) -> c_int;
    pub fn skinrt_node_download_aabbs(node: *mut skinrt_node, host_out: *mut f32) -> c_int;
}
