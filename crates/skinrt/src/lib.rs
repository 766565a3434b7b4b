//! Safe host side of libskinrt for the reference's Rust code: what `Entity::render`
//! (crates/renderer/src/client/entity/mod.rs:34) calls per frame, and the one-time upload of the loader
//! output.  Mirrors `renderer_amd/host/skinned_scene.hpp` (C++, compiled and tested in this repository) and
//! `renderer_amd/api.py` (Python, used by the tests and the bench).  Source only here: no cargo in the build image.
//!
//! There is no CPU fallback: `Context::new` fails with `SKINRT_E_DEVICE` without an MI355X.
pub mod flatten;

use std::ffi::CStr;
use std::marker::PhantomData;
use std::ptr;

use renderer_asset::animation::AnimationAsset;
use renderer_asset::scene::SceneAsset;
pub use skinrt_sys as sys;

#[derive(Debug, Clone)]
pub struct Error {
    pub code: i32,
    pub message: String,
}

impl Error {
    pub fn new(code: i32, message: &str) -> Self {
        Self { code, message: message.to_string() }
    }
    fn last(code: i32) -> Self {
        let message = unsafe { CStr::from_ptr(sys::skinrt_last_error()) }.to_string_lossy().into_owned();
        Self { code, message }
    }
}

impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "skinrt error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}

fn check(rc: i32) -> Result<(), Error> {
    if rc == sys::SKINRT_OK { Ok(()) } else { Err(Error::last(rc)) }
}

/// `SkinMatrix` exactly as the reference uploads it (crates/renderer/src/renderer/uniform/skin.rs:12-17): 112 bytes.
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct SkinMatrix {
    pub transform: [[f32; 4]; 4],
    pub normal: [[f32; 4]; 3],
}

/// `InstanceUniform` exactly as the reference uploads it for a mesh node without a skin
/// (crates/renderer/src/renderer/uniform/transform.rs:8-59; shader.wgsl:143-152): 124 bytes, align 4.
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct InstanceUniform {
    pub transform: [[f32; 4]; 4],
    /// Mat3(inverse(transform)^T), each column padded to 16 bytes
    pub normal: [[f32; 4]; 3],
    pub _pad: [f32; 3],
}
const _: () = assert!(std::mem::size_of::<InstanceUniform>() == sys::SKINRT_INSTANCE_UNIFORM_BYTES);

/// One HIP device + stream.  Drive a context and its crowds from one thread (the reference renders from one thread,
/// crates/renderer/src/state.rs:236).
pub struct Context {
    raw: *mut sys::skinrt_ctx,
}

impl Context {
    pub fn new(device: i32) -> Result<Self, Error> {
        Self::with_config(sys::skinrt_config { device, cubic_quirk: 1, ..Default::default() })
    }
    pub fn with_config(mut config: sys::skinrt_config) -> Result<Self, Error> {
        config.struct_size = std::mem::size_of::<sys::skinrt_config>() as u32;
        let mut raw = ptr::null_mut();
        check(unsafe { sys::skinrt_init(&config, &mut raw) })?;
        Ok(Self { raw })
    }
    pub fn synchronize(&self) -> Result<(), Error> {
        check(unsafe { sys::skinrt_synchronize(self.raw) })
    }
    /// Validate, compile and upload the loader output of one model (renderer-asset/src/loader/gltf/mod.rs:199).
    pub fn upload(&self, scenes: &[SceneAsset], animations: &[AnimationAsset]) -> Result<Rig<'_>, Error> {
        let flat = flatten::flatten(scenes, animations, None)?;
        self.upload_flat(&flat)
    }
    pub fn upload_flat(&self, flat: &flatten::FlatRig) -> Result<Rig<'_>, Error> {
        let d = flatten::RigDesc::new(flat);
        let mut raw = ptr::null_mut();
        check(unsafe { sys::skinrt_upload_rig(self.raw, &d.desc, &mut raw) })?;
        let mut info = sys::skinrt_rig_info::default();
        check(unsafe { sys::skinrt_get_rig_info(raw, &mut info) })?;
        let mut skins = Vec::with_capacity(info.skin_count as usize);
        for k in 0..info.skin_count {
            let mut l = sys::skinrt_skin_layout::default();
            check(unsafe { sys::skinrt_rig_skin_layout(raw, k, &mut l) })?;
            skins.push(l);
        }
        let mut sections = Vec::with_capacity(info.section_count as usize);
        for k in 0..info.section_count {
            let mut l = sys::skinrt_section_layout::default();
            check(unsafe { sys::skinrt_rig_section_layout(raw, k, &mut l) })?;
            sections.push(l);
        }
        Ok(Rig { raw, info, clip_length: flat.clip_length.clone(), skins, sections, _ctx: PhantomData })
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { sys::skinrt_destroy_ctx(self.raw) };
    }
}

pub struct Rig<'c> {
    raw: *mut sys::skinrt_rig,
    pub info: sys::skinrt_rig_info,
    pub clip_length: Vec<f32>,
    /// where each skin's joints lie in the palette (`FlatRig::skins` order)
    pub skins: Vec<sys::skinrt_skin_layout>,
    /// each mesh node's vertex range, target range, skin (or -1) and — un-skinned — its pseudo-joint (`FlatRig::sections` order)
    pub sections: Vec<sys::skinrt_section_layout>,
    _ctx: PhantomData<&'c Context>,
}

impl<'c> Rig<'c> {
    /// `instances` copies of this rig (one `SkinUniformBuffer` per entity in the reference, uniform/skin.rs:39-57).
    pub fn crowd(&self, ctx: &'c Context, instances: u32) -> Result<Crowd<'_>, Error> {
        let mut raw = ptr::null_mut();
        check(unsafe { sys::skinrt_create_crowd(ctx.raw, self.raw, instances, &mut raw) })?;
        Ok(Crowd { raw, ctx: ctx.raw, instances, joints: self.info.joint_count, vertices: self.info.vertex_count,
                   targets: self.info.target_count, _rig: PhantomData })
    }
}

impl Drop for Rig<'_> {
    fn drop(&mut self) {
        unsafe { sys::skinrt_destroy_rig(self.raw) };
    }
}

pub struct Crowd<'r> {
    raw: *mut sys::skinrt_crowd,
    ctx: *mut sys::skinrt_ctx,
    instances: u32,
    joints: u32,
    vertices: u32,
    targets: u32,
    _rig: PhantomData<&'r ()>,
}

impl Crowd<'_> {
    /// Per-instance clock (seconds, wrapped by the caller) and clip; `clip_id = None`: clip 0 for everyone.
    pub fn set_clocks(&mut self, tau: &[f32], clip_id: Option<&[u32]>) -> Result<(), Error> {
        if tau.len() != self.instances as usize || clip_id.map(|c| c.len() != tau.len()).unwrap_or(false) {
            return Err(Error::new(sys::SKINRT_E_INVALID, "one clock (and clip) per instance"));
        }
        check(unsafe { sys::skinrt_set_clocks(self.raw, tau.as_ptr(), clip_id.map(|c| c.as_ptr()).unwrap_or(ptr::null())) })
    }
    /// The same from DEVICE memory, stream-ordered: nothing is validated on the host and nothing waits (a clip id that is
    /// out of range means "at rest").  For frame loops that keep their clocks on the GPU.
    ///
    /// # Safety
    /// `dev_tau` (`instances` f32) and `dev_clip_id` (`instances` u32, or null = unchanged) must be valid device pointers
    /// that stay untouched until the stream has passed the copy.
    pub unsafe fn set_clocks_async(&mut self, dev_tau: *const f32, dev_clip_id: *const u32) -> Result<(), Error> {
        check(sys::skinrt_set_clocks_async(self.raw, dev_tau, dev_clip_id))
    }
    /// `tau = fmodf(tau + dt * rate, clip_length)` on the device, per instance (`dev_rate` null = 1); instances at rest
    /// keep their clock.  Enqueued like `evaluate`: a frame is `advance_clocks(dt)` + `evaluate(STAGE_ALL)`, no host wait.
    ///
    /// # Safety
    /// `dev_rate` is null or a device pointer to `instances` f32.
    pub unsafe fn advance_clocks(&mut self, dt: f32, dev_rate: *const f32) -> Result<(), Error> {
        check(sys::skinrt_advance_clocks(self.raw, dt, dev_rate))
    }
    /// `weights[instance][target]`; `None` restores the rig's defaults (NodeAsset.weights / MeshAsset.weights).
    pub fn set_morph_weights(&mut self, weights: Option<&[f32]>) -> Result<(), Error> {
        if let Some(w) = weights {
            if w.len() != self.instances as usize * self.targets as usize {
                return Err(Error::new(sys::SKINRT_E_INVALID, "instances x targets weights expected"));
            }
        }
        check(unsafe { sys::skinrt_set_morph_weights(self.raw, weights.map(|w| w.as_ptr()).unwrap_or(ptr::null())) })
    }
    /// Column-major `Mat4::to_cols_array` per instance (InstanceUniformBuffer.transform, uniform/transform.rs:17-20).
    pub fn set_root_transforms(&mut self, matrices: Option<&[[f32; 16]]>) -> Result<(), Error> {
        if let Some(m) = matrices {
            if m.len() != self.instances as usize {
                return Err(Error::new(sys::SKINRT_E_INVALID, "one root matrix per instance"));
            }
        }
        check(unsafe { sys::skinrt_set_root_transforms(self.raw, matrices.map(|m| m.as_ptr().cast()).unwrap_or(ptr::null())) })
    }
    /// Enqueue the stages (asynchronous): the body `Entity::render` leaves empty.
    pub fn evaluate(&mut self, stages: u32) -> Result<(), Error> {
        check(unsafe { sys::skinrt_evaluate(self.raw, stages) })
    }
    pub fn synchronize(&self) -> Result<(), Error> {
        check(unsafe { sys::skinrt_synchronize(self.ctx) })
    }
    /// `[SkinMatrix; 128]` of one instance, as `SkinUniformBuffer::update` builds it (uniform/skin.rs:41-50, :59-71).
    pub fn skin_uniform(&self, instance: u32) -> Result<Box<[SkinMatrix; 128]>, Error> {
        let mut out = Box::new([SkinMatrix { transform: [[0.0; 4]; 4], normal: [[0.0; 4]; 3] }; 128]);
        check(unsafe { sys::skinrt_export_skin_uniform(self.raw, instance, out.as_mut_ptr().cast()) })?;
        Ok(out)
    }
    /// The same for skin `skin` of a rig with several (`Rig::skins`): that skin's joints from slot 0, zeros behind them.
    pub fn skin_uniform_of(&self, skin: u32, instance: u32) -> Result<Box<[SkinMatrix; 128]>, Error> {
        let mut out = Box::new([SkinMatrix { transform: [[0.0; 4]; 4], normal: [[0.0; 4]; 3] }; 128]);
        check(unsafe { sys::skinrt_export_skin_uniform_of(self.raw, skin, instance, out.as_mut_ptr().cast()) })?;
        Ok(out)
    }
    /// `InstanceUniform` of node `node` (an index into `FlatRig::node_id`) for one instance: what
    /// `InstanceUniformBuffer::update` (uniform/transform.rs:40-59) would build from the node's world matrix.
    pub fn instance_uniform(&self, instance: u32, node: u32) -> Result<InstanceUniform, Error> {
        let mut out = InstanceUniform { transform: [[0.0; 4]; 4], normal: [[0.0; 4]; 3], _pad: [0.0; 3] };
        check(unsafe { sys::skinrt_export_instance_uniform(self.raw, instance, node, (&mut out as *mut InstanceUniform).cast()) })?;
        Ok(out)
    }
    /// Device pointers to the skinned positions / normals / tangents (`[instance][vertex][3]` f32 each), for a
    /// storage-buffer vertex pull or an interop import; valid until the crowd is dropped.
    pub fn device_outputs(&self) -> Result<[*mut f32; 3], Error> {
        let mut p = [ptr::null_mut::<f32>(); 3];
        let mut n = 0u64;
        check(unsafe { sys::skinrt_get_positions(self.raw, &mut p[0], &mut n) })?;
        check(unsafe { sys::skinrt_get_normals(self.raw, &mut p[1], &mut n) })?;
        check(unsafe { sys::skinrt_get_tangents(self.raw, &mut p[2], &mut n) })?;
        Ok(p)
    }
    /// Per-instance bounds `[min xyz, max xyz]`, downloaded.
    pub fn aabbs(&self) -> Result<Vec<[f32; 6]>, Error> {
        let mut dev = ptr::null_mut::<f32>();
        let mut n = 0u64;
        check(unsafe { sys::skinrt_get_aabbs(self.raw, &mut dev, &mut n) })?;
        let mut out = vec![[0.0f32; 6]; self.instances as usize];
        check(unsafe { sys::skinrt_download(self.ctx, dev.cast(), out.as_mut_ptr().cast(), n * 4) })?;
        Ok(out)
    }
    /// `TextureSkinVertex` records (68 bytes each, buffer/vertex.rs:82-91) of instances `first .. first + count`, written
    /// by one kernel into device memory the caller owns (16-byte aligned, `count * vertices * 68` bytes).
    ///
    /// # Safety
    /// `dev_dst` and `dev_tex_coords` (`[vertex][2]` f32, or null = zeros) must be valid device pointers of those sizes.
    pub unsafe fn texture_skin_vertices_device(&self, first: u32, count: u32, dev_tex_coords: *const f32, dev_dst: *mut u8) -> Result<(), Error> {
        check(sys::skinrt_export_interleaved_device(self.raw, first, count, sys::SKINRT_LAYOUT_TEXTURE_SKIN, dev_tex_coords, dev_dst.cast()))
    }
    /// The same for one mesh node (`Rig::sections[section]`): `count * section.vertex_count * 68` bytes; joint indices are
    /// local to the section's skin, as the loader gave them.
    ///
    /// # Safety
    /// As `texture_skin_vertices_device`; `dev_tex_coords` holds the SECTION's vertices.
    pub unsafe fn texture_skin_vertices_section_device(&self, section: u32, first: u32, count: u32, dev_tex_coords: *const f32,
                                                       dev_dst: *mut u8) -> Result<(), Error> {
        check(sys::skinrt_export_interleaved_section_device(self.raw, section, first, count, sys::SKINRT_LAYOUT_TEXTURE_SKIN, dev_tex_coords, dev_dst.cast()))
    }
    /// `calculate_tangent(TriangleList, positions[i], indices)` (renderer-asset/src/tangent.rs:15-58) of instances
    /// `first .. first + count` from the positions the last `evaluate` produced, written over their normals
    /// (`SKINRT_OUTPUT_NORMALS`) or tangents (`SKINRT_OUTPUT_TANGENTS`); enqueued behind the evaluate, no host wait.
    pub fn recompute_normals(&mut self, topology: &Topology<'_>, first: u32, count: u32, output: u32) -> Result<(), Error> {
        check(unsafe { sys::skinrt_recompute_normals(self.raw, topology.raw, first, count, output) })
    }
    pub fn instances(&self) -> u32 { self.instances }
    pub fn joints(&self) -> u32 { self.joints }
    pub fn vertices(&self) -> u32 { self.vertices }
}

impl Drop for Crowd<'_> {
    fn drop(&mut self) {
        unsafe { sys::skinrt_destroy_crowd(self.raw) };
    }
}

/// A mesh's triangle list prepared once for per-frame normals (`skinrt_create_topology`): `indices` as the loader
/// holds them (`PrimitiveAsset.indices`, primitive.rs:47), offset by the section's first vertex when the rig has several
/// mesh nodes; `None` = every three vertices form a triangle, as `calculate_tangent` treats a primitive without indices.
pub struct Topology<'c> {
    raw: *mut sys::skinrt_topology,
    _ctx: PhantomData<&'c ()>,
}

impl<'c> Topology<'c> {
    pub fn new(ctx: &'c Context, vertex_count: u32, indices: Option<&[u32]>) -> Result<Self, Error> {
        let mut raw = ptr::null_mut();
        check(unsafe {
            sys::skinrt_create_topology(ctx.raw, vertex_count, indices.map(|i| i.as_ptr()).unwrap_or(ptr::null()),
                                        indices.map(|i| i.len() as u32).unwrap_or(0), &mut raw)
        })?;
        Ok(Topology { raw, _ctx: PhantomData })
    }
}

impl Drop for Topology<'_> {
    fn drop(&mut self) {
        unsafe { sys::skinrt_destroy_topology(self.raw) };
    }
}

/// One crowd sharded over several devices of this process (`skinrt_node_*`): instance `i` lives on the shard
/// `skinrt_shard_range` assigns it to; there is no exchange step on the data path, `evaluate` enqueues on every device
/// and returns, `synchronize` waits for all of them.  Inputs are whole-crowd slices; outputs are reached per shard.
pub struct Node {
    raw: *mut sys::skinrt_node,
    instances: u32,
    targets: u32,
}

/// A shard's handles, borrowed from its [`Node`]: pass them to the per-device entry points of `skinrt-sys`.
pub struct Shard {
    pub ctx: *mut sys::skinrt_ctx,
    /// null for an empty shard (more devices than instances)
    pub crowd: *mut sys::skinrt_crowd,
    pub first: u32,
    pub count: u32,
}

impl Node {
    pub fn new(mut config: sys::skinrt_config, devices: &[i32], flat: &flatten::FlatRig, instances: u32) -> Result<Self, Error> {
        config.struct_size = std::mem::size_of::<sys::skinrt_config>() as u32;
        let d = flatten::RigDesc::new(flat);
        let mut raw = ptr::null_mut();
        check(unsafe { sys::skinrt_node_create(&config, devices.as_ptr(), devices.len() as u32, &d.desc, instances, &mut raw) })?;
        Ok(Self { raw, instances, targets: d.desc.target_count })
    }
    pub fn shards(&self) -> Result<Vec<Shard>, Error> {
        let mut n = 0u32;
        check(unsafe { sys::skinrt_node_shard_count(self.raw, &mut n) })?;
        (0..n).map(|g| {
            let mut s = Shard { ctx: ptr::null_mut(), crowd: ptr::null_mut(), first: 0, count: 0 };
            check(unsafe { sys::skinrt_node_get_shard(self.raw, g, &mut s.ctx, &mut s.crowd, &mut s.first, &mut s.count) })?;
            Ok(s)
        }).collect()
    }
    pub fn set_clocks(&mut self, tau: &[f32], clip_id: Option<&[u32]>) -> Result<(), Error> {
        if tau.len() != self.instances as usize || clip_id.map(|c| c.len() != tau.len()).unwrap_or(false) {
            return Err(Error::new(sys::SKINRT_E_INVALID, "one clock (and clip) per instance"));
        }
        check(unsafe { sys::skinrt_node_set_clocks(self.raw, tau.as_ptr(), clip_id.map_or(ptr::null(), |c| c.as_ptr())) })
    }
    pub fn set_morph_weights(&mut self, weights: Option<&[f32]>) -> Result<(), Error> {
        if weights.map(|w| w.len() != self.instances as usize * self.targets as usize).unwrap_or(false) {
            return Err(Error::new(sys::SKINRT_E_INVALID, "target_count weights per instance"));
        }
        check(unsafe { sys::skinrt_node_set_morph_weights(self.raw, weights.map_or(ptr::null(), |w| w.as_ptr())) })
    }
    pub fn set_root_transforms(&mut self, matrices: Option<&[[f32; 16]]>) -> Result<(), Error> {
        if matrices.map(|m| m.len() != self.instances as usize).unwrap_or(false) {
            return Err(Error::new(sys::SKINRT_E_INVALID, "one root matrix per instance"));
        }
        check(unsafe { sys::skinrt_node_set_root_transforms(self.raw, matrices.map_or(ptr::null(), |m| m.as_ptr().cast())) })
    }
    pub fn evaluate(&mut self, stages: u32) -> Result<(), Error> {
        check(unsafe { sys::skinrt_node_evaluate(self.raw, stages) })
    }
    pub fn synchronize(&self) -> Result<(), Error> {
        check(unsafe { sys::skinrt_node_synchronize(self.raw) })
    }
    /// Every instance's bounds `[min xyz, max xyz]`, gathered from all shards in instance order.
    pub fn aabbs(&self) -> Result<Vec<[f32; 6]>, Error> {
        let mut out = vec![[0.0f32; 6]; self.instances as usize];
        check(unsafe { sys::skinrt_node_download_aabbs(self.raw, out.as_mut_ptr().cast()) })?;
        Ok(out)
    }
}

impl Drop for Node {
    fn drop(&mut self) {
        unsafe { sys::skinrt_node_destroy(self.raw) };
    }
}
