//! An asset-backed entity: the body `Entity::render` (client/entity/mod.rs:34) leaves empty, for models that come out of
//! `renderer_asset::loader::gltf::load_glb_from_buffer` (loader/gltf/mod.rs:1358) with skins and animations.
//!
//! One `AssetCrowd` holds every entity that shows the same model: one `skinrt::Rig` (the loader output, uploaded
//! once) and one `skinrt::Crowd` with a slot per entity.  Per frame `AssetCrowd::update` sets the clocks, runs the POSE
//! stage on the MI355X — animation sampling, node propagation, joint matrices and their inverse transposes for EVERY
//! skin of the model in one pass — and copies what the reference's shaders bind:
//!   * per (entity, skin): `[SkinMatrix; 128]`, the bytes `SkinUniformBuffer::update` (renderer/uniform/skin.rs:59-71)
//!     would compute on the host with one `Mat4::inverse().transpose()` per joint — bind group 3 (`joint_layout`,
//!     renderer/mod.rs:236-248), read by `*_skin_vs_main` (shader.wgsl:154-189);
//!   * per (entity, mesh node without a skin): the 124-byte `InstanceUniform` of the node's animated world matrix
//!     (renderer/uniform/transform.rs:8-59) — bind group 1, read by `color_vs_main` / `texture_vs_main` (shader.wgsl:143-152).
//! `AssetEntity::render` then binds them and draws the model's static vertex buffers (bind pose, `TextureSkinVertex` /
//! `TextureVertex`, built once from the same `FlatRig` sections), so the WGSL vertex stage keeps skinning.
//! (The other way to use the library — `STAGE_ALL`, world-space vertices pulled from a storage buffer, no 128-joint
//! limit — is INTEGRATION.md §3 "Full path".)
//!
//! Source only: see ../../../README.md for where this file goes and the two signatures that have to carry a lifetime.
use std::sync::Arc;

use glam::Vec3;
use renderer_asset::{animation::AnimationAsset, scene::SceneAsset};
use renderer_protocol::entity::BaseEntityData;
use uuid::Uuid;
use wgpu::{BindGroup, BindGroupDescriptor, BindGroupEntry, Buffer, BufferDescriptor, BufferUsages, Device, IndexFormat, Queue};

use skinrt::{flatten, sys, Context, Crowd, Error, Rig};

use super::{Entity, Output, State};
use crate::renderer::{
    buffer::{index::IndexBuffer, vertex::VertexBuffer},
    pipeline::RenderPipelineItem,
    OngoingRenderState, RenderBindGroups, Renderer,
};

const SKIN_UNIFORM_BYTES: u64 = 128 * 112; // [SkinMatrix; MAX_JOINTS], uniform/skin.rs:10-17
const INSTANCE_UNIFORM_BYTES: u64 = sys::SKINRT_INSTANCE_UNIFORM_BYTES as u64;

/// What one mesh node of the model is drawn with: static buffers shared by every entity.
pub struct SectionDraw {
    /// index into `Rig::sections` / `FlatRig::sections`
    pub section: u32,
    /// `*SkinVertex` records for a skinned section (joint indices local to the section's skin, as the loader gave
    /// them), `*Vertex` records for an un-skinned one — built at load from the FlatRig's bind pose
    pub vertices: VertexBuffer,
    pub indices: Option<IndexBuffer>,
    /// ShaderType::{ColorSkin, TextureSkin} for skinned sections, {Color, Texture} otherwise (pipeline.rs:197-203)
    pub pipeline: Arc<RenderPipelineItem>,
    /// `None` = `RenderBindGroups::Color`
    pub texture: Option<BindGroup>,
}

/// Per entity slot: the uniform buffers the library fills and the bind groups over them.
struct Slot {
    skin_buffers: Vec<Buffer>,        // one per skin of the rig
    skin_groups: Vec<BindGroup>,      // bind group 3
    node_buffers: Vec<Buffer>,        // one per un-skinned section
    node_groups: Vec<BindGroup>,      // bind group 1
    clip: u32,
    started_at: f32,
}

/// Every entity showing one model.  Field order = drop order: the crowd goes before its rig, the rig before the context.
pub struct AssetCrowd<'c> {
    crowd: Crowd<'c>,
    rig: Rig<'c>,
    sections: Vec<SectionDraw>,
    /// for each un-skinned section: (position in `Slot::node_*`, node index)
    rigid: Vec<(usize, u32)>,
    slots: Vec<Slot>,
    tau: Vec<f32>,
    clip: Vec<u32>,
}

impl<'c> AssetCrowd<'c> {
    /// Upload the loader output once and make room for `capacity` entities.
    pub fn new(ctx: &'c Context, device: &Device, renderer: &Renderer, scenes: &[SceneAsset], animations: &[AnimationAsset],
               sections: Vec<SectionDraw>, capacity: u32) -> Result<Self, Error> {
        let flat = flatten::flatten(scenes, animations, None)?; // every skin, every mesh node
        let rig = ctx.upload_flat(&flat)?;
        if rig.skins.iter().any(|s| s.joint_count > 128) {
            return Err(Error::new(sys::SKINRT_E_LIMIT, "a skin has more than MAX_JOINTS = 128 joints: use the full path"));
        }
        let crowd = rig.crowd(ctx, capacity)?;
        let rigid: Vec<(usize, u32)> =
            rig.sections.iter().filter(|s| s.skin < 0).enumerate().map(|(k, s)| (k, s.node)).collect();
        let layouts = renderer.bind_group_layout();
        let uniform = |size: u64, label: &str| {
            device.create_buffer(&BufferDescriptor { label: Some(label), size, usage: BufferUsages::UNIFORM | BufferUsages::COPY_DST, mapped_at_creation: false })
        };
        let slots = (0..capacity)
            .map(|_| {
                let skin_buffers: Vec<Buffer> = rig.skins.iter().map(|_| uniform(SKIN_UNIFORM_BYTES, "skinrt joints")).collect();
                let skin_groups = skin_buffers
                    .iter()
                    .map(|b| device.create_bind_group(&BindGroupDescriptor {
                        label: Some("skinrt joint bind group"),
                        layout: layouts.joint_layout(),
                        entries: &[BindGroupEntry { binding: 0, resource: b.as_entire_binding() }],
                    }))
                    .collect();
                // InstanceUniform is 124 bytes; the reference's own buffer has that size too (uniform/transform.rs:44-49)
                let node_buffers: Vec<Buffer> = rigid.iter().map(|_| uniform(INSTANCE_UNIFORM_BYTES, "skinrt node")).collect();
                let node_groups = node_buffers
                    .iter()
                    .map(|b| device.create_bind_group(&BindGroupDescriptor {
                        label: Some("skinrt node bind group"),
                        layout: layouts.instance_uniform_layout(),
                        entries: &[BindGroupEntry { binding: 0, resource: b.as_entire_binding() }],
                    }))
                    .collect();
                Slot { skin_buffers, skin_groups, node_buffers, node_groups, clip: sys::SKINRT_NO_CLIP, started_at: 0.0 }
            })
            .collect();
        Ok(Self { crowd, rig, sections, rigid, slots, tau: vec![0.0; capacity as usize], clip: vec![sys::SKINRT_NO_CLIP; capacity as usize] })
    }

    /// Start (or stop, `None`) an animation of one entity; `now` is the frame clock `update` is called with.
    pub fn play(&mut self, slot: u32, clip: Option<u32>, now: f32) {
        let s = &mut self.slots[slot as usize];
        s.clip = clip.unwrap_or(sys::SKINRT_NO_CLIP);
        s.started_at = now;
    }

    /// Once per frame, before the render pass is opened (state.rs:236): clocks -> pose stage -> uniform buffers.
    /// The clock wraps at the clip's length here, as the path's rule a4 leaves it to the caller.
    pub fn update(&mut self, queue: &Queue, now: f32) -> Result<(), Error> {
        for (i, s) in self.slots.iter().enumerate() {
            self.clip[i] = s.clip;
            let len = self.rig.clip_length.get(s.clip as usize).copied().unwrap_or(0.0);
            self.tau[i] = if len > 0.0 { (now - s.started_at) % len } else { 0.0 };
        }
        self.crowd.set_clocks(&self.tau, Some(&self.clip))?;
        self.crowd.evaluate(sys::SKINRT_STAGE_POSE)?; // palettes of every skin + node world matrices, one pass
        for (i, s) in self.slots.iter().enumerate() {
            for (k, buffer) in s.skin_buffers.iter().enumerate() {
                let palette = self.crowd.skin_uniform_of(k as u32, i as u32)?; // synchronises with the stream
                let bytes = unsafe { std::slice::from_raw_parts(palette.as_ptr().cast::<u8>(), SKIN_UNIFORM_BYTES as usize) };
                queue.write_buffer(buffer, 0, bytes);
            }
            for ((k, node), buffer) in self.rigid.iter().zip(s.node_buffers.iter()) {
                debug_assert!(*k < s.node_buffers.len());
                let u = self.crowd.instance_uniform(i as u32, *node)?;
                let bytes = unsafe { std::slice::from_raw_parts((&u as *const skinrt::InstanceUniform).cast::<u8>(), INSTANCE_UNIFORM_BYTES as usize) };
                queue.write_buffer(buffer, 0, bytes);
            }
        }
        Ok(())
    }

    /// Draw entity `slot`: every mesh node of the model with its own skin or node uniform.
    pub fn draw<'a>(&'a self, slot: u32, state: &mut OngoingRenderState<'a>) {
        let s = &self.slots[slot as usize];
        let mut rigid_at = 0usize;
        for d in &self.sections {
            let layout = &self.rig.sections[d.section as usize];
            state.render_pass.set_pipeline(d.pipeline.render_pipeline());
            let restore = if layout.skin >= 0 {
                state.set_joint(Some(&s.skin_groups[layout.skin as usize]));
                None
            } else {
                state.set_joint(None);
                let previous = state.set_instance(&s.node_groups[rigid_at]);
                rigid_at += 1;
                Some(previous)
            };
            match &d.texture {
                Some(texture) => state.bind_groups(RenderBindGroups::Texture { texture }),
                None => state.bind_groups(RenderBindGroups::Color),
            }
            match &d.indices {
                Some(indices) => {
                    state.render_pass.set_vertex_buffer(0, d.vertices.buffer().slice(..));
                    state.render_pass.set_index_buffer(indices.buffer().slice(..), IndexFormat::Uint32);
                    state.render_pass.draw_indexed(0..indices.indices() as u32, 0, 0..1);
                }
                None => d.vertices.draw(&mut state.render_pass),
            }
            if let Some(previous) = restore {
                state.set_instance(previous);
            }
        }
        state.set_joint(None);
    }
}

/// The entity the world keeps (client/world.rs:15-19): its protocol data plus a slot in its model's crowd.
pub struct AssetEntity<'c> {
    base: BaseEntityData,
    crowd: Arc<AssetCrowd<'c>>,
    slot: u32,
}

impl std::fmt::Debug for AssetEntity<'_> {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("AssetEntity").field("base", &self.base).field("slot", &self.slot).finish()
    }
}

impl<'c> AssetEntity<'c> {
    pub fn new(base: BaseEntityData, crowd: Arc<AssetCrowd<'c>>, slot: u32) -> Self {
        Self { base, crowd, slot }
    }
    pub fn id(&self) -> Uuid {
        State::id(&self.base)
    }
    pub fn position(&self) -> Vec3 {
        State::position(&self.base)
    }
    /// What `Entity::render` does once the trait carries the lifetime (README.md): `impl Entity for AssetEntity` then
    /// forwards `render` here and `base_data` / `process_output` as `ObjectEntity` does (client/entity/object.rs:36-49).
    pub fn render<'a>(&'a self, render_state: &mut OngoingRenderState<'a>) {
        self.crowd.draw(self.slot, render_state);
    }
}

// `Output` is implemented for the protocol's output types where they are defined (client/entity/object.rs:16); an asset
// entity that only moves reuses `ObjectEntityOutput`.
#[allow(unused)]
fn _assert_traits<T: Output>() {}
#[allow(unused)]
fn _entity_is_object_shaped<E: Entity>() {}
