//! Loader output -> flat rig.  The same walk as `renderer_amd/host/flatten.cpp` (compiled and tested in this
//! repository); this is where the path's node order is fixed: DFS pre-order over `SceneAsset.nodes`, then
//! children, in file order (the trees come from load_node / load_scene, renderer-asset/src/loader/gltf/mod.rs:1168-1217).
//!
//! Every node with a mesh becomes a SECTION of the rig's vertex array (its primitives concatenated); a section names
//! the node's skin (distinct `SkinAsset`s by id, in the order they are met) or none — an un-skinned mesh node is drawn
//! with its world matrix (`InstanceUniform`, shader.wgsl:143-152), which the library evaluates through a pseudo-joint.
//! Morph targets, default weights and `weights` channels belong to a section (include/skinrt.h, `skinrt_section_desc`).
//!
//! Types consumed (fifth-light/renderer, crates/renderer-asset/src): node.rs:89-99 (`NodeAsset`), :9-33
//! (`NodeTransform`), skin.rs:5-11 (`SkinAsset`), animation.rs:8-40, primitive.rs:22-38, mesh.rs:3-8.
use std::collections::BTreeMap;
use std::sync::Arc;

use renderer_asset::animation::{AnimationAsset, AnimationKeyFrames, AnimationSampler};
use renderer_asset::index::AssetIndex;
use renderer_asset::node::{NodeAsset, NodeTransform};
use renderer_asset::primitive::PrimitiveAssetMode;
use renderer_asset::scene::SceneAsset;
use renderer_asset::skin::SkinAsset;
use skinrt_sys as sys;

use crate::Error;

#[derive(Debug, Clone, Default)]
pub struct FlatChannel {
    pub node: u32,
    pub path: u8,
    pub interpolation: u8,
    pub times: Vec<f32>,
    /// `[key][w]`, or `[key][in, value, out][w]` for CubicSpline (the order of AnimationKeyFrames::CubicSpline, animation.rs:18-19)
    pub values: Vec<f32>,
}

#[derive(Debug, Clone, Default)]
pub struct FlatSkin {
    pub id: Option<AssetIndex>,
    pub joint_node: Vec<u32>,
    /// empty = identity (loader/gltf/mod.rs:1139-1145)
    pub inverse_bind: Vec<f32>,
}

#[derive(Debug, Clone, Default)]
pub struct FlatTarget {
    /// not dense: rows of the SECTION (counted from its first vertex) the deltas belong to
    pub indices: Vec<u32>,
    pub dense: bool,
    pub position: Vec<f32>,
    pub normal: Vec<f32>,
    pub tangent: Vec<f32>,
}

/// Everything `skinrt_rig_desc` points at, owned.
#[derive(Debug, Clone, Default)]
pub struct FlatRig {
    pub node_id: Vec<AssetIndex>,
    pub parent: Vec<i32>,
    pub translation: Vec<f32>,
    pub rotation: Vec<f32>,
    pub scale: Vec<f32>,
    pub matrix: Vec<f32>,
    pub has_matrix: Vec<u8>,
    /// distinct skins in the order the walk meets them (a8)
    pub skins: Vec<FlatSkin>,
    /// one per mesh node, DFS order; they tile the vertex array and the target list
    pub sections: Vec<sys::skinrt_section_desc>,
    pub clips: Vec<Vec<FlatChannel>>,
    /// max over channels of `AnimationChannelAsset::length`
    pub clip_length: Vec<f32>,
    pub position: Vec<f32>,
    pub normal: Vec<f32>,
    /// stride 4 (`Tangent = Vec<[f32; 4]>`, primitive.rs:16)
    pub tangent: Vec<f32>,
    pub joints: Vec<Vec<u16>>,
    pub weights: Vec<Vec<f32>>,
    pub targets: Vec<FlatTarget>,
    pub default_weights: Vec<f32>,
    /// non-finite generated morph tangent deltas replaced by zero (see flatten.cpp)
    pub sanitized_tangent_deltas: u64,
    /// triangle corners of every `TriangleList` primitive as indices into the RIG's vertex array (`PrimitiveAsset.indices`,
    /// primitive.rs:43, offset by the primitive's first vertex; without indices: 3f, 3f+1, 3f+2 as tangent.rs:42-51 reads
    /// it): what `Topology::new` takes for per-frame normals.  Whole triangles only (`len() / 3`, tangent.rs:29).
    pub triangles: Vec<u32>,
}

/// What to take from the scenes (the defaults: everything).
#[derive(Debug, Clone)]
pub struct FlattenOptions {
    /// `None` = every scene in order
    pub scene: Option<usize>,
    /// only the mesh nodes bound to this skin (one-skin rigs, as before ABI 6)
    pub skin: Option<AssetIndex>,
    /// mesh nodes without a skin become un-skinned sections
    pub include_unskinned: bool,
}

impl Default for FlattenOptions {
    fn default() -> Self {
        Self { scene: None, skin: None, include_unskinned: true }
    }
}

struct Walk<'r, 'a> {
    rig: &'r mut FlatRig,
    index_of: BTreeMap<AssetIndex, u32>,
    /// (node index, node, skin index or -1)
    meshes: Vec<(u32, &'a NodeAsset, i32)>,
    skins: Vec<Arc<SkinAsset>>,
    opt: &'a FlattenOptions,
}

impl<'r, 'a> Walk<'r, 'a> {
    fn node(&mut self, n: &'a NodeAsset, parent: i32) {
        let me = self.rig.parent.len() as u32;
        self.index_of.entry(n.id.clone()).or_insert(me); // first occurrence wins
        self.rig.node_id.push(n.id.clone());
        self.rig.parent.push(parent);
        let mut t = [0.0f32; 3];
        let mut q = [0.0f32, 0.0, 0.0, 1.0];
        let mut s = [1.0f32; 3];
        let mut m = [0.0f32; 16];
        let mut is_matrix = 0u8;
        match n.transform.clone().unwrap_or_default() {
            // None = identity (loader/pmx/mod.rs:253)
            NodeTransform::Decomposed(d) => {
                t = d.translation.to_array();
                q = d.rotation.to_array(); // x, y, z, w — used un-normalised, as glam's from_quat does
                s = d.scale.to_array();
            }
            NodeTransform::Matrix(mm) => {
                m = mm.0.to_cols_array();
                is_matrix = 1;
            }
        }
        self.rig.translation.extend_from_slice(&t);
        self.rig.rotation.extend_from_slice(&q);
        self.rig.scale.extend_from_slice(&s);
        self.rig.matrix.extend_from_slice(&m);
        self.rig.has_matrix.push(is_matrix);
        match (&n.mesh, &n.skin) {
            // node.rs:95-96: the node names its skin (shared by Arc, cached per glTF skin: loader/gltf/mod.rs:1124-1128)
            (Some(_), Some(skin)) => {
                if self.opt.skin.as_ref().map(|want| *want == skin.id).unwrap_or(true) {
                    let k = match self.skins.iter().position(|s| s.id == skin.id) {
                        Some(k) => k,
                        None => {
                            self.skins.push(skin.clone());
                            self.skins.len() - 1
                        }
                    };
                    self.meshes.push((me, n, k as i32));
                }
            }
            (Some(_), None) if self.opt.skin.is_none() && self.opt.include_unskinned => self.meshes.push((me, n, -1)),
            _ => {}
        }
        for c in &n.children {
            self.node(c, me as i32);
        }
    }
}

fn push_keys<V: AsRef<[f32]> + std::fmt::Debug + Clone>(k: &AnimationKeyFrames<V>, ch: &mut FlatChannel) {
    match k {
        AnimationKeyFrames::Linear(frames) | AnimationKeyFrames::Step(frames) => {
            ch.interpolation = if matches!(k, AnimationKeyFrames::Step(_)) { sys::SKINRT_INTERP_STEP } else { sys::SKINRT_INTERP_LINEAR };
            for f in frames {
                ch.times.push(f.time);
                ch.values.extend_from_slice(f.value.as_ref());
            }
        }
        AnimationKeyFrames::CubicSpline(frames) => {
            ch.interpolation = sys::SKINRT_INTERP_CUBICSPLINE;
            for f in frames {
                ch.times.push(f.time);
                let (a, v, b) = &f.value; // in, val, out
                ch.values.extend_from_slice(a.as_ref());
                ch.values.extend_from_slice(v.as_ref());
                ch.values.extend_from_slice(b.as_ref());
            }
        }
    }
}

/// `scene`: `None` = every scene in order; every skin and every mesh node (see [`FlattenOptions`] for less).
pub fn flatten(scenes: &[SceneAsset], animations: &[AnimationAsset], scene: Option<usize>) -> Result<FlatRig, Error> {
    flatten_with(scenes, animations, &FlattenOptions { scene, ..Default::default() })
}

pub fn flatten_with(scenes: &[SceneAsset], animations: &[AnimationAsset], opt: &FlattenOptions) -> Result<FlatRig, Error> {
    let mut rig = FlatRig::default();
    if let Some(s) = opt.scene {
        if s >= scenes.len() {
            return Err(Error::new(sys::SKINRT_E_RANGE, "scene index out of range"));
        }
    }
    let (index_of, meshes, skins) = {
        let mut w = Walk { rig: &mut rig, index_of: BTreeMap::new(), meshes: Vec::new(), skins: Vec::new(), opt };
        for (k, sc) in scenes.iter().enumerate() {
            if opt.scene.map(|s| s != k).unwrap_or(false) {
                continue;
            }
            for n in &sc.nodes {
                w.node(n, -1);
            }
        }
        (w.index_of, w.meshes, w.skins)
    };
    if rig.parent.is_empty() {
        return Err(Error::new(sys::SKINRT_E_INVALID, "no nodes"));
    }
    if rig.has_matrix.iter().all(|&b| b == 0) {
        rig.matrix.clear();
        rig.has_matrix.clear();
    }
    // skins: joint -> node, inverse bind matrices (a8)
    for skin in &skins {
        let mut fs = FlatSkin { id: Some(skin.id.clone()), ..Default::default() };
        for j in &skin.joint_ids {
            let n = index_of.get(j).ok_or_else(|| Error::new(sys::SKINRT_E_RANGE, "skin joint id is not a node of the flattened scenes"))?;
            fs.joint_node.push(*n);
        }
        if !skin.inverse_bind_matrices.is_empty() {
            if skin.inverse_bind_matrices.len() != skin.joint_ids.len() {
                return Err(Error::new(sys::SKINRT_E_INVALID, "inverse_bind_matrices length != joint_ids length"));
            }
            for m in &skin.inverse_bind_matrices {
                fs.inverse_bind.extend_from_slice(&m.to_cols_array());
            }
        }
        rig.skins.push(fs);
    }
    // clips (a1): one per AnimationAsset, channels in file order
    for a in animations {
        let mut clip = Vec::with_capacity(a.channels.len());
        let mut length = 0.0f32;
        for c in &a.channels {
            let node = *index_of.get(&c.target_id).ok_or_else(|| Error::new(sys::SKINRT_E_RANGE, "animation channel targets an unknown node"))?;
            let mut ch = FlatChannel { node, ..Default::default() };
            match &c.sampler {
                AnimationSampler::Rotation(k) => {
                    ch.path = sys::SKINRT_PATH_ROTATION;
                    push_keys(k, &mut ch);
                }
                AnimationSampler::Translation(k) => {
                    ch.path = sys::SKINRT_PATH_TRANSLATION;
                    push_keys(k, &mut ch);
                }
                AnimationSampler::Scale(k) => {
                    ch.path = sys::SKINRT_PATH_SCALE;
                    push_keys(k, &mut ch);
                }
                // A MorphTargetWeights(AnimationKeyFrames<Vec<f32>>) variant — the reference's loader stops at todo!()
                // for it (loader/gltf/mod.rs:1312) — maps to SKINRT_PATH_WEIGHTS with the same push_keys call; the
                // channel then belongs to the section on its target node and carries that section's target count
                // per key (checked at the end of flatten.cpp; the same check goes here with the variant).
            }
            length = length.max(c.length);
            clip.push(ch);
        }
        rig.clips.push(clip);
        rig.clip_length.push(length);
    }
    // mesh (a11, a13): one section per mesh node, its primitives concatenated; the morph targets of a section are its
    // MeshAsset's (every primitive of a glTF mesh carries the same number of targets, weights live on the mesh: mesh.rs:4-8)
    let mut k_sets = 0usize;
    for (_, n, skin) in &meshes {
        if *skin >= 0 {
            for p in &n.mesh.as_ref().unwrap().primitives {
                k_sets = k_sets.max(p.attributes.joints.len().min(p.attributes.weights.len()));
            }
        }
    }
    if k_sets > 4 {
        return Err(Error::new(sys::SKINRT_E_LIMIT, "more than 4 joint/weight sets"));
    }
    if k_sets == 0 && !meshes.is_empty() {
        k_sets = 1; // un-skinned meshes only: one (unread) set keeps the descriptor regular
    }
    rig.joints = vec![Vec::new(); k_sets];
    rig.weights = vec![Vec::new(); k_sets];
    for (node, n, skin) in &meshes {
        let mesh = n.mesh.as_ref().unwrap();
        let m_targets = mesh.primitives.iter().map(|p| p.targets.len()).max().unwrap_or(0);
        let all_have_all = mesh.primitives.iter().all(|p| p.targets.len() == m_targets);
        let first_vertex = (rig.position.len() / 3) as u32;
        let first_target = rig.targets.len();
        // every primitive brings its rows: the concatenation is dense
        rig.targets.extend((0..m_targets).map(|_| FlatTarget { dense: all_have_all, ..Default::default() }));
        let mut any_normal = vec![false; m_targets];
        let mut any_tangent = vec![false; m_targets];
        for p in &mesh.primitives {
            for (m, t) in p.targets.iter().enumerate() {
                any_normal[m] |= !t.normal.is_empty();
                any_tangent[m] |= !t.tangent.is_empty();
            }
        }
        for p in &mesh.primitives {
            let at = &p.attributes;
            let count = at.position.len();
            let local_first = (rig.position.len() / 3) as u32 - first_vertex; // target rows count from the section's first vertex
            if (!at.normal.is_empty() && at.normal.len() != count) || (!at.tangent.is_empty() && at.tangent.len() != count) {
                return Err(Error::new(sys::SKINRT_E_INVALID, "primitive attribute lengths differ"));
            }
            for v in 0..count {
                rig.position.extend_from_slice(&at.position[v]);
                rig.normal.extend_from_slice(&if at.normal.is_empty() { [0.0; 3] } else { at.normal[v] });
                rig.tangent.extend_from_slice(&if at.tangent.is_empty() { [0.0; 4] } else { at.tangent[v] });
            }
            for k in 0..k_sets {
                // an un-skinned section's sets are not read by the library: zeros
                let have = *skin >= 0 && k < at.joints.len() && k < at.weights.len();
                if have && (at.joints[k].len() != count || at.weights[k].len() != count) {
                    return Err(Error::new(sys::SKINRT_E_INVALID, "joints / weights length differs from position length"));
                }
                for v in 0..count {
                    rig.joints[k].extend_from_slice(&if have { at.joints[k][v] } else { [0u16; 4] });
                    rig.weights[k].extend_from_slice(&if have { at.weights[k][v] } else { [0.0f32; 4] });
                }
            }
            if matches!(p.mode, PrimitiveAssetMode::TriangleList) {
                let first = (rig.position.len() / 3 - count) as u32;
                match &p.indices {
                    Some(idx) => {
                        for &i in &idx[..idx.len() / 3 * 3] {
                            if i as usize >= count {
                                return Err(Error::new(sys::SKINRT_E_RANGE, "primitive index beyond its positions (reference: index panic, tangent.rs:35)"));
                            }
                            rig.triangles.push(first + i);
                        }
                    }
                    None => rig.triangles.extend((0..(count / 3 * 3) as u32).map(|c| first + c)),
                }
            }
            for (m, src) in p.targets.iter().enumerate() {
                if src.position.len() != count
                    || (!src.normal.is_empty() && src.normal.len() != count)
                    || (!src.tangent.is_empty() && src.tangent.len() != count)
                {
                    return Err(Error::new(sys::SKINRT_E_INVALID, "morph target length differs from position length"));
                }
                let dst = &mut rig.targets[first_target + m];
                for v in 0..count {
                    if !dst.dense {
                        dst.indices.push(local_first + v as u32);
                    }
                    dst.position.extend_from_slice(&src.position[v]);
                    if any_normal[m] {
                        dst.normal.extend_from_slice(&if src.normal.is_empty() { [0.0; 3] } else { src.normal[v] });
                    }
                    if any_tangent[m] {
                        // generated tangents of unmoved vertices are NaN (loader/gltf/mod.rs:911-916 on zero deltas):
                        // a NaN delta poisons the vertex at any weight, so it becomes zero (as flatten.cpp)
                        let mut x = if src.tangent.is_empty() { [0.0; 4] } else { src.tangent[v] };
                        for e in x.iter_mut() {
                            if !e.is_finite() {
                                *e = 0.0;
                                rig.sanitized_tangent_deltas += 1;
                            }
                        }
                        dst.tangent.extend_from_slice(&x);
                    }
                }
            }
        }
        // default weights (a13): node.weights if non-empty, else mesh.weights, else zeros
        let src = if !n.weights.is_empty() { &n.weights } else { &mesh.weights };
        rig.default_weights.resize(rig.targets.len(), 0.0);
        for (d, s) in rig.default_weights[first_target..].iter_mut().zip(src.iter()) {
            *d = *s;
        }
        rig.sections.push(sys::skinrt_section_desc {
            node: *node,
            skin: *skin,
            first_vertex,
            vertex_count: (rig.position.len() / 3) as u32 - first_vertex,
            first_target: first_target as u32,
            target_count: m_targets as u32,
        });
    }
    Ok(rig)
}

/// `skinrt_rig_desc` over a `FlatRig`: the descriptor's pointers borrow from `rig` and from the tables held here.
pub struct RigDesc<'a> {
    pub desc: sys::skinrt_rig_desc,
    _rig: &'a FlatRig,
    _clips: Vec<sys::skinrt_clip_desc>,
    _channels: Vec<Vec<sys::skinrt_channel_desc>>,
    _targets: Vec<sys::skinrt_morph_target_desc>,
    _skins: Vec<sys::skinrt_skin_desc>,
    _joint_ptrs: Vec<*const u16>,
    _weight_ptrs: Vec<*const f32>,
}

fn ptr<T>(v: &[T]) -> *const T {
    if v.is_empty() { std::ptr::null() } else { v.as_ptr() }
}

impl<'a> RigDesc<'a> {
    pub fn new(rig: &'a FlatRig) -> Self {
        let channels: Vec<Vec<sys::skinrt_channel_desc>> = rig
            .clips
            .iter()
            .map(|clip| {
                clip.iter()
                    .map(|ch| sys::skinrt_channel_desc {
                        target_node: ch.node,
                        path: ch.path,
                        interpolation: ch.interpolation,
                        reserved: 0,
                        key_count: ch.times.len() as u32,
                        times: ptr(&ch.times),
                        values: ptr(&ch.values),
                    })
                    .collect()
            })
            .collect();
        let clips: Vec<sys::skinrt_clip_desc> =
            channels.iter().map(|c| sys::skinrt_clip_desc { channel_count: c.len() as u32, channels: ptr(c) }).collect();
        let vertex_count = (rig.position.len() / 3) as u32;
        let targets: Vec<sys::skinrt_morph_target_desc> = rig
            .targets
            .iter()
            .map(|t| sys::skinrt_morph_target_desc {
                count: (t.position.len() / 3) as u32, // dense: the section's vertex count
                indices: if t.dense { std::ptr::null() } else { ptr(&t.indices) },
                position: ptr(&t.position),
                normal: ptr(&t.normal),
                tangent: ptr(&t.tangent),
            })
            .collect();
        let joint_ptrs: Vec<*const u16> = rig.joints.iter().map(|j| ptr(j)).collect();
        let weight_ptrs: Vec<*const f32> = rig.weights.iter().map(|w| ptr(w)).collect();
        let skins: Vec<sys::skinrt_skin_desc> = rig
            .skins
            .iter()
            .map(|k| sys::skinrt_skin_desc { joint_count: k.joint_node.len() as u32, joint_node: ptr(&k.joint_node), inverse_bind: ptr(&k.inverse_bind) })
            .collect();
        let desc = sys::skinrt_rig_desc {
            struct_size: std::mem::size_of::<sys::skinrt_rig_desc>() as u32,
            node_count: rig.parent.len() as u32,
            parent: ptr(&rig.parent),
            translation: ptr(&rig.translation),
            rotation: ptr(&rig.rotation),
            scale: ptr(&rig.scale),
            matrix: ptr(&rig.matrix),
            has_matrix: ptr(&rig.has_matrix),
            // the long form: skins[] and sections[] (joint_count / joint_node / inverse_bind are the one-skin short form)
            joint_count: 0,
            joint_node: std::ptr::null(),
            inverse_bind: std::ptr::null(),
            clip_count: clips.len() as u32,
            clips: ptr(&clips),
            vertex_count,
            position: ptr(&rig.position),
            normal: ptr(&rig.normal),
            tangent: ptr(&rig.tangent),
            tangent_stride: 4,
            set_count: rig.joints.len() as u32,
            joints: ptr(&joint_ptrs),
            weights: ptr(&weight_ptrs),
            target_count: targets.len() as u32,
            targets: ptr(&targets),
            default_weights: ptr(&rig.default_weights),
            skin_count: skins.len() as u32,
            skins: ptr(&skins),
            section_count: rig.sections.len() as u32,
            sections: ptr(&rig.sections),
        };
        // moving the Vecs into the struct does not move their heap buffers: the pointers above stay valid
        RigDesc { desc, _rig: rig, _clips: clips, _channels: channels, _targets: targets, _skins: skins, _joint_ptrs: joint_ptrs, _weight_ptrs: weight_ptrs }
    }
}
