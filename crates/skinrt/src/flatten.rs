//! Loader output -> flat rig.  The same walk as `renderer_amd/host/flatten.cpp` (compiled and tested in this
//! repository); this is where the path's node order is fixed: DFS pre-order over `SceneAsset.nodes`, then
//! children, in file order (the trees come from load_node / load_scene, renderer-asset/src/loader/gltf/mod.rs:1168-1217).
//!
//! Types consumed (fifth-light/renderer, crates/renderer-asset/src): node.rs:89-99 (`NodeAsset`), :9-33
//! (`NodeTransform`), skin.rs:5-11 (`SkinAsset`), animation.rs:8-40, primitive.rs:22-38, mesh.rs:3-8.
use std::collections::BTreeMap;
use std::sync::Arc;

use renderer_asset::animation::{AnimationAsset, AnimationKeyFrames, AnimationSampler};
use renderer_asset::index::AssetIndex;
use renderer_asset::node::{NodeAsset, NodeTransform};
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
pub struct FlatTarget {
    /// empty: dense (count == vertex_count)
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
    pub joint_node: Vec<u32>,
    /// empty = identity (loader/gltf/mod.rs:1139-1145)
    pub inverse_bind: Vec<f32>,
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
}

struct Walk<'r, 'a> {
    rig: &'r mut FlatRig,
    index_of: BTreeMap<AssetIndex, u32>,
    skinned: Vec<(u32, &'a NodeAsset)>,
    skin: Option<Arc<SkinAsset>>,
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
        if let (Some(skin), Some(_)) = (&n.skin, &n.mesh) {
            if self.skin.is_none() {
                self.skin = Some(skin.clone());
            }
            if self.skin.as_ref().map(|s| s.id == skin.id).unwrap_or(false) {
                self.skinned.push((me, n));
            }
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

/// `scene`: `None` = every scene in order.
pub fn flatten(scenes: &[SceneAsset], animations: &[AnimationAsset], scene: Option<usize>) -> Result<FlatRig, Error> {
    let mut rig = FlatRig::default();
    if let Some(s) = scene {
        if s >= scenes.len() {
            return Err(Error::new(sys::SKINRT_E_RANGE, "scene index out of range"));
        }
    }
    let (index_of, skinned, skin) = {
        let mut w = Walk { rig: &mut rig, index_of: BTreeMap::new(), skinned: Vec::new(), skin: None };
        for (k, sc) in scenes.iter().enumerate() {
            if scene.map(|s| s != k).unwrap_or(false) {
                continue;
            }
            for n in &sc.nodes {
                w.node(n, -1);
            }
        }
        (w.index_of, w.skinned, w.skin)
    };
    if rig.parent.is_empty() {
        return Err(Error::new(sys::SKINRT_E_INVALID, "no nodes"));
    }
    if rig.has_matrix.iter().all(|&b| b == 0) {
        rig.matrix.clear();
        rig.has_matrix.clear();
    }
    // skin: joint -> node, inverse bind matrices (a8)
    if let Some(skin) = &skin {
        for j in &skin.joint_ids {
            let n = index_of.get(j).ok_or_else(|| Error::new(sys::SKINRT_E_RANGE, "skin joint id is not a node of the flattened scenes"))?;
            rig.joint_node.push(*n);
        }
        if !skin.inverse_bind_matrices.is_empty() {
            if skin.inverse_bind_matrices.len() != skin.joint_ids.len() {
                return Err(Error::new(sys::SKINRT_E_INVALID, "inverse_bind_matrices length != joint_ids length"));
            }
            for m in &skin.inverse_bind_matrices {
                rig.inverse_bind.extend_from_slice(&m.to_cols_array());
            }
        }
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
                // for it (loader/gltf/mod.rs:1312) — maps to SKINRT_PATH_WEIGHTS with the same push_keys call.
            }
            length = length.max(c.length);
            clip.push(ch);
        }
        rig.clips.push(clip);
        rig.clip_length.push(length);
    }
    // mesh (a11, a13): primitives of the nodes that use the skin, concatenated in DFS order
    let mut k_sets = 0usize;
    let mut m_targets = 0usize;
    let mut prims = 0usize;
    for (_, n) in &skinned {
        for p in &n.mesh.as_ref().unwrap().primitives {
            k_sets = k_sets.max(p.attributes.joints.len().min(p.attributes.weights.len()));
            m_targets = m_targets.max(p.targets.len());
            prims += 1;
        }
    }
    let k_sets = k_sets.max(1);
    rig.joints = vec![Vec::new(); k_sets];
    rig.weights = vec![Vec::new(); k_sets];
    rig.targets = vec![FlatTarget { dense: prims == 1, ..Default::default() }; m_targets];
    let mut any_normal = vec![false; m_targets];
    let mut any_tangent = vec![false; m_targets];
    for (_, n) in &skinned {
        for p in &n.mesh.as_ref().unwrap().primitives {
            for (m, t) in p.targets.iter().enumerate() {
                any_normal[m] |= !t.normal.is_empty();
                any_tangent[m] |= !t.tangent.is_empty();
            }
        }
    }
    for (_, n) in &skinned {
        for p in &n.mesh.as_ref().unwrap().primitives {
            let at = &p.attributes;
            let count = at.position.len();
            let first = (rig.position.len() / 3) as u32;
            if at.normal.len() != count || at.tangent.len() != count {
                return Err(Error::new(sys::SKINRT_E_INVALID, "normal / tangent length differs from position length"));
            }
            for v in 0..count {
                rig.position.extend_from_slice(&at.position[v]);
                rig.normal.extend_from_slice(&at.normal[v]);
                rig.tangent.extend_from_slice(&at.tangent[v]);
            }
            for k in 0..k_sets {
                let have = k < at.joints.len() && k < at.weights.len();
                if have && (at.joints[k].len() != count || at.weights[k].len() != count) {
                    return Err(Error::new(sys::SKINRT_E_INVALID, "joints / weights length differs from position length"));
                }
                for v in 0..count {
                    rig.joints[k].extend_from_slice(&if have { at.joints[k][v] } else { [0u16; 4] });
                    rig.weights[k].extend_from_slice(&if have { at.weights[k][v] } else { [0.0f32; 4] });
                }
            }
            for (m, src) in p.targets.iter().enumerate() {
                if src.position.len() != count
                    || (!src.normal.is_empty() && src.normal.len() != count)
                    || (!src.tangent.is_empty() && src.tangent.len() != count)
                {
                    return Err(Error::new(sys::SKINRT_E_INVALID, "morph target length differs from position length"));
                }
                let dst = &mut rig.targets[m];
                for v in 0..count {
                    if !dst.dense {
                        dst.indices.push(first + v as u32);
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
    }
    // default weights (a13): node.weights if non-empty, else mesh.weights, else zeros
    if m_targets > 0 {
        if let Some((_, n0)) = skinned.first() {
            let src = if !n0.weights.is_empty() { &n0.weights } else { &n0.mesh.as_ref().unwrap().weights };
            rig.default_weights = vec![0.0; m_targets];
            for (d, s) in rig.default_weights.iter_mut().zip(src.iter()) {
                *d = *s;
            }
        }
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
                count: if t.dense { vertex_count } else { t.indices.len() as u32 },
                indices: if t.dense { std::ptr::null() } else { ptr(&t.indices) },
                position: ptr(&t.position),
                normal: ptr(&t.normal),
                tangent: ptr(&t.tangent),
            })
            .collect();
        let joint_ptrs: Vec<*const u16> = rig.joints.iter().map(|j| ptr(j)).collect();
        let weight_ptrs: Vec<*const f32> = rig.weights.iter().map(|w| ptr(w)).collect();
        let desc = sys::skinrt_rig_desc {
            struct_size: std::mem::size_of::<sys::skinrt_rig_desc>() as u32,
            node_count: rig.parent.len() as u32,
            parent: ptr(&rig.parent),
            translation: ptr(&rig.translation),
            rotation: ptr(&rig.rotation),
            scale: ptr(&rig.scale),
            matrix: ptr(&rig.matrix),
            has_matrix: ptr(&rig.has_matrix),
            joint_count: rig.joint_node.len() as u32,
            joint_node: ptr(&rig.joint_node),
            inverse_bind: ptr(&rig.inverse_bind),
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
        };
        // moving the Vecs into the struct does not move their heap buffers: the pointers above stay valid
        RigDesc { desc, _rig: rig, _clips: clips, _channels: channels, _targets: targets, _joint_ptrs: joint_ptrs, _weight_ptrs: weight_ptrs }
    }
}
