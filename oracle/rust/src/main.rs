//! Cross-check of the C oracle (oracle/skin_oracle.c) against the reference's own code and glam 0.29.2.
//! TEST INFRASTRUCTURE ONLY; optional; not built in the repository's image (no cargo there).
//!
//! What comes from the reference / glam here, and is therefore what this run pins:
//!   * `renderer_asset::animation::Interpolate::{linear, cubic_spline}` (animation.rs:46-64), called on glam
//!     `Vec3` / `Quat` / `f32` values — including the `t2 = t.powi(3)` line as compiled by rustc;
//!   * `Mat4::from(NodeTransform)` (node.rs:41-87): `from_translation * from_quat * from_scale`, and
//!     `to_scale_rotation_translation` for an animated Matrix node;
//!   * glam's `Mat4 * Mat4`, `Mat4::inverse`, `transpose`, `Mat3::from_mat4` (uniform/skin.rs:30-37);
//!   * the WGSL association of shader.wgsl:127-141,159-163 written with glam `Mat4 * f32`, `Mat4 + Mat4`,
//!     `Mat4 * Vec4`, `Mat3 * Vec3` (column sums in the same order as WGSL's matrix-vector product).
//! What is NOT in the reference and stays this repository's rule: the keyframe search (a4), the DFS order of the
//! flat rig (fixed by the fixture), morph blending order (a13), bounds (a14), "weights" channels.
//!
//! Fixtures with cubic_quirk = 0 are skipped (the reference has only the quirk).
use std::{env, fs, process::exit};

use glam::{Mat3, Mat4, Quat, Vec3, Vec4};
use renderer_asset::animation::Interpolate;
use renderer_asset::node::{DecomposedTransform, MatrixNodeTransform, NodeTransform};

struct Reader {
    data: Vec<u8>,
    at: usize,
}

impl Reader {
    fn u32s(&mut self, n: usize) -> Vec<u32> {
        let s = &self.data[self.at..self.at + 4 * n];
        self.at += 4 * n;
        s.chunks_exact(4).map(|c| u32::from_le_bytes([c[0], c[1], c[2], c[3]])).collect()
    }
    fn i32s(&mut self, n: usize) -> Vec<i32> {
        self.u32s(n).into_iter().map(|x| x as i32).collect()
    }
    fn f32s(&mut self, n: usize) -> Vec<f32> {
        self.u32s(n).into_iter().map(f32::from_bits).collect()
    }
}

struct Channel {
    node: usize,
    path: u32,
    interp: u32,
    times: Vec<f32>,
    values: Vec<f32>,
}

struct Target {
    indices: Option<Vec<u32>>,
    position: Vec<f32>,
    normal: Option<Vec<f32>>,
    tangent: Option<Vec<f32>>,
}

/// a4 (this repository's rule; absent in the reference): clamp at the ends, else the largest k with times[k] <= tau
fn find_key(times: &[f32], tau: f32) -> (usize, f32, bool) {
    let n = times.len();
    if n == 1 || tau <= times[0] {
        return (0, 0.0, true);
    }
    if tau >= times[n - 1] {
        return (n - 1, 0.0, true);
    }
    let (mut lo, mut hi) = (0usize, n - 1);
    while hi - lo > 1 {
        let mid = (lo + hi) >> 1;
        if times[mid] <= tau { lo = mid } else { hi = mid }
    }
    (lo, (tau - times[lo]) / (times[lo + 1] - times[lo]), false)
}

/// Sample with the reference's trait on a glam / scalar type built by `make` from `w` floats.
fn sample<T: Interpolate + Copy>(ch: &Channel, w: usize, tau: f32, make: impl Fn(&[f32]) -> T) -> (usize, T) {
    let (k, u, clamped) = find_key(&ch.times, tau);
    let v = &ch.values;
    if ch.interp == 2 {
        let at = |key: usize, part: usize| make(&v[(key * 3 + part) * w..(key * 3 + part + 1) * w]);
        if clamped {
            return (k, at(k, 1));
        }
        let td = ch.times[k + 1] - ch.times[k];
        // cubic_spline(vk, bk, vk_1, ak_1, t, td): value k, OUT tangent k, value k+1, IN tangent k+1
        return (k, T::cubic_spline(at(k, 1), at(k, 2), at(k + 1, 1), at(k + 1, 0), u, td));
    }
    let at = |key: usize| make(&v[key * w..(key + 1) * w]);
    if clamped || ch.interp == 1 {
        return (k, at(k));
    }
    (k, T::linear(at(k), at(k + 1), u))
}

struct Diff {
    name: &'static str,
    bad: usize,
    total: usize,
    max_abs: f32,
}

fn compare(name: &'static str, got: &[f32], want: &[f32]) -> Diff {
    assert_eq!(got.len(), want.len(), "{name}: length");
    let mut d = Diff { name, bad: 0, total: got.len(), max_abs: 0.0 };
    for (g, w) in got.iter().zip(want) {
        // +0 / -0 count as equal (the comparison the repository's tests use), everything else bit for bit
        if g.to_bits() != w.to_bits() && !(*g == 0.0 && *w == 0.0) {
            d.bad += 1;
            d.max_abs = d.max_abs.max((g - w).abs());
        }
    }
    d
}

fn main() {
    let path = env::args().nth(1).unwrap_or_else(|| {
        eprintln!("usage: skin-oracle-rs <fixture.skrf>   (python oracle/rust/export_fixture.py golden.npz fixture.skrf)");
        exit(2)
    });
    let mut r = Reader { data: fs::read(&path).expect("cannot read the fixture"), at: 0 };
    assert_eq!(&r.data[0..4], b"SKRF", "not a fixture file");
    r.at = 4;
    assert_eq!(r.u32s(1)[0], 1, "fixture version");
    let h = r.u32s(16);
    let (n, j, v, k_sets, m, clips, inst, cmax, ts) =
        (h[0] as usize, h[1] as usize, h[2] as usize, h[3] as usize, h[4] as usize, h[5] as usize, h[6] as usize, h[7] as usize, h[8] as usize);
    let (has_matrix, has_ibm, has_dw, has_mw, has_root, quirk, normalize) = (h[9] != 0, h[10] != 0, h[11] != 0, h[12] != 0, h[13] != 0, h[14], h[15] != 0);
    if quirk == 0 {
        println!("SKIPPED: cubic_quirk = 0 is the glTF-spec Hermite basis, which the reference does not have");
        return;
    }
    let parent = r.i32s(n);
    let translation = r.f32s(n * 3);
    let rotation = r.f32s(n * 4);
    let scale = r.f32s(n * 3);
    let (matrix, node_is_matrix) = if has_matrix { (r.f32s(n * 16), r.u32s(n)) } else { (vec![], vec![0; n]) };
    let joint_node = r.u32s(j);
    let ibm = if has_ibm { r.f32s(j * 16) } else { vec![] };
    let mut clip_channels: Vec<Vec<Channel>> = Vec::new();
    for _ in 0..clips {
        let cc = r.u32s(1)[0] as usize;
        let mut chans = Vec::with_capacity(cc);
        for _ in 0..cc {
            let hd = r.u32s(4);
            let w = match hd[1] { 1 => 4, 3 => m, _ => 3 };
            let kc = hd[3] as usize;
            let times = r.f32s(kc);
            let values = r.f32s(kc * w * if hd[2] == 2 { 3 } else { 1 });
            chans.push(Channel { node: hd[0] as usize, path: hd[1], interp: hd[2], times, values });
        }
        clip_channels.push(chans);
    }
    let position = r.f32s(v * 3);
    let normal = r.f32s(v * 3);
    let tangent = r.f32s(v * ts);
    let mut joints = Vec::new();
    let mut weights = Vec::new();
    for _ in 0..k_sets {
        joints.push(r.u32s(v * 4));
        weights.push(r.f32s(v * 4));
    }
    let mut targets = Vec::new();
    for _ in 0..m {
        let hd = r.u32s(4);
        let cnt = hd[0] as usize;
        let indices = if hd[1] != 0 { Some(r.u32s(cnt)) } else { None };
        let p = r.f32s(cnt * 3);
        let nn = if hd[2] != 0 { Some(r.f32s(cnt * 3)) } else { None };
        let tt = if hd[3] != 0 { Some(r.f32s(cnt * ts)) } else { None };
        targets.push(Target { indices, position: p, normal: nn, tangent: tt });
    }
    let default_weights = if has_dw { r.f32s(m) } else { vec![0.0; m] };
    let tau = r.f32s(inst);
    let clip_id = r.u32s(inst);
    let host_mw = if has_mw { Some(r.f32s(inst * m)) } else { None };
    let root = if has_root { Some(r.f32s(inst * 16)) } else { None };
    let want_key = r.u32s(inst * cmax);
    let want_world = r.f32s(inst * n * 12);
    let want_palette = r.f32s(inst * j * 24);
    let want_pos = r.f32s(inst * v * 3);
    let want_nrm = r.f32s(inst * v * 3);
    let want_tan = r.f32s(inst * v * 3);
    let want_aabb = r.f32s(inst * 6);
    assert_eq!(r.at, r.data.len(), "trailing bytes in the fixture");

    let mut got_key = vec![u32::MAX; inst * cmax];
    let mut got_world = vec![0.0f32; inst * n * 12];
    let mut got_palette = vec![0.0f32; inst * j * 24];
    let mut got_pos = vec![0.0f32; inst * v * 3];
    let mut got_nrm = vec![0.0f32; inst * v * 3];
    let mut got_tan = vec![0.0f32; inst * v * 3];
    let mut got_aabb = vec![0.0f32; inst * 6];

    for i in 0..inst {
        // a6: local transforms as the loader hands them over
        let mut local: Vec<NodeTransform> = (0..n)
            .map(|q| {
                if node_is_matrix[q] != 0 {
                    NodeTransform::Matrix(MatrixNodeTransform(Mat4::from_cols_slice(&matrix[q * 16..q * 16 + 16])))
                } else {
                    NodeTransform::Decomposed(DecomposedTransform {
                        translation: Vec3::from_slice(&translation[q * 3..]),
                        rotation: Quat::from_slice(&rotation[q * 4..]),
                        scale: Vec3::from_slice(&scale[q * 3..]),
                    })
                }
            })
            .collect();
        let mut anim_weights: Option<Vec<f32>> = None;
        let clip = clip_id[i] as usize;
        if clip < clips {
            for (c, ch) in clip_channels[clip].iter().enumerate() {
                if ch.path == 3 {
                    let mut out = vec![0.0f32; m];
                    let mut key = 0;
                    for t in 0..m {
                        // component t of the M-wide keys, through the same trait
                        let sub = Channel {
                            node: ch.node,
                            path: 3,
                            interp: ch.interp,
                            times: ch.times.clone(),
                            values: ch.values.chunks_exact(m).map(|row| row[t]).collect(),
                        };
                        let (k, val) = sample::<f32>(&sub, 1, tau[i], |s| s[0]);
                        out[t] = val;
                        key = k;
                    }
                    got_key[i * cmax + c] = key as u32;
                    anim_weights = Some(out);
                    continue;
                }
                // an animated node is used in decomposed form (node.rs:73-87: to_scale_rotation_translation)
                let mut d: DecomposedTransform = local[ch.node].clone().into();
                let key = match ch.path {
                    0 => {
                        let (k, val) = sample::<Vec3>(ch, 3, tau[i], Vec3::from_slice);
                        d.translation = val;
                        k
                    }
                    1 => {
                        let (k, val) = sample::<Quat>(ch, 4, tau[i], Quat::from_slice);
                        d.rotation = val; // not normalised: the reference never normalises
                        k
                    }
                    _ => {
                        let (k, val) = sample::<Vec3>(ch, 3, tau[i], Vec3::from_slice);
                        d.scale = val;
                        k
                    }
                };
                got_key[i * cmax + c] = key as u32;
                local[ch.node] = NodeTransform::Decomposed(d);
            }
        }
        // a5 + a7: DFS pre-order (parent index < own index), world = parent world * local
        let mut world = vec![Mat4::IDENTITY; n];
        for q in 0..n {
            let l: Mat4 = local[q].clone().into();
            world[q] = if parent[q] >= 0 {
                world[parent[q] as usize] * l
            } else if let Some(rt) = &root {
                Mat4::from_cols_slice(&rt[i * 16..i * 16 + 16]) * l
            } else {
                l
            };
            let a = world[q].to_cols_array();
            for row in 0..3 {
                for col in 0..4 {
                    got_world[(i * n + q) * 12 + row * 4 + col] = a[col * 4 + row];
                }
            }
        }
        // a8 + a9: joint matrices and SkinMatrix.normal (uniform/skin.rs:30-37)
        let mut joint_m = vec![Mat4::IDENTITY; j];
        let mut joint_n = vec![Mat3::IDENTITY; j];
        for q in 0..j {
            let wj = world[joint_node[q] as usize];
            joint_m[q] = if has_ibm { wj * Mat4::from_cols_slice(&ibm[q * 16..q * 16 + 16]) } else { wj };
            joint_n[q] = Mat3::from_mat4(joint_m[q].inverse().transpose());
            let a = joint_m[q].to_cols_array();
            let b = joint_n[q].to_cols_array();
            let o = (i * j + q) * 24;
            for row in 0..3 {
                for col in 0..4 {
                    got_palette[o + row * 4 + col] = a[col * 4 + row];
                }
                for col in 0..3 {
                    got_palette[o + 12 + row * 4 + col] = b[col * 3 + row];
                }
            }
        }
        // a13 weights: channel sample, else host, else defaults
        let mw: Vec<f32> = match (&anim_weights, &host_mw) {
            (Some(a), _) => a.clone(),
            (None, Some(hw)) => hw[i * m..(i + 1) * m].to_vec(),
            _ => default_weights.clone(),
        };
        let mut cursor = vec![0usize; m];
        let (mut lo, mut hi) = ([f32::INFINITY; 3], [f32::NEG_INFINITY; 3]);
        for q in 0..v {
            let mut p = Vec3::from_slice(&position[q * 3..]);
            let mut nn = Vec3::from_slice(&normal[q * 3..]);
            let mut tt = Vec3::from_slice(&tangent[q * ts..]);
            for (t, tg) in targets.iter().enumerate() {
                let row = match &tg.indices {
                    Some(ix) => {
                        if cursor[t] < ix.len() && ix[cursor[t]] as usize == q {
                            cursor[t] += 1;
                            cursor[t] - 1
                        } else {
                            continue;
                        }
                    }
                    None => q,
                };
                // p = p + w * delta, targets ascending (component-wise mul then add: two roundings, as the C oracle)
                p = p + mw[t] * Vec3::from_slice(&tg.position[row * 3..]);
                if let Some(d) = &tg.normal {
                    nn = nn + mw[t] * Vec3::from_slice(&d[row * 3..]);
                }
                if let Some(d) = &tg.tangent {
                    tt = tt + mw[t] * Vec3::from_slice(&d[row * ts..]);
                }
            }
            // a10: shader.wgsl:127-141 — w.x * M[j.x] + w.y * M[j.y] + w.z * M[j.z] + w.w * M[j.w], further sets continue the chain
            let mut s = Mat4::ZERO;
            let mut nm = Mat3::ZERO;
            let mut first = true;
            for set in 0..k_sets {
                for a in 0..4 {
                    let ji = joints[set][q * 4 + a] as usize;
                    let wt = weights[set][q * 4 + a];
                    if first {
                        s = joint_m[ji] * wt;
                        nm = joint_n[ji] * wt;
                        first = false;
                    } else {
                        s = s + joint_m[ji] * wt;
                        nm = nm + joint_n[ji] * wt;
                    }
                }
            }
            let op = (s * Vec4::new(p.x, p.y, p.z, 1.0)).truncate(); // shader.wgsl:159-163
            let mut on = nm * nn;
            let mut ot = nm * tt;
            if normalize {
                on = on / on.dot(on).sqrt(); // shader.wgsl:217-218: normalize(v) = v / length(v)
                ot = ot / ot.dot(ot).sqrt();
            }
            let o = (i * v + q) * 3;
            got_pos[o..o + 3].copy_from_slice(&op.to_array());
            got_nrm[o..o + 3].copy_from_slice(&on.to_array());
            got_tan[o..o + 3].copy_from_slice(&ot.to_array());
            for e in 0..3 {
                lo[e] = lo[e].min(op[e]);
                hi[e] = hi[e].max(op[e]);
            }
        }
        got_aabb[i * 6..i * 6 + 3].copy_from_slice(&lo);
        got_aabb[i * 6 + 3..i * 6 + 6].copy_from_slice(&hi);
    }

    let key_bad = got_key.iter().zip(&want_key).filter(|(a, b)| a != b).count();
    println!("key_index   {} of {} differ", key_bad, want_key.len());
    let mut failed = key_bad != 0;
    for d in [
        compare("world", &got_world, &want_world),
        compare("palette", &got_palette, &want_palette),
        compare("position", &got_pos, &want_pos),
        compare("normal", &got_nrm, &want_nrm),
        compare("tangent", &got_tan, &want_tan),
        compare("aabb", &got_aabb, &want_aabb),
    ] {
        println!("{:11} {} of {} differ, max |diff| {:e}", d.name, d.bad, d.total, d.max_abs);
        failed |= d.bad != 0;
    }
    if failed {
        println!("MISMATCH: the C oracle and the reference's arithmetic disagree on this fixture (see the counts above)");
        exit(1);
    }
    println!("PINNED: bit-equal to the reference's Interpolate / NodeTransform / glam arithmetic on {path}");
}
