/*
 * skin_oracle.c — CPU restatement of the reference's skinned-scene evaluation path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under renderer_amd/ (the product) may call, link or
 * import this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED: the reference (fifth-light/renderer @ 2024-12-20) is Rust + WGSL, there is
 * no cargo/rustc/wgpu in this image, and the reference holds no test, golden vector or fixture
 * for this path.  This file follows the reference's own arithmetic where
 * it exists (file:line cited per function, under /root/reference/crates/) and, for the glue the
 * reference does not implement (keyframe search, hierarchy walk, world*IBM, morph blend, AABB),
 * the rules frozen in SURVEY.md §8a rows a4, a7, a8, a13, a14.  Third-party arithmetic is
 * glam 0.29.2 (Cargo.lock:1343-1344), absent from /root/reference and from this image; its
 * algorithms (Mat4::from_quat, Mat4*Mat4, Mat4::inverse, to_scale_rotation_translation,
 * Quat::from_rotation_axes, scalar code path) are restated from the published source as
 * recalled — not verifiable offline.  glam's SSE2 path computes the same products and may
 * associate one 4-term sum differently (dot4 in inverse); that is inside the fp32 tolerance
 * the parity tests use (1e-5 rel / 1e-6 abs).
 *
 * Build twice (oracle/Makefile): REAL=float  with -O2 -ffp-contract=off  -> the parity oracle and
 * the timed CPU baseline; REAL=double -> error-budget second opinion.  Operation order is the
 * reference's; nothing is re-associated.
 *
 * Input format: skinrt_rig_desc of include/skinrt.h (the flat loader output).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/skinrt.h"

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------ */
/* Interpolate::linear — renderer-asset/src/animation.rs:51-53:  a * (1.0 - t) + b * t   */
ORC_API void orc_linear(const real* a, const real* b, real t, int w, real* out) {
    real omt = (real)1.0 - t;
    for (int i = 0; i < w; ++i) out[i] = a[i] * omt + b[i] * t;
}

/* Interpolate::cubic_spline — renderer-asset/src/animation.rs:55-63.
 * quirk=1: t2 = t.powi(3) exactly as written at :57.  quirk=0: t2 = t*t (glTF spec).
 * powi(3) is restated as (t*t)*t (LLVM's powi expansion for a constant exponent 3). */
ORC_API void orc_cubic_spline(const real* vk, const real* bk, const real* vk1, const real* ak1,
                              real t, real td, int w, int quirk, real* out) {
    real t3 = (t * t) * t;
    real t2 = quirk ? t3 : t * t;
    real c1 = (real)2.0 * t3 - (real)3.0 * t2 + (real)1.0;
    real c2 = t3 - (real)2.0 * t2 + t;
    real c3 = (real)-2.0 * t3 + (real)3.0 * t2;
    real c4 = t3 - t2;
    for (int i = 0; i < w; ++i) {
        real first = vk[i] * c1;
        real second = bk[i] * td * c2; /* (bk * td) * c2 — Mul is left-assoc */
        real third = vk1[i] * c3;
        real forth = ak1[i] * td * c4;
        out[i] = first + second + third + forth; /* ((first+second)+third)+forth */
    }
}

/* Keyframe search — MISSING in the reference; rule SURVEY.md §8a a4.
 * Returns key k; *u = interpolation parameter (0 when clamped). *clamped=1 when no
 * interpolation happens (single key, tau before first / after last key). */
ORC_API uint32_t orc_find_key(const float* times, uint32_t n, real tau, real* u, int* clamped) {
    *u = 0;
    *clamped = 1;
    if (n == 1 || tau <= (real)times[0]) return 0;
    if (tau >= (real)times[n - 1]) return n - 1;
    uint32_t lo = 0, hi = n - 1; /* invariant: times[lo] <= tau < times[hi] */
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if ((real)times[mid] <= tau) lo = mid; else hi = mid;
    }
    *clamped = 0;
    *u = (tau - (real)times[lo]) / ((real)times[lo + 1] - (real)times[lo]);
    return lo;
}

/* Sample one channel at tau -> out[w].  a1-a4. */
static uint32_t sample_channel(const skinrt_channel_desc* ch, real tau, int quirk, real* out) {
    int w = ch->path == SKINRT_PATH_ROTATION ? 4 : 3;
    real u; int clamped;
    uint32_t k = orc_find_key(ch->times, ch->key_count, tau, &u, &clamped);
    const float* v = ch->values;
    real a[4], b[4], c[4], d[4];
    if (ch->interpolation == SKINRT_INTERP_CUBICSPLINE) {
        /* (in, value, out) triples — animation.rs:18-19, loader/gltf/mod.rs:1251-1266 */
        if (clamped) { for (int i = 0; i < w; ++i) out[i] = (real)v[(k * 3 + 1) * w + i]; return k; }
        for (int i = 0; i < w; ++i) {
            a[i] = (real)v[(k * 3 + 1) * w + i];       /* vk   */
            b[i] = (real)v[(k * 3 + 2) * w + i];       /* bk  = out tangent of k   */
            c[i] = (real)v[((k + 1) * 3 + 1) * w + i]; /* vk+1 */
            d[i] = (real)v[((k + 1) * 3 + 0) * w + i]; /* ak+1 = in tangent of k+1 */
        }
        real td = (real)ch->times[k + 1] - (real)ch->times[k];
        orc_cubic_spline(a, b, c, d, u, td, w, quirk, out);
        return k;
    }
    if (clamped || ch->interpolation == SKINRT_INTERP_STEP) {
        for (int i = 0; i < w; ++i) out[i] = (real)v[k * w + i];
        return k;
    }
    for (int i = 0; i < w; ++i) { a[i] = (real)v[k * w + i]; b[i] = (real)v[(k + 1) * w + i]; }
    orc_linear(a, b, u, w, out);
    return k;
}

/* glTF "weights" channel (SKINRT_PATH_WEIGHTS) at tau -> out[M].  The reference's loader stops at todo!() for
 * Property::MorphTargetWeights (loader/gltf/mod.rs:1312), so there is nothing to restate: the rule is the one the
 * other channels follow — same key search (a4), Interpolate::linear / cubic_spline (animation.rs:51-63) applied to
 * each of the M components (the trait is component-wise for every T it is used with), values laid out
 * [key][M] or [key][in, value, out][M]. */
ORC_API uint32_t orc_sample_weights(const skinrt_channel_desc* ch, uint32_t M, real tau, int quirk, real* out) {
    real u; int clamped;
    uint32_t k = orc_find_key(ch->times, ch->key_count, tau, &u, &clamped);
    const float* v = ch->values;
    for (uint32_t m = 0; m < M; ++m) {
        if (ch->interpolation == SKINRT_INTERP_CUBICSPLINE) {
            if (clamped) { out[m] = (real)v[((size_t)k * 3 + 1) * M + m]; continue; }
            real a = (real)v[((size_t)k * 3 + 1) * M + m], b = (real)v[((size_t)k * 3 + 2) * M + m];
            real c = (real)v[((size_t)(k + 1) * 3 + 1) * M + m], d = (real)v[((size_t)(k + 1) * 3 + 0) * M + m];
            real td = (real)ch->times[k + 1] - (real)ch->times[k];
            orc_cubic_spline(&a, &b, &c, &d, u, td, 1, quirk, &out[m]);
        } else if (clamped || ch->interpolation == SKINRT_INTERP_STEP) {
            out[m] = (real)v[(size_t)k * M + m];
        } else {
            real a = (real)v[(size_t)k * M + m], b = (real)v[(size_t)(k + 1) * M + m];
            orc_linear(&a, &b, u, 1, &out[m]);
        }
    }
    return k;
}

/* ------------------------------------------------------------------------------------ */
/* glam Mat4 helpers; m[c*4+r] column-major (Mat4::to_cols_array).                        */

/* Mat4 * Mat4 — glam mul_mat4: column c of the result = self.mul_vec4(rhs.col(c)),
 * mul_vec4 = ((x_axis*v.x + y_axis*v.y) + z_axis*v.z) + w_axis*v.w.  Used at node.rs:49-51. */
ORC_API void orc_mat4_mul(const real* a, const real* b, real* out) {
    real r[16];
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row) {
            real s = a[0 * 4 + row] * b[c * 4 + 0];
            s = s + a[1 * 4 + row] * b[c * 4 + 1];
            s = s + a[2 * 4 + row] * b[c * 4 + 2];
            s = s + a[3 * 4 + row] * b[c * 4 + 3];
            r[c * 4 + row] = s;
        }
    memcpy(out, r, sizeof r);
}

static void mat4_identity(real* m) {
    memset(m, 0, 16 * sizeof(real));
    m[0] = m[5] = m[10] = m[15] = (real)1.0;
}

/* impl From<DecomposedTransform> for Mat4 — renderer-asset/src/node.rs:47-53:
 *   Mat4::from_translation(t) * Mat4::from_quat(q) * Mat4::from_scale(s)
 * glam from_quat (quat_to_axes): x2=x+x ...; xx=x*x2 ...; quaternion used un-normalised. */
ORC_API void orc_trs_to_mat4(const real* t, const real* q, const real* s, real* out) {
    real T[16], R[16], S[16], TR[16];
    mat4_identity(T); T[12] = t[0]; T[13] = t[1]; T[14] = t[2];
    real x = q[0], y = q[1], z = q[2], w = q[3];
    real x2 = x + x, y2 = y + y, z2 = z + z;
    real xx = x * x2, xy = x * y2, xz = x * z2;
    real yy = y * y2, yz = y * z2, zz = z * z2;
    real wx = w * x2, wy = w * y2, wz = w * z2;
    mat4_identity(R);
    R[0] = (real)1.0 - (yy + zz); R[1] = xy + wz;              R[2] = xz - wy;
    R[4] = xy - wz;              R[5] = (real)1.0 - (xx + zz); R[6] = yz + wx;
    R[8] = xz + wy;              R[9] = yz - wx;              R[10] = (real)1.0 - (xx + yy);
    mat4_identity(S); S[0] = s[0]; S[5] = s[1]; S[10] = s[2];
    orc_mat4_mul(T, R, TR);
    orc_mat4_mul(TR, S, out);
}

/* glam Mat4::inverse (scalar path: cofactors, then * 1/det). Used at uniform/skin.rs:31. */
ORC_API void orc_mat4_inverse(const real* m, real* out) {
    real m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3];
    real m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
    real m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11];
    real m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];

    real coef00 = m22 * m33 - m32 * m23;
    real coef02 = m12 * m33 - m32 * m13;
    real coef03 = m12 * m23 - m22 * m13;
    real coef04 = m21 * m33 - m31 * m23;
    real coef06 = m11 * m33 - m31 * m13;
    real coef07 = m11 * m23 - m21 * m13;
    real coef08 = m21 * m32 - m31 * m22;
    real coef10 = m11 * m32 - m31 * m12;
    real coef11 = m11 * m22 - m21 * m12;
    real coef12 = m20 * m33 - m30 * m23;
    real coef14 = m10 * m33 - m30 * m13;
    real coef15 = m10 * m23 - m20 * m13;
    real coef16 = m20 * m32 - m30 * m22;
    real coef18 = m10 * m32 - m30 * m12;
    real coef19 = m10 * m22 - m20 * m12;
    real coef20 = m20 * m31 - m30 * m21;
    real coef22 = m10 * m31 - m30 * m11;
    real coef23 = m10 * m21 - m20 * m11;

    real fac0[4] = {coef00, coef00, coef02, coef03};
    real fac1[4] = {coef04, coef04, coef06, coef07};
    real fac2[4] = {coef08, coef08, coef10, coef11};
    real fac3[4] = {coef12, coef12, coef14, coef15};
    real fac4[4] = {coef16, coef16, coef18, coef19};
    real fac5[4] = {coef20, coef20, coef22, coef23};
    real vec0[4] = {m10, m00, m00, m00};
    real vec1[4] = {m11, m01, m01, m01};
    real vec2[4] = {m12, m02, m02, m02};
    real vec3[4] = {m13, m03, m03, m03};
    static const real sign_a[4] = {1, -1, 1, -1};
    static const real sign_b[4] = {-1, 1, -1, 1};
    real inv[16];
    for (int i = 0; i < 4; ++i) {
        real inv0 = (vec1[i] * fac0[i] - vec2[i] * fac1[i]) + vec3[i] * fac2[i];
        real inv1 = (vec0[i] * fac0[i] - vec2[i] * fac3[i]) + vec3[i] * fac4[i];
        real inv2 = (vec0[i] * fac1[i] - vec1[i] * fac3[i]) + vec3[i] * fac5[i];
        real inv3 = (vec0[i] * fac2[i] - vec1[i] * fac4[i]) + vec2[i] * fac5[i];
        inv[0 * 4 + i] = inv0 * sign_a[i];
        inv[1 * 4 + i] = inv1 * sign_b[i];
        inv[2 * 4 + i] = inv2 * sign_a[i];
        inv[3 * 4 + i] = inv3 * sign_b[i];
    }
    /* col0 = first row of the adjugate; dot with self.x_axis */
    real d0 = m00 * inv[0], d1 = m01 * inv[4], d2 = m02 * inv[8], d3 = m03 * inv[12];
    real dot1 = ((d0 + d1) + d2) + d3;
    real rcp_det = (real)1.0 / dot1;
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * rcp_det;
}

/* SkinUniformBuffer::calculate_normal_transform — renderer/src/renderer/uniform/skin.rs:30-37
 * (same math: uniform/transform.rs:27-34): Mat3::from_mat4(m.inverse().transpose()), columns
 * padded to vec4 with 0.  out[3][4]. */
ORC_API void orc_normal_transform(const real* m, real* out12) {
    real inv[16];
    orc_mat4_inverse(m, inv);
    /* transpose: T[c*4+r] = inv[r*4+c]; Mat3::from_mat4 takes the upper-left 3x3 */
    for (int c = 0; c < 3; ++c) {
        for (int r = 0; r < 3; ++r) out12[c * 4 + r] = inv[r * 4 + c];
        out12[c * 4 + 3] = (real)0.0;
    }
}

/* glam Mat4::determinant (scalar path) — needed by to_scale_rotation_translation. */
static real mat4_determinant(const real* m) {
    real m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3];
    real m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
    real m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11];
    real m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];
    real a2323 = m22 * m33 - m23 * m32;
    real a1323 = m21 * m33 - m23 * m31;
    real a1223 = m21 * m32 - m22 * m31;
    real a0323 = m20 * m33 - m23 * m30;
    real a0223 = m20 * m32 - m22 * m30;
    real a0123 = m20 * m31 - m21 * m30;
    return m00 * (m11 * a2323 - m12 * a1323 + m13 * a1223)
         - m01 * (m10 * a2323 - m12 * a0323 + m13 * a0223)
         + m02 * (m10 * a1323 - m11 * a0323 + m13 * a0123)
         - m03 * (m10 * a1223 - m11 * a0223 + m12 * a0123);
}

static real rsqrt_half(real four_sq) {
#if defined(ORC_DOUBLE)
    return (real)0.5 / sqrt(four_sq);
#else
    return (real)0.5 / sqrtf(four_sq);
#endif
}
static real vlen3(const real* v) {
#if defined(ORC_DOUBLE)
    return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
#else
    return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
#endif
}

/* impl From<NodeTransform> for DecomposedTransform (Matrix arm) — renderer-asset/src/node.rs:73-87:
 * glam Mat4::to_scale_rotation_translation + Quat::from_rotation_axes. */
ORC_API void orc_decompose(const real* m, real* t, real* q, real* s) {
    real det = mat4_determinant(m);
    real sg = signbit(det) ? (real)-1.0 : (real)1.0; /* f32::signum: +-1 by sign bit (det != 0 is glam_assert'ed) */
    s[0] = vlen3(m + 0) * sg; s[1] = vlen3(m + 4); s[2] = vlen3(m + 8);
    real is0 = (real)1.0 / s[0], is1 = (real)1.0 / s[1], is2 = (real)1.0 / s[2];
    real m00 = m[0] * is0, m01 = m[1] * is0, m02 = m[2] * is0;
    real m10 = m[4] * is1, m11 = m[5] * is1, m12 = m[6] * is1;
    real m20 = m[8] * is2, m21 = m[9] * is2, m22 = m[10] * is2;
    if (m22 <= 0) {
        real dif10 = m11 - m00, omm22 = (real)1.0 - m22;
        if (dif10 <= 0) {
            real f = omm22 - dif10, i4 = rsqrt_half(f);
            q[0] = f * i4; q[1] = (m01 + m10) * i4; q[2] = (m02 + m20) * i4; q[3] = (m12 - m21) * i4;
        } else {
            real f = omm22 + dif10, i4 = rsqrt_half(f);
            q[0] = (m01 + m10) * i4; q[1] = f * i4; q[2] = (m12 + m21) * i4; q[3] = (m20 - m02) * i4;
        }
    } else {
        real sum10 = m11 + m00, opm22 = (real)1.0 + m22;
        if (sum10 <= 0) {
            real f = opm22 - sum10, i4 = rsqrt_half(f);
            q[0] = (m02 + m20) * i4; q[1] = (m12 + m21) * i4; q[2] = f * i4; q[3] = (m01 - m10) * i4;
        } else {
            real f = opm22 + sum10, i4 = rsqrt_half(f);
            q[0] = (m12 - m21) * i4; q[1] = (m20 - m02) * i4; q[2] = (m01 - m10) * i4; q[3] = f * i4;
        }
    }
    t[0] = m[12]; t[1] = m[13]; t[2] = m[14];
}

/* ------------------------------------------------------------------------------------ */
/* WGSL skinning — renderer/src/shader/shader.wgsl:127-141 (matrix blend), :159-163 / :223-226
 * (matrix * vector).  pal: [J][28] = SkinMatrix (mat4 col-major, then 3 padded normal columns).
 * S = ((w.x*M[j.x] + w.y*M[j.y]) + w.z*M[j.z]) + w.w*M[j.w]; further weight sets (K>1, which
 * the reference shader cannot take) continue the same left-to-right chain.
 * mat*vec = ((c0*x + c1*y) + c2*z) + c3*w. */
ORC_API void orc_skin_vertex(const real* pal, uint32_t sets, const uint16_t* const* jidx,
                             const float* const* jw, uint32_t v, const real* p, const real* n,
                             const real* t, int normalize, real* op, real* on, real* ot) {
    real S[16], Nm[12];
    for (uint32_t k = 0; k < sets; ++k)
        for (int c = 0; c < 4; ++c) {
            const real* M = pal + (size_t)jidx[k][v * 4 + c] * 28;
            real w = (real)jw[k][v * 4 + c];
            if (k == 0 && c == 0) {
                for (int e = 0; e < 16; ++e) S[e] = w * M[e];
                for (int e = 0; e < 12; ++e) Nm[e] = w * M[16 + e];
            } else {
                for (int e = 0; e < 16; ++e) S[e] = S[e] + w * M[e];
                for (int e = 0; e < 12; ++e) Nm[e] = Nm[e] + w * M[16 + e];
            }
        }
    for (int r = 0; r < 3; ++r) {
        op[r] = ((S[0 + r] * p[0] + S[4 + r] * p[1]) + S[8 + r] * p[2]) + S[12 + r] * (real)1.0;
        on[r] = (Nm[0 + r] * n[0] + Nm[4 + r] * n[1]) + Nm[8 + r] * n[2];
        ot[r] = (Nm[0 + r] * t[0] + Nm[4 + r] * t[1]) + Nm[8 + r] * t[2];
    }
    if (normalize) { /* shader.wgsl:217-218 */
        real ln = vlen3(on), lt = vlen3(ot);
        for (int r = 0; r < 3; ++r) { on[r] = on[r] / ln; ot[r] = ot[r] / lt; }
    }
}

/* ------------------------------------------------------------------------------------ */
/* calculate_tangent — renderer-asset/src/tangent.rs:15-58 (load time).
 * calculate_triangle_tangent (:5-12): normalize((p2 - p1) x (p0 - p1)).  glam 0.29 scalar Vec3:
 *   cross     = (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)
 *   normalize = v * (1 / sqrt((v.x*v.x + v.y*v.y) + v.z*v.z))            (mul by length_recip)
 * Triangles in order, each unit normal ADDED to its three vertices in corner order (:38, :49; a vertex
 * named twice gets it twice), then every vertex sum normalised (:54-56; zero sums give NaN).
 * mode: PrimitiveAssetMode ordinal; 0-2 (points, lines) have no faces (:23-25); 3 (TriangleStrip) is
 * todo!() (:26) -> returns -1; an index >= vertices is a panic in the reference -> returns -2. */
static real rlen_recip(const real* v) {
#if defined(ORC_DOUBLE)
    return (real)1.0 / sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
#else
    return (real)1.0 / sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
#endif
}
ORC_API int orc_calculate_tangent(uint32_t mode, const float* positions, uint32_t vertices, const uint32_t* indices,
                                  uint32_t index_count, real* out) {
    if (mode == 3) return -1;
    for (size_t i = 0; i < (size_t)vertices * 3; ++i) out[i] = (real)0.0;
    if (mode == 4) {
        uint32_t triangles = (indices ? index_count : vertices) / 3;
        for (uint32_t f = 0; f < triangles; ++f) {
            uint32_t ix[3];
            for (int k = 0; k < 3; ++k) {
                ix[k] = indices ? indices[3 * f + k] : 3 * f + k;
                if (ix[k] >= vertices) return -2;
            }
            real p0[3], p1[3], p2[3], a[3], b[3], c[3];
            for (int e = 0; e < 3; ++e) {
                p0[e] = (real)positions[(size_t)ix[0] * 3 + e];
                p1[e] = (real)positions[(size_t)ix[1] * 3 + e];
                p2[e] = (real)positions[(size_t)ix[2] * 3 + e];
                a[e] = p2[e] - p1[e];
                b[e] = p0[e] - p1[e];
            }
            c[0] = a[1] * b[2] - b[1] * a[2];
            c[1] = a[2] * b[0] - b[2] * a[0];
            c[2] = a[0] * b[1] - b[0] * a[1];
            real r = rlen_recip(c);
            real n[3] = {c[0] * r, c[1] * r, c[2] * r};
            for (int k = 0; k < 3; ++k)
                for (int e = 0; e < 3; ++e) out[(size_t)ix[k] * 3 + e] = out[(size_t)ix[k] * 3 + e] + n[e];
        }
    }
    for (uint32_t v = 0; v < vertices; ++v) {
        real r = rlen_recip(out + (size_t)v * 3);
        for (int e = 0; e < 3; ++e) out[(size_t)v * 3 + e] = out[(size_t)v * 3 + e] * r;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
typedef struct orc_crowd_in {
    uint32_t instances;
    const float* tau;            /* [I] */
    const uint32_t* clip_id;     /* [I] or NULL (= clip 0); UINT32_MAX = rest pose */
    const float* morph_weights;  /* [I][M] or NULL (= rig defaults) */
    const float* root;           /* [I][16] or NULL */
    uint32_t cubic_quirk;
    uint32_t normalize_outputs;
} orc_crowd_in;

typedef struct orc_out {         /* any pointer may be NULL = not wanted */
    real* world;                 /* [I][N][16] column-major */
    real* palette;               /* [I][J][28]  SkinMatrix layout */
    uint32_t* key_index;         /* [I][Cmax] */
    real* position;              /* [I][V][3] */
    real* normal;                /* [I][V][3] */
    real* tangent;               /* [I][V][3] */
    real* aabb;                  /* [I][2][3] */
} orc_out;

ORC_API uint32_t orc_real_size(void) { return (uint32_t)sizeof(real); }

static uint32_t max_channels(const skinrt_rig_desc* rig) {
    uint32_t m = 0;
    for (uint32_t c = 0; c < rig->clip_count; ++c)
        if (rig->clips[c].channel_count > m) m = rig->clips[c].channel_count;
    return m;
}

/* ---- skins and sections (skinrt.h ABI 6): the short forms resolved exactly as the header states them ------------- */
typedef struct orc_layout {
    uint32_t skin_count, section_count, rigid_count, joints, palette; /* palette = joints + rigid_count */
    skinrt_skin_desc* skins;
    uint32_t* skin_first;            /* first palette entry of each skin */
    skinrt_section_desc* sections;
    uint32_t* section_entry;         /* skinned: its skin's first entry; un-skinned: the entry of its node */
    uint32_t* rigid_node;            /* [rigid_count] */
    int one_section;                 /* short form: weights channels are not matched by node */
} orc_layout;

static void layout_free(orc_layout* L) {
    free(L->skins); free(L->skin_first); free(L->sections); free(L->section_entry); free(L->rigid_node);
}

static int layout_build(const skinrt_rig_desc* rig, orc_layout* L) {
    memset(L, 0, sizeof *L);
    L->skin_count = rig->skin_count ? rig->skin_count : (rig->joint_count ? 1u : 0u);
    L->skins = (skinrt_skin_desc*)calloc(L->skin_count ? L->skin_count : 1, sizeof(skinrt_skin_desc));
    L->skin_first = (uint32_t*)calloc(L->skin_count ? L->skin_count : 1, sizeof(uint32_t));
    L->one_section = rig->section_count == 0;
    L->section_count = rig->section_count ? rig->section_count : ((rig->vertex_count || rig->target_count) ? 1u : 0u);
    L->sections = (skinrt_section_desc*)calloc(L->section_count ? L->section_count : 1, sizeof(skinrt_section_desc));
    L->section_entry = (uint32_t*)calloc(L->section_count ? L->section_count : 1, sizeof(uint32_t));
    L->rigid_node = (uint32_t*)calloc(L->section_count ? L->section_count : 1, sizeof(uint32_t));
    if (!L->skins || !L->skin_first || !L->sections || !L->section_entry || !L->rigid_node) return SKINRT_E_NOMEM;
    if (rig->skin_count) memcpy(L->skins, rig->skins, rig->skin_count * sizeof(skinrt_skin_desc));
    else if (rig->joint_count) { L->skins[0].joint_count = rig->joint_count; L->skins[0].joint_node = rig->joint_node; L->skins[0].inverse_bind = rig->inverse_bind; }
    for (uint32_t k = 0; k < L->skin_count; ++k) { L->skin_first[k] = L->joints; L->joints += L->skins[k].joint_count; }
    if (rig->section_count) memcpy(L->sections, rig->sections, rig->section_count * sizeof(skinrt_section_desc));
    else if (L->section_count) {
        skinrt_section_desc one = {0u, L->skin_count ? 0 : -1, 0u, rig->vertex_count, 0u, rig->target_count};
        L->sections[0] = one;
    }
    for (uint32_t k = 0; k < L->section_count; ++k) {
        const skinrt_section_desc* sc = &L->sections[k];
        if (sc->skin >= 0) { L->section_entry[k] = L->skin_first[sc->skin]; continue; }
        if (!sc->vertex_count) continue;
        uint32_t r = 0;
        while (r < L->rigid_count && L->rigid_node[r] != sc->node) ++r;
        if (r == L->rigid_count) L->rigid_node[L->rigid_count++] = sc->node;
        L->section_entry[k] = L->joints + r;
    }
    L->palette = L->joints + L->rigid_count;
    return SKINRT_OK;
}

/* Number of palette entries orc_evaluate writes per instance (orc_out.palette is [I][that][28]). */
ORC_API uint32_t orc_palette_count(const skinrt_rig_desc* rig) {
    orc_layout L;
    uint32_t n = 0;
    if (layout_build(rig, &L) == SKINRT_OK) n = L.palette;
    layout_free(&L);
    return n;
}

/* InstanceUniform — renderer/src/renderer/uniform/transform.rs:8-14, :36-42: transform.to_cols_array_2d(), then
 * calculate_normal_transform (:27-34), then 12 bytes of padding.  out31: 16 + 12 + 3 values. */
ORC_API void orc_instance_uniform(const real* world16, real* out31) {
    memcpy(out31, world16, 16 * sizeof(real));
    orc_normal_transform(world16, out31 + 16);
    out31[28] = out31[29] = out31[30] = (real)0.0;
}

/* Un-skinned vertex — renderer/src/shader/shader.wgsl:143-152 (color_vs_main; texture_vs_main :167-176 is the same):
 * instance.transform * vec4(position, 1), instance.normal * normal, instance.normal * tangent;
 * mat*vec = ((c0*x + c1*y) + c2*z) + c3*w.  The outline variants normalize() (:198-199). */
ORC_API void orc_rigid_vertex(const real* iu31, const real* p, const real* n, const real* t, int normalize, real* op, real* on, real* ot) {
    const real* S = iu31;
    const real* Nm = iu31 + 16;
    for (int r = 0; r < 3; ++r) {
        op[r] = ((S[0 + r] * p[0] + S[4 + r] * p[1]) + S[8 + r] * p[2]) + S[12 + r] * (real)1.0;
        on[r] = (Nm[0 + r] * n[0] + Nm[4 + r] * n[1]) + Nm[8 + r] * n[2];
        ot[r] = (Nm[0 + r] * t[0] + Nm[4 + r] * t[1]) + Nm[8 + r] * t[2];
    }
    if (normalize) {
        real ln = vlen3(on), lt = vlen3(ot);
        for (int r = 0; r < 3; ++r) { on[r] = on[r] / ln; ot[r] = ot[r] / lt; }
    }
}

/* Evaluate instances [begin, end).  Output arrays are indexed by absolute instance.
 * vertex_step > 1 (orc_evaluate_sampled): positions / normals / tangents are computed only for the vertices v of
 * instance i with v % vertex_step == (vertex_phase + i) % vertex_step — the phase walks with the instance, so a crowd
 * covers every vertex — and are written COMPACTED, row v / vertex_step of [I][ceil(V / vertex_step)][3]; bounds are
 * not computed.  The full-size parity tests' way to meet every tile of every instance in bounded time and memory. */
static int evaluate_impl(const skinrt_rig_desc* rig, const orc_crowd_in* in, const orc_out* out,
                         uint32_t begin, uint32_t end, uint32_t vertex_step, uint32_t vertex_phase) {
    const uint32_t N = rig->node_count, V = rig->vertex_count;
    const uint32_t M = rig->target_count, K = rig->set_count, Cmax = max_channels(rig);
    const uint32_t ts = rig->tangent_stride ? rig->tangent_stride : 3;
    orc_layout L;
    if (layout_build(rig, &L) != SKINRT_OK) { layout_free(&L); return SKINRT_E_NOMEM; }
    const uint32_t J = L.palette;
    real* trs = (real*)malloc((size_t)N * 10 * sizeof(real));
    unsigned char* is_trs = (unsigned char*)malloc(N ? N : 1);
    real* world = (real*)malloc((size_t)(N ? N : 1) * 16 * sizeof(real));
    real* pal = (real*)malloc((size_t)(J ? J : 1) * 28 * sizeof(real));
    real* mw = (real*)malloc((size_t)(M ? M : 1) * sizeof(real));
    real* mw_anim = (real*)malloc((size_t)(M ? M : 1) * sizeof(real));
    unsigned char* mw_have = (unsigned char*)malloc(M ? M : 1);
    uint32_t* cur = (uint32_t*)malloc((M ? M : 1) * sizeof(uint32_t));
    uint32_t* target_section = (uint32_t*)calloc(M ? M : 1, sizeof(uint32_t));
    if (!trs || !is_trs || !world || !pal || !mw || !mw_anim || !mw_have || !cur || !target_section) return SKINRT_E_NOMEM;
    for (uint32_t s = 0; s < L.section_count; ++s)
        for (uint32_t m = 0; m < L.sections[s].target_count; ++m) target_section[L.sections[s].first_target + m] = s;

    for (uint32_t i = begin; i < end; ++i) {
        /* -- a6: local transform selection ------------------------------------------- */
        for (uint32_t n = 0; n < N; ++n) {
            is_trs[n] = !(rig->has_matrix && rig->has_matrix[n]);
            for (int e = 0; e < 3; ++e) trs[n * 10 + e] = (real)rig->translation[n * 3 + e];
            for (int e = 0; e < 4; ++e) trs[n * 10 + 3 + e] = (real)rig->rotation[n * 4 + e];
            for (int e = 0; e < 3; ++e) trs[n * 10 + 7 + e] = (real)rig->scale[n * 3 + e];
        }
        uint32_t clip = in->clip_id ? in->clip_id[i] : 0;
        memset(mw_have, 0, M ? M : 1);
        if (out->key_index)
            for (uint32_t c = 0; c < Cmax; ++c) out->key_index[(size_t)i * Cmax + c] = UINT32_MAX;
        if (clip != UINT32_MAX && clip < rig->clip_count) {
            const skinrt_clip_desc* cl = &rig->clips[clip];
            real tau = (real)in->tau[i];
            for (uint32_t c = 0; c < cl->channel_count; ++c) {
                const skinrt_channel_desc* ch = &cl->channels[c];
                uint32_t n = ch->target_node;
                if (ch->path == SKINRT_PATH_WEIGHTS) { /* morph weights of the mesh on that node, not the node's transform */
                    uint32_t first = 0, width = M;
                    if (!L.one_section) {
                        width = 0;
                        for (uint32_t s = 0; s < L.section_count; ++s)
                            if (L.sections[s].node == n && L.sections[s].target_count) { first = L.sections[s].first_target; width = L.sections[s].target_count; break; }
                    }
                    uint32_t kw = orc_sample_weights(ch, width, tau, (int)in->cubic_quirk, mw_anim + first);
                    if (out->key_index) out->key_index[(size_t)i * Cmax + c] = kw;
                    memset(mw_have + first, 1, width);
                    continue;
                }
                if (!is_trs[n]) { /* Matrix node that is animated: decompose once (node.rs:76-82) */
                    real m[16];
                    for (int e = 0; e < 16; ++e) m[e] = (real)rig->matrix[n * 16 + e];
                    orc_decompose(m, &trs[n * 10], &trs[n * 10 + 3], &trs[n * 10 + 7]);
                    is_trs[n] = 1;
                }
                real v[4];
                uint32_t k = sample_channel(ch, tau, (int)in->cubic_quirk, v);
                if (out->key_index) out->key_index[(size_t)i * Cmax + c] = k;
                int off = ch->path == SKINRT_PATH_TRANSLATION ? 0 : (ch->path == SKINRT_PATH_ROTATION ? 3 : 7);
                int w = ch->path == SKINRT_PATH_ROTATION ? 4 : 3;
                for (int e = 0; e < w; ++e) trs[n * 10 + off + e] = v[e];
            }
        }
        /* -- a5 + a7: local matrices and DFS pre-order propagation -------------------- */
        for (uint32_t n = 0; n < N; ++n) {
            real local[16];
            if (is_trs[n]) orc_trs_to_mat4(&trs[n * 10], &trs[n * 10 + 3], &trs[n * 10 + 7], local);
            else for (int e = 0; e < 16; ++e) local[e] = (real)rig->matrix[n * 16 + e];
            int32_t p = rig->parent[n];
            if (p >= 0) orc_mat4_mul(&world[(size_t)p * 16], local, &world[(size_t)n * 16]);
            else if (in->root) {
                real r[16];
                for (int e = 0; e < 16; ++e) r[e] = (real)in->root[(size_t)i * 16 + e];
                orc_mat4_mul(r, local, &world[(size_t)n * 16]);
            } else memcpy(&world[(size_t)n * 16], local, sizeof local);
        }
        if (out->world) memcpy(out->world + (size_t)i * N * 16, world, (size_t)N * 16 * sizeof(real));
        /* -- a8 + a9: joint matrices and SkinMatrix entries, skin after skin (skin.rs:5-11) ----------- */
        for (uint32_t s = 0; s < L.skin_count; ++s)
            for (uint32_t q = 0; q < L.skins[s].joint_count; ++q) {
                real* e = pal + (size_t)(L.skin_first[s] + q) * 28;
                const real* wj = &world[(size_t)L.skins[s].joint_node[q] * 16];
                if (L.skins[s].inverse_bind) {
                    real ibm[16];
                    for (int x = 0; x < 16; ++x) ibm[x] = (real)L.skins[s].inverse_bind[(size_t)q * 16 + x];
                    orc_mat4_mul(wj, ibm, e);
                } else memcpy(e, wj, 16 * sizeof(real));
                orc_normal_transform(e, e + 16);
            }
        /* un-skinned sections: the InstanceUniform of their node, kept behind the joints (28 of its 31 values) */
        for (uint32_t r = 0; r < L.rigid_count; ++r) {
            real iu[31];
            orc_instance_uniform(&world[(size_t)L.rigid_node[r] * 16], iu);
            memcpy(pal + (size_t)(L.joints + r) * 28, iu, 28 * sizeof(real));
        }
        if (out->palette) memcpy(out->palette + (size_t)i * J * 28, pal, (size_t)J * 28 * sizeof(real));
        if (!out->position && !out->normal && !out->tangent && !out->aabb) continue;
        /* -- a13: morph weights: weights channel of the clip, else node override, else mesh default, else 0 -- */
        for (uint32_t m = 0; m < M; ++m)
            mw[m] = mw_have[m] ? mw_anim[m]  /* the clip's weights channel overrides (skinrt.h) */
                  : in->morph_weights ? (real)in->morph_weights[(size_t)i * M + m]
                                      : (rig->default_weights ? (real)rig->default_weights[m] : (real)0.0);
        real lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        /* cursors into compacted targets (indices strictly increasing, counted from the section's first vertex) */
        memset(cur, 0, (M ? M : 1) * sizeof(uint32_t));
        for (uint32_t s = 0; s < L.section_count; ++s) {
            const skinrt_section_desc* sc = &L.sections[s];
            for (uint32_t lv = 0; lv < sc->vertex_count; ++lv) {
                const uint32_t v = sc->first_vertex + lv;
                if (vertex_step > 1 && v % vertex_step != (vertex_phase + i) % vertex_step) continue;
                real p[3], n[3], t[3];
                for (int e = 0; e < 3; ++e) {
                    p[e] = (real)rig->position[(size_t)v * 3 + e];
                    n[e] = (real)rig->normal[(size_t)v * 3 + e];
                    t[e] = (real)rig->tangent[(size_t)v * ts + e];
                }
                /* p = p0 + sum_m w_m * delta_m, the section's targets in order; absent delta = zeros */
                for (uint32_t m = sc->first_target; m < sc->first_target + sc->target_count; ++m) {
                    const skinrt_morph_target_desc* tg = &rig->targets[m];
                    uint32_t row;
                    if (tg->indices && vertex_step > 1) { /* sampled vertices: no cursor to carry, search the row */
                        uint32_t a = 0, b = tg->count;
                        while (a < b) { uint32_t mid = (a + b) >> 1; if (tg->indices[mid] < lv) a = mid + 1; else b = mid; }
                        if (a < tg->count && tg->indices[a] == lv) row = a; else continue;
                    } else if (tg->indices) {
                        if (cur[m] < tg->count && tg->indices[cur[m]] == lv) row = cur[m]++; else continue;
                    } else row = lv;
                    for (int e = 0; e < 3; ++e) {
                        if (tg->position) p[e] = p[e] + mw[m] * (real)tg->position[(size_t)row * 3 + e];
                        if (tg->normal) n[e] = n[e] + mw[m] * (real)tg->normal[(size_t)row * 3 + e];
                        if (tg->tangent) t[e] = t[e] + mw[m] * (real)tg->tangent[(size_t)row * ts + e];
                    }
                }
                real op[3], on[3], ot[3];
                if (sc->skin >= 0)
                    orc_skin_vertex(pal + (size_t)L.section_entry[s] * 28, K, rig->joints, rig->weights, v, p, n, t, (int)in->normalize_outputs, op, on, ot);
                else {
                    real iu[31];
                    orc_instance_uniform(&world[(size_t)sc->node * 16], iu);
                    orc_rigid_vertex(iu, p, n, t, (int)in->normalize_outputs, op, on, ot);
                }
                size_t o = vertex_step > 1 ? ((size_t)i * ((V + vertex_step - 1) / vertex_step) + v / vertex_step) * 3 : ((size_t)i * V + v) * 3;
                for (int e = 0; e < 3; ++e) {
                    if (out->position) out->position[o + e] = op[e];
                    if (out->normal) out->normal[o + e] = on[e];
                    if (out->tangent) out->tangent[o + e] = ot[e];
                    if (op[e] < lo[e]) lo[e] = op[e];
                    if (op[e] > hi[e]) hi[e] = op[e];
                }
            }
        }
        if (out->aabb && vertex_step <= 1) for (int e = 0; e < 3; ++e) { out->aabb[(size_t)i * 6 + e] = lo[e]; out->aabb[(size_t)i * 6 + 3 + e] = hi[e]; }
    }
    free(trs); free(is_trs); free(world); free(pal); free(mw); free(mw_anim); free(mw_have); free(cur); free(target_section);
    layout_free(&L);
    return SKINRT_OK;
}

ORC_API int orc_evaluate(const skinrt_rig_desc* rig, const orc_crowd_in* in, const orc_out* out, uint32_t begin, uint32_t end) {
    return evaluate_impl(rig, in, out, begin, end, 1, 0);
}

ORC_API int orc_evaluate_sampled(const skinrt_rig_desc* rig, const orc_crowd_in* in, const orc_out* out, uint32_t begin, uint32_t end,
                                 uint32_t vertex_step, uint32_t vertex_phase) {
    return evaluate_impl(rig, in, out, begin, end, vertex_step, vertex_phase);
}
