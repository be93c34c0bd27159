// Rotation conversions and the SMPL kinematic chain of the training loss, forward AND reverse mode, as plain
// per-pose functions (no memory traffic, no HIP builtins) so that the same source compiles
//   * as __device__ code into libtcdiff_gfx950.so (csrc/train.hip), and
//   * as host code under g++ (tests/host/fk_host.cpp): the hand-written reverse mode is checked on the CPU against
//     torch autograd through the oracle's restatement (tests/test_train_cpu.py) before it ever runs on a GPU.
//
// What it replaces (reference file:line):
//   ax_from_6v                 dataset/quaternion.py:28-32 -> pytorch3d rotation_6d_to_matrix, matrix_to_axis_angle
//   SMPLSkeleton.forward       vis.py:358-406 -> pytorch3d axis_angle_to_quaternion, quaternion_apply, quaternion_multiply
//   and torch autograd through both (model/diffusion.py:692-733 is differentiated by accelerator.backward, TCDiff.py:232).
// The pytorch3d arithmetic (0.7.1, absent from the reference tree) is restated from its published definitions: real-first
// quaternions, Gram-Schmidt rows b1 b2 b3, candidate quaternion with the largest component, small-angle series
// 0.5 - theta^2 / 48 -- "parity unpinned" (DESIGN.md section 2).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define TC_HD __host__ __device__ __forceinline__
#else
#define TC_HD static inline
#endif

#define TC_FK_J 24

struct Q4 { float w, x, y, z; };
struct V3 { float x, y, z; };

TC_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
TC_HD Q4 q4(float w, float x, float y, float z) { Q4 r; r.w = w; r.x = x; r.y = y; r.z = z; return r; }
TC_HD float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
TC_HD V3 cross3(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
TC_HD V3 add3(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
TC_HD V3 scale3(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
TC_HD Q4 qadd(Q4 a, Q4 b) { return q4(a.w + b.w, a.x + b.x, a.y + b.y, a.z + b.z); }
TC_HD Q4 qconj(Q4 q) { return q4(q.w, -q.x, -q.y, -q.z); }

// ---- forward --------------------------------------------------------------------------------------------------------
TC_HD V3 normalize3(V3 a) {       // F.normalize: v / max(|v|, 1e-12)
    const float n = fmaxf(sqrtf(dot3(a, a)), 1e-12f);
    return v3(a.x / n, a.y / n, a.z / n);
}
TC_HD float sqrt_pos(float v) { return v > 0.0f ? sqrtf(v) : 0.0f; }

// The 3x3 rotation (rows b1, b2, b3) of a 6-D rotation, and which quaternion candidate matrix_to_quaternion selects.
struct Rot6 {
    V3 b1, b2, b3;     // matrix rows
    float dt;          // b1 . a2
    float n1, n2;      // |a1|, |a2 - dt b1| (clamped like F.normalize)
    float qa[4];       // sqrt_positive_part of the four traces
    int best;
    float den;         // 2 max(qa[best], 0.1)
    Q4 num;            // numerators of the selected candidate
};
TC_HD Rot6 rot6_forward(const float* d6) {
    Rot6 r;
    const V3 a1 = v3(d6[0], d6[1], d6[2]), a2 = v3(d6[3], d6[4], d6[5]);
    r.n1 = fmaxf(sqrtf(dot3(a1, a1)), 1e-12f);
    r.b1 = v3(a1.x / r.n1, a1.y / r.n1, a1.z / r.n1);
    r.dt = dot3(r.b1, a2);
    const V3 u2 = v3(a2.x - r.dt * r.b1.x, a2.y - r.dt * r.b1.y, a2.z - r.dt * r.b1.z);
    r.n2 = fmaxf(sqrtf(dot3(u2, u2)), 1e-12f);
    r.b2 = v3(u2.x / r.n2, u2.y / r.n2, u2.z / r.n2);
    r.b3 = cross3(r.b1, r.b2);
    const float m00 = r.b1.x, m01 = r.b1.y, m02 = r.b1.z, m10 = r.b2.x, m11 = r.b2.y, m12 = r.b2.z;
    const float m20 = r.b3.x, m21 = r.b3.y, m22 = r.b3.z;
    r.qa[0] = sqrt_pos(1.0f + m00 + m11 + m22);
    r.qa[1] = sqrt_pos(1.0f + m00 - m11 - m22);
    r.qa[2] = sqrt_pos(1.0f - m00 + m11 - m22);
    r.qa[3] = sqrt_pos(1.0f - m00 - m11 + m22);
    r.best = 0;                     // argmax, first maximum wins (torch.argmax)
    float bm = r.qa[0];
    if (r.qa[1] > bm) { bm = r.qa[1]; r.best = 1; }
    if (r.qa[2] > bm) { bm = r.qa[2]; r.best = 2; }
    if (r.qa[3] > bm) { bm = r.qa[3]; r.best = 3; }
    r.den = 2.0f * fmaxf(bm, 0.1f);
    if (r.best == 0) r.num = q4(bm * bm, m21 - m12, m02 - m20, m10 - m01);
    else if (r.best == 1) r.num = q4(m21 - m12, bm * bm, m10 + m01, m02 + m20);
    else if (r.best == 2) r.num = q4(m02 - m20, m10 + m01, bm * bm, m12 + m21);
    else r.num = q4(m10 - m01, m20 + m02, m21 + m12, bm * bm);
    return r;
}
TC_HD Q4 quat_from_6d(const float* d6) {
    const Rot6 r = rot6_forward(d6);
    return q4(r.num.w / r.den, r.num.x / r.den, r.num.y / r.den, r.num.z / r.den);
}
// sin(half) / angle and its derivative with respect to the angle (small-angle series below 1e-6)
TC_HD float sinc_half(float ang, float half) { return fabsf(ang) < 1e-6f ? 0.5f - (ang * ang) / 48.0f : sinf(half) / ang; }
TC_HD float sinc_half_grad(float ang, float half) {
    return fabsf(ang) < 1e-6f ? -ang / 24.0f : (0.5f * cosf(half) * ang - sinf(half)) / (ang * ang);
}
TC_HD V3 axis_angle_from_quat(Q4 q) {
    const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z);
    const float half = atan2f(nrm, q.w), ang = 2.0f * half;
    const float k = sinc_half(ang, half);
    return v3(q.x / k, q.y / k, q.z / k);
}
TC_HD Q4 quat_from_axis_angle(V3 a) {
    const float ang = sqrtf(dot3(a, a)), half = ang * 0.5f;
    const float k = sinc_half(ang, half);
    return q4(cosf(half), a.x * k, a.y * k, a.z * k);
}
TC_HD Q4 qmul_raw(Q4 a, Q4 b) {
    return q4(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w);
}
TC_HD Q4 qmul_std(Q4 a, Q4 b) {
    Q4 r = qmul_raw(a, b);
    if (r.w < 0.0f) r = q4(-r.w, -r.x, -r.y, -r.z);
    return r;
}
TC_HD V3 qapply(Q4 q, V3 p) {
    const Q4 t = qmul_raw(qmul_raw(q, q4(0.0f, p.x, p.y, p.z)), qconj(q));
    return v3(t.x, t.y, t.z);
}

struct FkSkel { int parent[TC_FK_J]; int has_children[TC_FK_J]; float off[TC_FK_J][3]; };

// joints[j] = world position of joint j; rw (optional, [24]) receives the world rotations the chain used
TC_HD void fk_forward(const float* aa, const float* root, const FkSkel& sk, float* joints, Q4* rw_out) {
    Q4 rw[TC_FK_J];
    V3 pw[TC_FK_J];
    for (int j = 0; j < TC_FK_J; ++j) {
        const Q4 q = quat_from_axis_angle(v3(aa[3 * j], aa[3 * j + 1], aa[3 * j + 2]));
        const int p = sk.parent[j];
        if (p < 0) {
            pw[j] = v3(root[0], root[1], root[2]);
            rw[j] = q;
        } else {
            pw[j] = add3(qapply(rw[p], v3(sk.off[j][0], sk.off[j][1], sk.off[j][2])), pw[p]);
            rw[j] = sk.has_children[j] ? qmul_std(rw[p], q) : q;
        }
        joints[3 * j + 0] = pw[j].x;
        joints[3 * j + 1] = pw[j].y;
        joints[3 * j + 2] = pw[j].z;
        if (rw_out) rw_out[j] = rw[j];
    }
}

// ---- reverse mode ---------------------------------------------------------------------------------------------------
// r = a (x) b (quaternion product, Euclidean inner product on R^4):  g_a = g (x) conj(b),  g_b = conj(a) (x) g
TC_HD void qmul_raw_bwd(Q4 a, Q4 b, Q4 g, Q4& ga, Q4& gb) {
    ga = qmul_raw(g, qconj(b));
    gb = qmul_raw(qconj(a), g);
}
// out = (q (x) (0, p) (x) conj(q)).xyz;  returns d loss / d q for a constant point p
TC_HD Q4 qapply_bwd_q(Q4 q, V3 p, V3 g) {
    const Q4 P = q4(0.0f, p.x, p.y, p.z), G = q4(0.0f, g.x, g.y, g.z);
    const Q4 t1 = qmul_raw(q, P);
    Q4 g_t1, g_qc, g_q, g_p;
    qmul_raw_bwd(t1, qconj(q), G, g_t1, g_qc);
    qmul_raw_bwd(q, P, g_t1, g_q, g_p);
    (void)g_p;
    return qadd(g_q, qconj(g_qc));      // conj is self-adjoint
}
TC_HD V3 quat_from_axis_angle_bwd(V3 a, Q4 g) {
    const float ang = sqrtf(dot3(a, a)), half = ang * 0.5f;
    const float k = sinc_half(ang, half), dk = sinc_half_grad(ang, half);
    const V3 gv = v3(g.x, g.y, g.z);
    const float g_ang = g.w * (-0.5f * sinf(half)) + dot3(gv, a) * dk;
    const float s = ang > 0.0f ? g_ang / ang : 0.0f;          // d|a| / da = a / |a| (0 at the origin, as torch.norm)
    return v3(k * gv.x + s * a.x, k * gv.y + s * a.y, k * gv.z + s * a.z);
}
TC_HD Q4 axis_angle_from_quat_bwd(Q4 q, V3 g) {
    const V3 v = v3(q.x, q.y, q.z);
    const float n = sqrtf(dot3(v, v));
    const float half = atan2f(n, q.w), ang = 2.0f * half;
    const float k = sinc_half(ang, half), dk = sinc_half_grad(ang, half);
    const float g_k = -dot3(g, v) / (k * k);
    const float g_half = 2.0f * g_k * dk;
    const float d2 = n * n + q.w * q.w;
    const float g_n = d2 > 0.0f ? g_half * q.w / d2 : 0.0f;
    const float g_w = d2 > 0.0f ? -g_half * n / d2 : 0.0f;
    const float s = n > 0.0f ? g_n / n : 0.0f;
    return q4(g_w, g.x / k + s * v.x, g.y / k + s * v.y, g.z / k + s * v.z);
}
// d loss / d (6-D rotation) from d loss / d quaternion (of quat_from_6d)
TC_HD void quat_from_6d_bwd(const float* d6, Q4 g, float* g6) {
    const Rot6 r = rot6_forward(d6);
    const float qb = r.qa[r.best];
    // out = num / den
    const Q4 g_num = q4(g.w / r.den, g.x / r.den, g.y / r.den, g.z / r.den);
    const float g_den = -(g.w * r.num.w + g.x * r.num.x + g.y * r.num.y + g.z * r.num.z) / (r.den * r.den);
    const float gn[4] = {g_num.w, g_num.x, g_num.y, g_num.z};
    float g_qb = (qb > 0.1f ? 2.0f * g_den : 0.0f) + 2.0f * qb * gn[r.best];
    const float g_s = qb > 0.0f ? g_qb / (2.0f * qb) : 0.0f;       // qb = sqrt(s), s > 0
    float gm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    // trace term s = 1 + e0 m00 + e1 m11 + e2 m22
    const float e[4][3] = {{1, 1, 1}, {1, -1, -1}, {-1, 1, -1}, {-1, -1, 1}};
    gm[0][0] += e[r.best][0] * g_s;
    gm[1][1] += e[r.best][1] * g_s;
    gm[2][2] += e[r.best][2] * g_s;
    // linear numerators
    if (r.best == 0) {          // {q0^2, m21 - m12, m02 - m20, m10 - m01}
        gm[2][1] += gn[1]; gm[1][2] -= gn[1]; gm[0][2] += gn[2]; gm[2][0] -= gn[2]; gm[1][0] += gn[3]; gm[0][1] -= gn[3];
    } else if (r.best == 1) {   // {m21 - m12, q1^2, m10 + m01, m02 + m20}
        gm[2][1] += gn[0]; gm[1][2] -= gn[0]; gm[1][0] += gn[2]; gm[0][1] += gn[2]; gm[0][2] += gn[3]; gm[2][0] += gn[3];
    } else if (r.best == 2) {   // {m02 - m20, m10 + m01, q2^2, m12 + m21}
        gm[0][2] += gn[0]; gm[2][0] -= gn[0]; gm[1][0] += gn[1]; gm[0][1] += gn[1]; gm[1][2] += gn[3]; gm[2][1] += gn[3];
    } else {                    // {m10 - m01, m20 + m02, m21 + m12, q3^2}
        gm[1][0] += gn[0]; gm[0][1] -= gn[0]; gm[2][0] += gn[1]; gm[0][2] += gn[1]; gm[2][1] += gn[2]; gm[1][2] += gn[2];
    }
    V3 g_b1 = v3(gm[0][0], gm[0][1], gm[0][2]), g_b2 = v3(gm[1][0], gm[1][1], gm[1][2]);
    const V3 g_b3 = v3(gm[2][0], gm[2][1], gm[2][2]);
    // b3 = b1 x b2
    g_b1 = add3(g_b1, cross3(r.b2, g_b3));
    g_b2 = add3(g_b2, cross3(g_b3, r.b1));
    // b2 = u2 / n2 (F.normalize; no gradient through the 1e-12 clamp)
    const V3 a2 = v3(d6[3], d6[4], d6[5]);
    const V3 g_u2 = scale3(add3(g_b2, scale3(r.b2, -dot3(r.b2, g_b2))), 1.0f / r.n2);
    // u2 = a2 - dt b1,  dt = b1 . a2
    V3 g_a2 = g_u2;
    const float g_dt = -dot3(g_u2, r.b1);
    g_b1 = add3(g_b1, scale3(g_u2, -r.dt));
    g_b1 = add3(g_b1, scale3(a2, g_dt));
    g_a2 = add3(g_a2, scale3(r.b1, g_dt));
    // b1 = a1 / n1
    const V3 g_a1 = scale3(add3(g_b1, scale3(r.b1, -dot3(r.b1, g_b1))), 1.0f / r.n1);
    g6[0] = g_a1.x; g6[1] = g_a1.y; g6[2] = g_a1.z;
    g6[3] = g_a2.x; g6[4] = g_a2.y; g6[5] = g_a2.z;
}
// d loss / d (6-D rotation) from d loss / d axis-angle  (reverse of ax_from_6v)
TC_HD void ax_from_6v_bwd(const float* d6, V3 g_aa, float* g6) {
    const Q4 q = quat_from_6d(d6);
    quat_from_6d_bwd(d6, axis_angle_from_quat_bwd(q, g_aa), g6);
}

// reverse of fk_forward: g_joints [24][3] -> g_aa [24][3], g_root [3]
TC_HD void fk_backward(const float* aa, const FkSkel& sk, const float* g_joints, float* g_aa, float* g_root) {
    Q4 rw[TC_FK_J], ql[TC_FK_J];
    bool flipped[TC_FK_J];
    for (int j = 0; j < TC_FK_J; ++j) {          // recompute the world rotations
        ql[j] = quat_from_axis_angle(v3(aa[3 * j], aa[3 * j + 1], aa[3 * j + 2]));
        const int p = sk.parent[j];
        flipped[j] = false;
        if (p < 0) {
            rw[j] = ql[j];
        } else if (sk.has_children[j]) {
            const Q4 r = qmul_raw(rw[p], ql[j]);
            flipped[j] = r.w < 0.0f;
            rw[j] = flipped[j] ? q4(-r.w, -r.x, -r.y, -r.z) : r;
        } else {
            rw[j] = ql[j];
        }
    }
    V3 gp[TC_FK_J];
    Q4 gr[TC_FK_J];
    for (int j = 0; j < TC_FK_J; ++j) {
        gp[j] = v3(g_joints[3 * j], g_joints[3 * j + 1], g_joints[3 * j + 2]);
        gr[j] = q4(0, 0, 0, 0);
    }
    for (int j = TC_FK_J - 1; j >= 0; --j) {     // children before parents (a parent precedes its children)
        const int p = sk.parent[j];
        V3 ga = v3(0, 0, 0);
        if (p < 0) {
            ga = quat_from_axis_angle_bwd(v3(aa[0 + 3 * j], aa[1 + 3 * j], aa[2 + 3 * j]), gr[j]);
            g_root[0] = gp[j].x; g_root[1] = gp[j].y; g_root[2] = gp[j].z;
        } else {
            // pw[j] = qapply(rw[p], off[j]) + pw[p]
            gp[p] = add3(gp[p], gp[j]);
            gr[p] = qadd(gr[p], qapply_bwd_q(rw[p], v3(sk.off[j][0], sk.off[j][1], sk.off[j][2]), gp[j]));
            if (sk.has_children[j]) {            // rw[j] = standardize(rw[p] (x) q_j); a leaf's rotation is never used
                const Q4 g = flipped[j] ? q4(-gr[j].w, -gr[j].x, -gr[j].y, -gr[j].z) : gr[j];
                Q4 g_rp, g_q;
                qmul_raw_bwd(rw[p], ql[j], g, g_rp, g_q);
                gr[p] = qadd(gr[p], g_rp);
                ga = quat_from_axis_angle_bwd(v3(aa[3 * j], aa[3 * j + 1], aa[3 * j + 2]), g_q);
            }
        }
        g_aa[3 * j] = ga.x; g_aa[3 * j + 1] = ga.y; g_aa[3 * j + 2] = ga.z;
    }
}
