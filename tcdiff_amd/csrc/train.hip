// Training-side rows of the TCDiff hot path (a15, f1, f2), gfx950: forward pieces only -- the noising
// step, the rotation conversions + SMPL forward kinematics of the FK / foot-skate terms, the four loss reductions, and
// the Adan parameter update.  All fp32 and bandwidth- or latency-bound (a few MB per call): plain one-thread-per-item
// kernels with coalesced rows; nothing here is worth MFMA or LDS tiling.
//
//   q_sample + trajectory restore + permute   model/diffusion.py:625-634,640-651
//   ax_from_6v                                 dataset/quaternion.py:28-32   (pytorch3d rotation_6d_to_matrix, matrix_to_axis_angle)
//   SMPLSkeleton.forward                       vis.py:358-406                (pytorch3d axis_angle_to_quaternion, quaternion_apply/_multiply)
//   the four loss terms                        model/diffusion.py:668-741
//   Adan.step                                  model/adan.py:33-123
// The pytorch3d arithmetic is restated from its published definitions (oracle/tcdiff_oracle.py, "parity unpinned").
#include "common.h"
#include "fk_math.h"
#include "tcdiff_hip.h"

// ---------------------------------------------------------------------------------------------------------------------
// x_noisy[b][s*dn + d][c] = a[t_b] * x[b][d][s][c] + s1m[t_b] * noise[b][s][d][c];  channels 4, 5 keep x (trajectory)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void q_sample_traj_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                     const long* __restrict__ t, const float* __restrict__ sa,
                                     const float* __restrict__ s1m, float* __restrict__ out, int b, int dn, int S, int C) {
#pragma clang fp contract(off)   // two rounded products and a rounded sum, like the reference's tensor expression
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)dn * S * C;
    if (i >= (long)b * per) return;
    const int bi = (int)(i / per);
    long rem = i - (long)bi * per;
    const int c = (int)(rem % C);
    rem /= C;
    const int d = (int)(rem % dn), s = (int)(rem / dn);            // output order: frame-major tokens (s, d)
    const float xv = x[(((long)bi * dn + d) * S + s) * C + c];     // dataset layout (b, dn, S, C)
    const float a = sa[t[bi]], bb = s1m[t[bi]];
    const float v = a * xv + bb * noise[i];                        // two roundings and a sum, as the reference's expression
    out[i] = (c == 4 || c == 5) ? xv : v;
}

extern "C" int tcdiff_q_sample_traj(const float* x_start, const float* noise, const long* t, const float* sqrt_ac,
                                    const float* sqrt_1mac, float* x_noisy, int b, int dn, int S, int C,
                                    hipStream_t stream) {
    if (!x_start || !noise || !t || !sqrt_ac || !sqrt_1mac || !x_noisy || b <= 0 || dn <= 0 || S <= 0 || C <= 5)
        return TC_ERR_ARG;
    const long n = (long)b * dn * S * C;
    hipLaunchKernelGGL(q_sample_traj_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x_start, noise, t,
                       sqrt_ac, sqrt_1mac, x_noisy, b, dn, S, C);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// rotation conversions and the SMPL chain: the per-pose arithmetic is csrc/fk_math.h (shared with the backward kernels of
// train_ops.hip and, compiled for the host, with tests/host/fk_host.cpp)
// ---------------------------------------------------------------------------------------------------------------------
// one thread per rotation: rotation j of row i is the 6 floats at in + i * row_stride + 6 j; out [n_rows * per_row][3]
__global__ void ax_from_6v_kernel(const float* __restrict__ in, long n_rows, int per_row, long row_stride,
                                  float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * per_row) return;
    const V3 a = axis_angle_from_quat(quat_from_6d(in + (i / per_row) * row_stride + (i % per_row) * 6));
    out[i * 3 + 0] = a.x;
    out[i * 3 + 1] = a.y;
    out[i * 3 + 2] = a.z;
}

extern "C" int tcdiff_ax_from_6v(const float* rot6d, long n_rows, int per_row, long row_stride, float* axis_angle,
                                 hipStream_t stream) {
    if (!rot6d || !axis_angle || n_rows <= 0 || per_row <= 0 || row_stride < 6L * per_row) return TC_ERR_ARG;
    const long n = n_rows * per_row;
    hipLaunchKernelGGL(ax_from_6v_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rot6d, n_rows, per_row,
                       row_stride, axis_angle);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// SMPL chain: one thread per pose.  parents / offsets are the SMPL constants of vis.py:48-101 (passed in so that a
// caller-supplied skeleton works too); joints whose parents precede them (true for SMPL) only.
__global__ void smpl_fk_kernel(const float* __restrict__ aa, const float* __restrict__ root, long n, FkSkel sk,
                               float* __restrict__ joints) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float out[TC_FK_J * 3];
    fk_forward(aa + i * (TC_FK_J * 3), root + i * 3, sk, out, nullptr);
#pragma unroll
    for (int q = 0; q < TC_FK_J * 3; ++q) joints[i * (TC_FK_J * 3) + q] = out[q];
}

extern "C" int tcdiff_smpl_fk(const float* axis_angle, const float* root, long n, const int* parents,
                              const float* offsets, float* joints, hipStream_t stream) {
    if (!axis_angle || !root || !parents || !offsets || !joints || n <= 0) return TC_ERR_ARG;
    FkSkel sk;
    for (int j = 0; j < TC_FK_J; ++j) sk.has_children[j] = 0;
    for (int j = 0; j < TC_FK_J; ++j) {
        sk.parent[j] = parents[j];
        if (parents[j] >= j) return TC_ERR_ARG;          // a parent must precede its children
        if (parents[j] >= 0) sk.has_children[parents[j]] = 1;
        for (int k = 0; k < 3; ++k) sk.off[j][k] = offsets[3 * j + k];
    }
    hipLaunchKernelGGL(smpl_fk_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, stream, axis_angle, root, n, sk,
                       joints);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// loss terms: out[b][k], k = recon, velocity, fk, foot: per-clip means (times p2 weight for the first three)
// ---------------------------------------------------------------------------------------------------------------------
DEVINL float lossf(float a, float b, int l1) {
    const float d = a - b;
    return l1 ? fabsf(d) : d * d;
}

__global__ __launch_bounds__(1024) void loss_terms_kernel(const float* __restrict__ mo, const float* __restrict__ xs,
                                                         const float* __restrict__ jm, const float* __restrict__ jt,
                                                         const float* __restrict__ w, const long* __restrict__ t,
                                                         float* __restrict__ out, int dn, int S, int C, int l1) {
    const int bi = blockIdx.x, term = blockIdx.y, tid = threadIdx.x;
    const long Lq = (long)S * dn;
    auto model = [&](int s, int d, int c) { return mo[((long)bi * Lq + (long)s * dn + d) * C + c]; };
    auto target = [&](int s, int d, int c) { return xs[(((long)bi * dn + d) * S + s) * C + c]; };   // dataset layout
    // 1024 threads per (clip, term) and 32-bit index arithmetic: with 256 threads walking a clip's 68 k elements through 64-bit
    // divisions the launch took 139 us for 2 M elements
    double acc = 0.0;          // per-thread partial in fp64: the reduction order must not matter at 1e-6
    int count = 1;
    if (term == 0) {
        count = (int)Lq * C;
        for (int i = tid; i < count; i += 1024) {
            const int c = i % C;
            const int sd = i / C;
            const int d = (int)(sd % dn), s = (int)(sd / dn);
            acc += lossf(model(s, d, c), target(s, d, c), l1);
        }
    } else if (term == 1) {
        const int C4 = C - 4;
        count = (S - 1) * dn * C4;
        for (int i = tid; i < count; i += 1024) {
            const int c = 4 + i % C4;
            const int sd = i / C4;
            const int d = (int)(sd % dn), s = (int)(sd / dn);
            acc += lossf(model(s + 1, d, c) - model(s, d, c), target(s + 1, d, c) - target(s, d, c), l1);
        }
    } else if (term == 2) {
        count = (int)Lq * 23 * 3;
        for (int i = tid; i < count; i += 1024) {
            const int k = i % 3;
            const int r2 = i / 3;
            const int j = 1 + (int)(r2 % 23);
            const long row = (long)bi * Lq + r2 / 23;
            acc += lossf(jm[(row * 24 + j) * 3 + k] - jm[(row * 24) * 3 + k], jt[(row * 24 + j) * 3 + k] - jt[(row * 24) * 3 + k], l1);
        }
    } else {
        const int foot[4] = {7, 8, 10, 11};
        count = (int)Lq * 4 * 3;
        for (int i = tid; i < count; i += 1024) {
            const int k = i % 3;
            const int r2 = i / 3;
            const int f = r2 % 4;
            const int sd = r2 / 4;
            const int d = (int)(sd % dn), s = (int)(sd / dn);
            float v = 0.0f;
            if (s + 1 < S && model(s, d, f) > 0.95f) {
                const long r0 = (long)bi * Lq + (long)s * dn + d, r1 = r0 + dn;
                v = jm[(r1 * 24 + foot[f]) * 3 + k] - jm[(r0 * 24 + foot[f]) * 3 + k];
            }
            acc += lossf(v, 0.0f, l1);
        }
    }
    __shared__ double red[1024];
    red[tid] = acc;
    __syncthreads();
    for (int s2 = 512; s2 > 0; s2 >>= 1) {
        if (tid < s2) red[tid] += red[tid + s2];
        __syncthreads();
    }
    if (tid == 0) {
        const float mean = (float)(red[0] / (double)count);
        out[bi * 4 + term] = term < 3 ? mean * w[t[bi]] : mean;
    }
}

extern "C" int tcdiff_loss_terms(const float* model_out, const float* x_start, const float* joints_model,
                                 const float* joints_target, const float* p2_weight, const long* t, float* out, int b,
                                 int dn, int S, int C, int l1, hipStream_t stream) {
    if (!model_out || !x_start || !joints_model || !joints_target || !p2_weight || !t || !out || b <= 0 || dn <= 0 ||
        S < 2 || C <= 7)
        return TC_ERR_ARG;
    if ((long)S * dn * C >= (1L << 30)) return TC_ERR_ARG;          // 32-bit element indices inside a clip
    hipLaunchKernelGGL(loss_terms_kernel, dim3(b, 4), dim3(1024), 0, stream, model_out, x_start, joints_model,
                       joints_target, p2_weight, t, out, dn, S, C, l1);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Adan (model/adan.py:33-123), all parameter tensors in one launch; the reference's rounding points (oracle adan_step):
//   m = fma(g, b1, m * (1 - b1)); gd = g - pg; v = fma(gd, b2, v * (1 - b2)); nn = g + (1 - b2) * gd; nn = nn * nn;
//   n = fma(nn, b3, n * (1 - b3))                                             [skipped on the first step, adan.py:71]
//   wss = (1 / (sqrt(n * cn) + eps)) * lr;  upd = m * cm + ((1 - b2) * v) * cv;  p = fma(-wss, upd, p) / (1 + wd * lr)
// ---------------------------------------------------------------------------------------------------------------------
// one element of the update; every rounding point is the reference's (see tcdiff_hip.h)
DEVINL void adan_element(const tcdiff_adan_scalars& k, float g, float pg, float& p, float& m, float& v, float& n) {
#pragma clang fp contract(off)
    if (k.first & 2) {            // restart (adan.py:109-114): m = g, v = 0, n = g^2, then the parameter update once more
        m = g;
        v = 0.0f;
        n = g * g;
    } else if (!(k.first & 1)) {
        m = __builtin_fmaf(g, k.b1, m * k.omb1);
        const float gd = g - pg;
        v = __builtin_fmaf(gd, k.b2, v * k.omb2);
        float nn = g + k.omb2 * gd;
        nn = nn * nn;
        n = __builtin_fmaf(nn, k.b3, n * k.omb3);
    }
    const float wss = __frcp_rn(sqrtf(n * k.cn) + k.eps) * k.lr;
    const float upd = m * k.cm + (k.omb2 * v) * k.cv;
    p = __fdiv_rn(__builtin_fmaf(-1.0f * wss, upd, p), k.denom);
}

__global__ __launch_bounds__(256) void adan_step_kernel(const tcdiff_adan_chunk* __restrict__ chunks, tcdiff_adan_scalars k) {
    const tcdiff_adan_chunk c = chunks[blockIdx.x];
    // HBM-bound: 6 streams in, 5 out.  16-byte accesses when all six chunk pointers allow it (torch allocations are
    // 512-B aligned and chunk offsets are multiples of 4 elements, so only odd-sized tails take the scalar loop).
    const bool vec = ((reinterpret_cast<uintptr_t>(c.p) | reinterpret_cast<uintptr_t>(c.g) | reinterpret_cast<uintptr_t>(c.m) |
                       reinterpret_cast<uintptr_t>(c.v) | reinterpret_cast<uintptr_t>(c.n_) |
                       reinterpret_cast<uintptr_t>(c.pg)) & 15) == 0;
    long i0 = 0;
    if (vec) {
        const long n4 = c.n >> 2;
        for (long i = threadIdx.x; i < n4; i += 256) {
            const f32x4_t g = reinterpret_cast<const f32x4_t*>(c.g)[i];
            f32x4_t p = reinterpret_cast<f32x4_t*>(c.p)[i], m = reinterpret_cast<f32x4_t*>(c.m)[i];
            f32x4_t v = reinterpret_cast<f32x4_t*>(c.v)[i], n = reinterpret_cast<f32x4_t*>(c.n_)[i];
            f32x4_t pg = {0.f, 0.f, 0.f, 0.f};
            if (!(k.first & 3)) pg = reinterpret_cast<f32x4_t*>(c.pg)[i];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float pe = p[t], me = m[t], ve = v[t], ne = n[t];
                adan_element(k, g[t], pg[t], pe, me, ve, ne);
                p[t] = pe; m[t] = me; v[t] = ve; n[t] = ne;
            }
            if (!(k.first & 1)) {
                reinterpret_cast<f32x4_t*>(c.m)[i] = m;
                reinterpret_cast<f32x4_t*>(c.v)[i] = v;
                reinterpret_cast<f32x4_t*>(c.n_)[i] = n;
            }
            reinterpret_cast<f32x4_t*>(c.p)[i] = p;
            if (!(k.first & 4)) reinterpret_cast<f32x4_t*>(c.pg)[i] = g;
        }
        i0 = n4 << 2;
    }
    for (long i = i0 + threadIdx.x; i < c.n; i += 256) {
        const float g = c.g[i];
        float p = c.p[i], m = c.m[i], v = c.v[i], n = c.n_[i];
        const float pg = (k.first & 3) ? 0.0f : c.pg[i];
        adan_element(k, g, pg, p, m, v, n);
        if (!(k.first & 1)) {
            c.m[i] = m;
            c.v[i] = v;
            c.n_[i] = n;
        }
        c.p[i] = p;
        if (!(k.first & 4)) c.pg[i] = g;
    }
}

extern "C" int tcdiff_adan_step(const tcdiff_adan_chunk* chunks, int n_chunks, const tcdiff_adan_scalars* scalars,
                                hipStream_t stream) {
    if (!chunks || !scalars || n_chunks <= 0) return TC_ERR_ARG;
    hipLaunchKernelGGL(adan_step_kernel, dim3(n_chunks), dim3(256), 0, stream, chunks, *scalars);
    TC_CHECK_LAUNCH();
    return TC_OK;
}
