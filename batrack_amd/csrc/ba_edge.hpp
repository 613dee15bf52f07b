// ba_edge.hpp — device helpers shared by the tile kernels (ba_kernels.hip) and the wave-per-tile
// streaming kernels (ba_stream.hip): relative pose of a camera pair, the per-edge reprojection /
// Jacobian / robust-weight arithmetic (projective_ops.py:54-100, ba.py:228-266), and the wave-wide
// reduce-scatter of the 27 per-pair products.
#pragma once
#include <hip/hip_runtime.h>

#include "ba_kernels.hpp"

namespace bt {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ pair geometry
// 1/sqrt(x) in double from the fp32 hardware seed and two Newton steps (relative error < 1e-15; the IEEE
// sqrt + divide it replaces is ~60 instructions on the prologue's critical path)
__device__ __forceinline__ double rsqrt_nr2(double x) {
    double y = (double)__builtin_amdgcn_rsqf((float)x);
    y = y * (1.5 - 0.5 * x * y * y);
    return y * (1.5 - 0.5 * x * y * y);
}

__device__ inline void quat_to_rot(const double *q, double R[9]) {
    const double n = rsqrt_nr2(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
    const double x = q[0]*n, y = q[1]*n, z = q[2]*n, w = q[3]*n;
    R[0] = 1 - 2*(y*y + z*z); R[1] = 2*(x*y - z*w);     R[2] = 2*(x*z + y*w);
    R[3] = 2*(x*y + z*w);     R[4] = 1 - 2*(x*x + z*z); R[5] = 2*(y*z - x*w);
    R[6] = 2*(x*z - y*w);     R[7] = 2*(y*z + x*w);     R[8] = 1 - 2*(x*x + y*y);
}

// Relative pose of camera pair (i, j) and the intrinsics the edge maths needs, 20 floats:
// R_ij (9, row-major), t_ij (3), (1/fx_i, 1/fy_i, cx_i, cy_i), (fx_j, fy_j, cx_j, cy_j).
// Gij = Gj * Gi^-1 (projective_ops.py:61) in double; a self edge is exactly the identity.
__device__ inline void pair_geometry(const float *poses, const float *intr, int i, int j, float *g) {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
    if (i != j) {
        double qi[4], qj[4], Ri[9], Rj[9], ti[3], tj[3];
        for (int c = 0; c < 3; ++c) { ti[c] = poses[7*i + c]; tj[c] = poses[7*j + c]; }
        for (int c = 0; c < 4; ++c) { qi[c] = poses[7*i + 3 + c]; qj[c] = poses[7*j + 3 + c]; }
        quat_to_rot(qi, Ri); quat_to_rot(qj, Rj);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                R[3*r + c] = Rj[3*r]*Ri[3*c] + Rj[3*r + 1]*Ri[3*c + 1] + Rj[3*r + 2]*Ri[3*c + 2];
        for (int r = 0; r < 3; ++r)
            t[r] = tj[r] - (R[3*r]*ti[0] + R[3*r + 1]*ti[1] + R[3*r + 2]*ti[2]);
    }
    for (int c = 0; c < 9; ++c) g[c] = (float)R[c];
    for (int c = 0; c < 3; ++c) g[9 + c] = (float)t[c];
    // source intrinsics as (1/fx, 1/fy, cx, cy): iproj divides (projective_ops.py:25-26)
    g[12] = 1.0f / intr[4*i]; g[13] = 1.0f / intr[4*i + 1]; g[14] = intr[4*i + 2]; g[15] = intr[4*i + 3];
    for (int c = 0; c < 4; ++c) g[16 + c] = intr[4*j + c];
}

// ------------------------------------------------------------------ per-edge math
struct EdgeQ {
    float a0, a2, a3, a4, a5;      // Jj row 0 = (a0, 0, a2, a3, a4, a5)
    float b1, b2, b3, b4, b5;      // Jj row 1 = (0, b1, b2, b3, b4, b5)
    float jz0, jz1, r0, r1, W0, W1;
};

// Reciprocal and reciprocal square root on the hardware approximations (v_rcp_f32 / v_rsq_f32, 1 ulp) instead of
// the IEEE-exact sequences (~10 instructions each, five of them per edge): an ulp here is the same size as the
// float32 rounding of every other operation of the edge maths.
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float robust_weight(float r, int loss) {          // ba.py:81-100
    const float s = r * r;
    if (loss == BT_LOSS_HUBER) return s > 1.0f ? __builtin_amdgcn_rsqf(s) : 1.0f;
    if (loss == BT_LOSS_CAUCHY) return frcp(1.0f + s);
    return 1.0f;
}

__device__ __forceinline__ void edge_eval(const float *g, float x, float y, float d, float tu, float tv,
                                          float w0, float w1, const StepArgs &a, EdgeQ &o) {
    // projective_ops.py:19-29 (iproj), :61-66 (act4), :43-45 (proj)
    const float X0 = (x - g[14]) * g[12], Y0 = (y - g[15]) * g[13];
    const float X = fmaf(g[0], X0, fmaf(g[1], Y0, g[2])) + g[9] * d;
    const float Y = fmaf(g[3], X0, fmaf(g[4], Y0, g[5])) + g[10] * d;
    const float Z = fmaf(g[6], X0, fmaf(g[7], Y0, g[8])) + g[11] * d;
    const float fx = g[16], fy = g[17];
    const float iz = frcp(fmaxf(Z, 1e-2f));
    const float u = fmaf(fx, iz * X, g[18]), v = fmaf(fy, iz * Y, g[19]);
    // projective_ops.py:80-98
    const float dj = fabsf(Z) > 0.2f ? frcp(Z) : 0.0f;
    const float A = fx * dj, B = -fx * X * dj * dj, C = fy * dj, Dd = -fy * Y * dj * dj;
    o.a0 = d * A;  o.a2 = d * B;  o.a3 = B * Y;            o.a4 = A * Z - B * X;  o.a5 = -A * Y;
    o.b1 = d * C;  o.b2 = d * Dd; o.b3 = Dd * Y - C * Z;   o.b4 = -Dd * X;        o.b5 = C * X;
    o.jz0 = fmaf(A, g[9], B * g[11]);
    o.jz1 = fmaf(C, g[10], Dd * g[11]);
    // ba.py:230-251
    const float r0 = tu - u, r1 = tv - v;
    float vld = Z > 0.2f ? 1.0f : 0.0f;
    vld *= r0 * r0 + r1 * r1 < 62500.0f ? 1.0f : 0.0f;        // |r| < 250 (ba.py:233), compared squared
    vld *= (u > a.b0 && v > a.b1 && u < a.b2 && v < a.b3) ? 1.0f : 0.0f;
    o.W0 = vld * (w0 * robust_weight(r0, a.loss));
    o.W1 = vld * (w1 * robust_weight(r1, a.loss));
    o.r0 = vld * r0; o.r1 = vld * r1;
}

// Sum v[0..31] over the 64 lanes of a wave; afterwards every lane holds, in v[0], the
// total of element ((lane >> 1) & 31).  Halving exchange: 16 + 8 lane-swap instructions
// (v_permlane32_swap / v_permlane16_swap move two registers at once) and 7 shuffles,
// instead of 32 * 6 shuffles for a plain butterfly.
typedef unsigned uint2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wave_reduce_scatter32(float (&v)[32], int lane) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {       // lanes < 32 keep v[i], lanes >= 32 keep v[i+16]
        const uint2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 16]), false, false);
        v[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {        // even 16-lane rows keep v[i], odd rows keep v[i+8]
        const uint2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 8]), false, false);
        v[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
    // the last four halvings stay inside a row of 16 lanes: DPP operands of the adds (v_add_f32_dpp: full VALU rate, no
    // LDS round trip as a ds_bpermute shuffle would cost).  The partner of a step only has to differ in the step's lane
    // bit and agree in the higher ones: row_ror:8 (lane ^ 8), row_half_mirror (lane ^ 7), quad_perm (lane ^ 2, lane ^ 1).
#define BT_DPP(x, ctrl) __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), (ctrl), 0xf, 0xf, true))
#define BT_RS_STEP(M, H, CTRL)                                      \
    {                                                               \
        const bool up = (lane & (M)) != 0;                          \
        _Pragma("unroll") for (int i = 0; i < (H); ++i) {           \
            const float send = up ? v[i] : v[i + (H)];              \
            const float keep = up ? v[i + (H)] : v[i];              \
            v[i] = keep + BT_DPP(send, CTRL);                       \
        }                                                           \
    }
    BT_RS_STEP(8, 4, 0x128)
    BT_RS_STEP(4, 2, 0x141)
    BT_RS_STEP(2, 1, 0x4e)
#undef BT_RS_STEP
    v[0] += BT_DPP(v[0], 0xb1);
#undef BT_DPP
}

}  // namespace bt
