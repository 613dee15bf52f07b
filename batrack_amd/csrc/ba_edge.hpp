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

template <typename T> struct Vec2;
template <> struct Vec2<float> { typedef float2 type; };
template <> struct Vec2<double> { typedef double2 type; };

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

// Relative pose of camera pair (i, j) and the intrinsics the edge maths needs, 20 numbers:
// R_ij (9, row-major), t_ij (3), (1/fx_i, 1/fy_i, cx_i, cy_i), (fx_j, fy_j, cx_j, cy_j).
// Gij = Gj * Gi^-1 (projective_ops.py:61) in double; a self edge is exactly the identity.
// T = float: rounded to float32 (the float32 per-edge path, k_stream / k_edge2); T = double: kept (the float64 per-edge path).
// RAWK: g[12], g[13] hold fx_i, fy_i themselves instead of their reciprocals (edge_eval_mixed divides in double: a float32
// reciprocal of the one focal length every camera shares is the SAME 6e-8 off for every edge of the graph — a bias along the
// weakly constrained directions of the reduced system, which it took to find: dX 3e-5 with it, 2e-6 without).
template <typename T, bool RAWK = false>
__device__ inline void pair_geometry(const float *poses, const float *intr, int i, int j, T *g) {
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
    for (int c = 0; c < 9; ++c) g[c] = (T)R[c];
    for (int c = 0; c < 3; ++c) g[9 + c] = (T)t[c];
    // source intrinsics as (1/fx, 1/fy, cx, cy): iproj divides (projective_ops.py:25-26)
    if (RAWK) { g[12] = (T)intr[4*i]; g[13] = (T)intr[4*i + 1]; }
    else { g[12] = (T)1 / (T)intr[4*i]; g[13] = (T)1 / (T)intr[4*i + 1]; }
    g[14] = intr[4*i + 2]; g[15] = intr[4*i + 3];
    for (int c = 0; c < 4; ++c) g[16 + c] = intr[4*j + c];
}

// ------------------------------------------------------------------ per-edge math
// T = float: the reference's own precision.  T = double: every operation of the edge maths in float64 on the float32
// inputs — what brings the pose / depth UPDATE within 1e-5 of the reference's float64 result on ill-conditioned
// windows (8 frames, one fixed pose: the float32 Jacobians' rounding is amplified by cond(S) ~ 1e3..1e4).
template <typename T>
struct EdgeQT {
    T a0, a2, a3, a4, a5;      // Jj row 0 = (a0, 0, a2, a3, a4, a5)
    T b1, b2, b3, b4, b5;      // Jj row 1 = (0, b1, b2, b3, b4, b5)
    T jz0, jz1, r0, r1, W0, W1;
};
typedef EdgeQT<float> EdgeQ;

// Reciprocal and reciprocal square root.  float: the hardware approximations (v_rcp_f32 / v_rsq_f32, 1 ulp) instead of
// the IEEE-exact sequences (~10 instructions each, five of them per edge): an ulp here is the same size as the
// float32 rounding of every other operation of the edge maths.  double: the float32 seed and two Newton steps
// (relative error < 1e-15; the compiler's IEEE divide is ~25 instructions on the quarter-rate pipe).
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double frcp(double x) {
    double y = (double)__builtin_amdgcn_rcpf((float)x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double frsq(double x) { return rsqrt_nr2(x); }

template <typename T>
__device__ __forceinline__ T robust_weight(T r, int loss) {          // ba.py:81-100
    const T s = r * r;
    if (loss == BT_LOSS_HUBER) return s > (T)1 ? frsq(s) : (T)1;
    if (loss == BT_LOSS_CAUCHY) return frcp((T)1 + s);
    return (T)1;
}

__device__ __forceinline__ float fmax_t(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double fmax_t(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float fabs_t(float a) { return fabsf(a); }
__device__ __forceinline__ double fabs_t(double a) { return fabs(a); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return fma(a, b, c); }

template <typename T>
__device__ __forceinline__ void edge_eval(const T *g, T x, T y, T d, T tu, T tv,
                                          T w0, T w1, const StepArgs &a, EdgeQT<T> &o) {
    // projective_ops.py:19-29 (iproj), :61-66 (act4), :43-45 (proj)
    const T X0 = (x - g[14]) * g[12], Y0 = (y - g[15]) * g[13];
    const T X = fma_t(g[0], X0, fma_t(g[1], Y0, g[2])) + g[9] * d;
    const T Y = fma_t(g[3], X0, fma_t(g[4], Y0, g[5])) + g[10] * d;
    const T Z = fma_t(g[6], X0, fma_t(g[7], Y0, g[8])) + g[11] * d;
    const T fx = g[16], fy = g[17];
    const T iz = frcp(fmax_t(Z, (T)1e-2));
    const T u = fma_t(fx, iz * X, g[18]), v = fma_t(fy, iz * Y, g[19]);
    // projective_ops.py:80-98
    const T dj = fabs_t(Z) > (T)0.2 ? frcp(Z) : (T)0;
    const T A = fx * dj, B = -fx * X * dj * dj, C = fy * dj, Dd = -fy * Y * dj * dj;
    o.a0 = d * A;  o.a2 = d * B;  o.a3 = B * Y;            o.a4 = A * Z - B * X;  o.a5 = -A * Y;
    o.b1 = d * C;  o.b2 = d * Dd; o.b3 = Dd * Y - C * Z;   o.b4 = -Dd * X;        o.b5 = C * X;
    o.jz0 = fma_t(A, g[9], B * g[11]);
    o.jz1 = fma_t(C, g[10], Dd * g[11]);
    // ba.py:230-251
    const T r0 = tu - u, r1 = tv - v;
    T vld = Z > (T)0.2 ? (T)1 : (T)0;
    vld *= r0 * r0 + r1 * r1 < (T)62500 ? (T)1 : (T)0;        // |r| < 250 (ba.py:233), compared squared
    vld *= (u > (T)a.b0 && v > (T)a.b1 && u < (T)a.b2 && v < (T)a.b3) ? (T)1 : (T)0;
    o.W0 = vld * (w0 * robust_weight(r0, a.loss));
    o.W1 = vld * (w1 * robust_weight(r1, a.loss));
    o.r0 = vld * r0; o.r1 = vld * r1;
}

// MIXED precision (the wave-per-tile kernels of large graphs, round 4): the reprojection and the residual in float64 on the
// float32 inputs and on the pair's float32 geometry — u = fx X / Z + cx is ~500 px and r = target - u ~0.5 px, so a float32
// u (3e-5 px) costs r five digits, and that, not the Jacobians' rounding, is what put the float32 kernels' update at
// 1e-5 .. 5e-5 from the reference's float64 run on the benchmark graphs (measured stage by stage on the host: projection in
// float32 alone dX 3.3e-5; geometry, Jacobians and their products in float32 with the projection in float64: 2e-6) —,
// validity and bounds decided on those float64 values, then Jacobians, robust weights and every product in float32.
__device__ __forceinline__ void edge_eval_mixed(const float *g, float x, float y, float d, float tu, float tv,
                                                float w0, float w1, const StepArgs &a, EdgeQT<float> &o) {
    // projective_ops.py:19-29 (iproj), :61-66 (act4), :43-45 (proj) in double
    // (g[12], g[13] = fx_i, fy_i: pair_geometry<float, true>) the quotient to double precision from the float32 hardware
    // reciprocal: q = fl32(n * rcp), X0 = q + (n - q fx) rcp
    const double nx = (double)x - (double)g[14], ny = (double)y - (double)g[15];
    const float rfx = frcp(g[12]), rfy = frcp(g[13]);
    const float qx = (float)nx * rfx, qy = (float)ny * rfy;
    const double X0 = fma(fma(-(double)qx, (double)g[12], nx), (double)rfx, (double)qx);
    const double Y0 = fma(fma(-(double)qy, (double)g[13], ny), (double)rfy, (double)qy);
    const double dd = (double)d;
    const double Xd = fma((double)g[0], X0, fma((double)g[1], Y0, (double)g[2])) + (double)g[9] * dd;
    const double Yd = fma((double)g[3], X0, fma((double)g[4], Y0, (double)g[5])) + (double)g[10] * dd;
    const double Zd = fma((double)g[6], X0, fma((double)g[7], Y0, (double)g[8])) + (double)g[11] * dd;
    const double izd = frcp(fmax(Zd, 1e-2));
    const double ud = fma((double)g[16], izd * Xd, (double)g[18]), vd = fma((double)g[17], izd * Yd, (double)g[19]);
    const double r0d = (double)tu - ud, r1d = (double)tv - vd;
    float vld = Zd > 0.2 ? 1.0f : 0.0f;
    vld *= r0d * r0d + r1d * r1d < 62500.0 ? 1.0f : 0.0f;               // |r| < 250 (ba.py:233), compared squared
    vld *= (ud > (double)a.b0 && vd > (double)a.b1 && ud < (double)a.b2 && vd < (double)a.b3) ? 1.0f : 0.0f;
    // projective_ops.py:80-98 in float on the rounded point
    const float X = (float)Xd, Y = (float)Yd, Z = (float)Zd, fx = g[16], fy = g[17];
    const float dj = fabsf(Z) > 0.2f ? frcp(Z) : 0.0f;
    const float A = fx * dj, B = -fx * X * dj * dj, C = fy * dj, Dd = -fy * Y * dj * dj;
    o.a0 = d * A;  o.a2 = d * B;  o.a3 = B * Y;            o.a4 = A * Z - B * X;  o.a5 = -A * Y;
    o.b1 = d * C;  o.b2 = d * Dd; o.b3 = Dd * Y - C * Z;   o.b4 = -Dd * X;        o.b5 = C * X;
    o.jz0 = fmaf(A, g[9], B * g[11]);
    o.jz1 = fmaf(C, g[10], Dd * g[11]);
    const float r0 = (float)r0d, r1 = (float)r1d;
    o.W0 = vld * (w0 * robust_weight(r0, a.loss));
    o.W1 = vld * (w1 * robust_weight(r1, a.loss));
    o.r0 = vld * r0; o.r1 = vld * r1;
}

// what the wave-per-tile kernels (k_stream, k_edge2, k_edge2u) evaluate an edge with; -DBT_WPT_MIXED=0: plain float32 (measurement builds)
#ifndef BT_WPT_MIXED
#define BT_WPT_MIXED 1
#endif
#if BT_WPT_MIXED
#define BT_WPT_EDGE_EVAL edge_eval_mixed
#else
#define BT_WPT_EDGE_EVAL edge_eval<float>
#endif

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


// The same for doubles (the float64 per-edge path): the lane swaps and DPP moves act on the two 32-bit halves.
__device__ __forceinline__ void wave_reduce_scatter32(double (&v)[32], int lane) {
    auto lo = [](double x) { return (unsigned)(__double_as_longlong(x) & 0xffffffffll); };
    auto hi = [](double x) { return (unsigned)((unsigned long long)__double_as_longlong(x) >> 32); };
    auto mk = [](unsigned l, unsigned h) { return __longlong_as_double((long long)(((unsigned long long)h << 32) | l)); };
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint2_t rl = __builtin_amdgcn_permlane32_swap(lo(v[i]), lo(v[i + 16]), false, false);
        const uint2_t rh = __builtin_amdgcn_permlane32_swap(hi(v[i]), hi(v[i + 16]), false, false);
        v[i] = mk(rl.x, rh.x) + mk(rl.y, rh.y);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint2_t rl = __builtin_amdgcn_permlane16_swap(lo(v[i]), lo(v[i + 8]), false, false);
        const uint2_t rh = __builtin_amdgcn_permlane16_swap(hi(v[i]), hi(v[i + 8]), false, false);
        v[i] = mk(rl.x, rh.x) + mk(rl.y, rh.y);
    }
#define BT_DPPD(x, ctrl) mk((unsigned)__builtin_amdgcn_update_dpp(0, (int)lo(x), (ctrl), 0xf, 0xf, true), \
                            (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi(x), (ctrl), 0xf, 0xf, true))
#define BT_RS_STEPD(M, H, CTRL)                                     \
    {                                                               \
        const bool up = (lane & (M)) != 0;                          \
        _Pragma("unroll") for (int i = 0; i < (H); ++i) {           \
            const double send = up ? v[i] : v[i + (H)];             \
            const double keep = up ? v[i + (H)] : v[i];             \
            v[i] = keep + BT_DPPD(send, CTRL);                      \
        }                                                           \
    }
    BT_RS_STEPD(8, 4, 0x128)
    BT_RS_STEPD(4, 2, 0x141)
    BT_RS_STEPD(2, 1, 0x4e)
#undef BT_RS_STEPD
    v[0] += BT_DPPD(v[0], 0xb1);
#undef BT_DPPD
}

}  // namespace bt
