// ba_edge2.hpp — what the two-edges-per-lane kernels of the edge-major layout share (ba_edge2.hip: k_edge2, the pose+structure
// reduce; ba_edge2u.hip: k_edge2u, the structure-only step and the depth back-substitution): the tile record, packed-float32
// helpers, lane-group sums by DPP, the float64 reprojection of one edge.
#pragma once
#include <hip/hip_runtime.h>

#include "ba_edge.hpp"
#include "ba_kernels.hpp"

namespace bt {
namespace e2 {

typedef float f2 __attribute__((ext_vector_type(2)));
#ifndef BT_E2_PRIO_SHIFT
#define BT_E2_PRIO_SHIFT 13        // log2 of the priority slice in shader clocks
#endif
#ifndef BT_E2_PRIO_YOUNG
#define BT_E2_PRIO_YOUNG 4         // eighths of a priority period in which the SIMD's SECOND wave is the raised one
#endif
#ifndef BT_E2_SB
#define BT_E2_SB __builtin_amdgcn_sched_barrier(0)
#endif

constexpr int kGeoD = 16;              // doubles per pair: R (9), t (3), fx_j, fy_j, cx_j, cy_j
constexpr int kGeoF = 16;              // floats per pair: t (3), fx_j | R (9), fy_j, -, -
// ... and their strides in LDS: the lanes of a wave read the rows of S different pairs with b128 reads; with rows of 128 / 64
// bytes they meet in two / four banks groups (measured: a third of the kernel's LDS cycles were bank conflicts)
constexpr int kGeoDS = 18, kGeoFS = 20;
struct Rec { int ntrk, ncam, npair, flags, cam0, pair0, trk0, it0, lgS, nit; };
// A tile's record as loaded (vector registers, every lane the same): requested two tiles ahead and decoded into scalar
// registers only when its tile comes up — decoding where it is loaded is a wait for the load, once per tile.
struct RawRec { int4 r0, r1; };
__device__ __forceinline__ RawRec load_raw(const PlanDev &pd, int t) {
    RawRec w;
    w.r0 = reinterpret_cast<const int4 *>(pd.tile_rec)[2 * t]; w.r1 = reinterpret_cast<const int4 *>(pd.tile_rec)[2 * t + 1];
    return w;
}

__device__ __forceinline__ Rec decode_rec(const RawRec &w) {
    const int4 r0 = w.r0, r1 = w.r1;
    Rec r;
    // (the tile index is wave-uniform; the loads are vector loads all the same — the kernel stores to global memory, so the
    //  compiler may not use the scalar cache — and the fields go to scalar registers by hand)
    const int x0 = __builtin_amdgcn_readfirstlane(r0.x), w1 = __builtin_amdgcn_readfirstlane(r1.w);
    r.ntrk = x0 & 0xff; r.ncam = (x0 >> 8) & 0xff; r.npair = (x0 >> 16) & 0xff; r.flags = (x0 >> 24) & 0xff;
    r.cam0 = __builtin_amdgcn_readfirstlane(r0.w); r.pair0 = __builtin_amdgcn_readfirstlane(r1.x); r.trk0 = __builtin_amdgcn_readfirstlane(r1.y);
    r.it0 = __builtin_amdgcn_readfirstlane(r1.z); r.lgS = w1 & 0xff; r.nit = w1 >> 8;
    return r;
}
__device__ __forceinline__ Rec load_rec(const PlanDev &pd, int t) { return decode_rec(load_raw(pd, t)); }

template <int P> struct IC { static constexpr int value = P; };

__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float x) { return f2{x, x}; }

#define BT_DPPF(x, ctrl) __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), (ctrl), 0xf, 0xf, true))
template <int M>
__device__ __forceinline__ float xor_add(float x) {
    if (M == 1) return x + BT_DPPF(x, 0xb1);
    if (M == 2) return x + BT_DPPF(x, 0x4e);
    if (M == 4) {
        int t = __builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0x104, 0xf, 0x5, false);
        t = __builtin_amdgcn_update_dpp(t, (int)__float_as_uint(x), 0x114, 0xf, 0xa, false);
        return x + __uint_as_float((unsigned)t);
    }
    if (M == 8) return x + BT_DPPF(x, 0x128);
    if (M == 16) { const uint2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false); return __uint_as_float(r.x) + __uint_as_float(r.y); }
    const uint2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// sums over the groups of 2^lg adjacent lanes, both halves of the 2-vectors (every lane of a group gets the totals)
template <int N>
__device__ __forceinline__ void group_sum2(f2 (&x)[N], int lg) {
#define BT_GS(ctrl) { _Pragma("unroll") for (int i = 0; i < N; ++i) { x[i].x += BT_DPPF(x[i].x, ctrl); x[i].y += BT_DPPF(x[i].y, ctrl); } }
    if (lg > 0) BT_GS(0xb1)
    if (lg > 1) BT_GS(0x4e)
    if (lg > 2) BT_GS(0x141)
    if (lg > 3) BT_GS(0x140)
#undef BT_GS
    if (lg > 4) { _Pragma("unroll") for (int i = 0; i < N; ++i) { x[i].x = xor_add<16>(x[i].x); x[i].y = xor_add<16>(x[i].y); } }
    if (lg > 5) { _Pragma("unroll") for (int i = 0; i < N; ++i) { x[i].x = xor_add<32>(x[i].x); x[i].y = xor_add<32>(x[i].y); } }
}
// sum of x over the lanes with the same (lane mod 2^lg) (lane < 2^lg then holds the total of its residue class)
template <int N>
__device__ __forceinline__ void stride_sum(float (&x)[N], int lg) {
    if (lg <= 5) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] = xor_add<32>(x[i]); }
    if (lg <= 4) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] = xor_add<16>(x[i]); }
    if (lg <= 3) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] = xor_add<8>(x[i]); }
    if (lg <= 2) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] = xor_add<4>(x[i]); }
    if (lg <= 1) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] = xor_add<2>(x[i]); }
    if (lg <= 0) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] = xor_add<1>(x[i]); }
}

// 1 / x in double from the float32 hardware seed and ONE Newton step: the seed is 1 ulp (2^-23) off, the step squares that
// (< 1e-13 relative; u = fx X / Z + cx ~ 500 px then carries < 1e-10 px)
__device__ __forceinline__ double rcp_nr1(double x) {
    const double y = (double)__builtin_amdgcn_rcpf((float)x);
    return fma(y, fma(-x, y, 1.0), y);
}

// float64 part of one edge (projective_ops.py:19-66, ba.py:230-242): the point in the target frame, the residual and the
// validity decided on the float64 values.  gd = the pair's geometry in LDS as doubles.
struct Proj { float X, Y, Z, r0, r1; bool ok; };
__device__ __forceinline__ Proj project(const double (&gd)[kGeoD], double X0, double Y0, float d, float tu, float tv, bool act,
                                        double b0, double b1, double b2, double b3) {
    const double dd = (double)d;
    const double Xd = fma(gd[0], X0, fma(gd[1], Y0, fma(gd[9], dd, gd[2])));
    const double Yd = fma(gd[3], X0, fma(gd[4], Y0, fma(gd[10], dd, gd[5])));
    const double Zd = fma(gd[6], X0, fma(gd[7], Y0, fma(gd[11], dd, gd[8])));
    const double iz = rcp_nr1(fmax(Zd, 1e-2));
    const double ud = fma(gd[12], iz * Xd, gd[14]), vd = fma(gd[13], iz * Yd, gd[15]);
    const double r0d = (double)tu - ud, r1d = (double)tv - vd;
    Proj p;
    p.ok = act && Zd > 0.2 && fma(r0d, r0d, r1d * r1d) < 62500.0 && ud > b0 && vd > b1 && ud < b2 && vd < b3;
    p.X = (float)Xd; p.Y = (float)Yd; p.Z = (float)Zd; p.r0 = (float)r0d; p.r1 = (float)r1d;
    return p;
}

// 1 / x in float32: the hardware seed (1 ulp) and a Newton step (then inside the last ulp)
__device__ __forceinline__ float rcp_f32(float x) {
    const float y = __builtin_amdgcn_rcpf(x);
    return fmaf(y, fmaf(-x, y, 1.0f), y);
}

template <int LOSS>
__device__ __forceinline__ float robust1(float s) {          // ba.py:81-100 (s = r * r)
    if (LOSS == BT_LOSS_HUBER) return s > 1.0f ? __builtin_amdgcn_rsqf(s) : 1.0f;
    if (LOSS == BT_LOSS_CAUCHY) return __builtin_amdgcn_rcpf(1.0f + s);
    return 1.0f;
}


#define BT_E2_WAVE_SYNC()                                        \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

}  // namespace e2
}  // namespace bt
