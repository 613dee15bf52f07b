// ba_stream3.hip — k_edge: the Jacobian kernel for graphs of many SLOT-UNIFORM tiles (gfx950, wave64).
//
// Layout (ba_plan.cpp "edge-major"): a tile is still 64 tracks with their local E in LDS, but a wave's 64 lanes
// are 64 / S consecutive tracks x their S slots (S = slots per track, padded to a power of two), lane = track_in_group
// * S + slot.  On the track-major edge lists the caller builds (batrack.py:399-410: every patch of a keyframe x every
// frame of the window) that is 64 CONSECUTIVE edges per wave instruction — the coalesced stream of targets and weights —
// and it removes the two things that made a slot of k_tile / k_stream expensive:
//   * a lane meets the same camera pair in every iteration of a tile (the pair of slot s), so the 27 per-pair products
//     (ba.py:260,266) are summed per lane in registers and reduced across lanes once per tile, not once per slot;
//   * a track's sums (C, w, its source-camera E; ba.py:284-292) are reductions over S adjacent lanes (DPP), and its
//     target-camera E entries are plain LDS stores (one lane per (track, camera); LDS atomics only where the plan
//     marks a repeated observation).
// Every wave owns whole tiles and walks a contiguous range of them with no barrier (workgroup = one wave, private LDS);
// the operands of iteration i+1 are gathered and the edge ids of iteration i+2 loaded while iteration i is computed —
// one flat pipeline across tile boundaries.  The Schur product E Q E^T runs on v_mfma_f32_16x16x4_f32 in partial sums of 16
// tracks that are added to float64 accumulators kept in registers across consecutive tiles with the same cameras.
// MODE kEmSO: structure-only steps; kEmUpd: the depth back-substitution (see k_update in ba_kernels.hip).
// Reference: ba.py:228-337, projective_ops.py:54-100.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>

#include "ba_edge.hpp"
#include "ba_kernels.hpp"

namespace bt {

enum { kEmFull = 0, kEmSO = 1, kEmUpd = 2 };
constexpr int kEmUpdGeo = 28;          // floats per pair in LDS for kEmUpd: geometry (20), delta (6), padding

struct EmRec { int ntrk, ncam, npair, flags, cam0, pair0, trk0, it0, lgS, nit; };

__device__ __forceinline__ EmRec em_load_rec(const PlanDev &pd, int t) {
    const int4 r0 = reinterpret_cast<const int4 *>(pd.tile_rec)[2 * t], r1 = reinterpret_cast<const int4 *>(pd.tile_rec)[2 * t + 1];
    EmRec r;
    r.ntrk = r0.x & 0xff; r.ncam = (r0.x >> 8) & 0xff; r.npair = (r0.x >> 16) & 0xff; r.flags = (r0.x >> 24) & 0xff;
    r.cam0 = r0.w; r.pair0 = r1.x; r.trk0 = r1.y; r.it0 = r1.z; r.lgS = r1.w & 0xff; r.nit = r1.w >> 8;
    return r;
}

#define BT_DPPF(x, ctrl) __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), (ctrl), 0xf, 0xf, true))
// x + (x of lane ^ M), exact partner: M = 1, 2, 8 as the DPP operand of the add, 4 as two masked DPP moves, 16 / 32 by the
// lane-swap instructions
template <int M>
__device__ __forceinline__ float xor_add(float x) {
    if (M == 1) return x + BT_DPPF(x, 0xb1);
    if (M == 2) return x + BT_DPPF(x, 0x4e);
    if (M == 4) {
        int t = __builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0x104, 0xf, 0x5, false);       // row_shl:4 on the banks with bit 2 clear
        t = __builtin_amdgcn_update_dpp(t, (int)__float_as_uint(x), 0x114, 0xf, 0xa, false);           // row_shr:4 on the others
        return x + __uint_as_float((unsigned)t);
    }
    if (M == 8) return x + BT_DPPF(x, 0x128);
    if (M == 16) { const uint2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false); return __uint_as_float(r.x) + __uint_as_float(r.y); }
    const uint2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// sum of x over the groups of 2^lg adjacent lanes (every lane of a group gets the total); lg is wave-uniform.  Inside a
// row the partners are quad_perm (lane ^ 1, lane ^ 2), row_half_mirror (7 - lane of the 8) and row_mirror (15 - lane of the 16)
template <int N>
__device__ __forceinline__ void group_sum(float (&x)[N], int lg) {
    if (lg > 0) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] += BT_DPPF(x[i], 0xb1); }
    if (lg > 1) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] += BT_DPPF(x[i], 0x4e); }
    if (lg > 2) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] += BT_DPPF(x[i], 0x141); }
    if (lg > 3) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] += BT_DPPF(x[i], 0x140); }
    if (lg > 4) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] = xor_add<16>(x[i]); }
    if (lg > 5) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] = xor_add<32>(x[i]); }
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

#ifndef BT_EDGE_SCHUR_CHUNK
// E Q E^T: the float64 matrix pipe (v_mfma_f64_16x16x4_f64, 64 cycles) was a quarter of the kernel's time.  E in LDS is float32
// already, so the products go to the float32 pipe (half the cycles) in partial sums of N x 4 tracks, each added to the float64
// accumulators that run across the tiles of a camera set: S, dX and the update are as close to the float64 oracle as with the
// float64 pipe (measured: S 1.39e-6, dX 2.1e-5 either way on the golden cases).  0 selects the float64 pipe.
#define BT_EDGE_SCHUR_CHUNK 4
#endif

// E Q E^T of one tile into the float64 register accumulators, and E (Q w') — the Schur term of y (ba.py:311) — as one more
// product on the float32 matrix pipe: the A operand (16 rows of E x 4 tracks) is the one the Schur tiles load anyway, B holds
// beta = Q w' of the 4 tracks in every column, so every column of the 16x16 result is the tile's y rows; column 0 adds them
// to the wave's float64 sums in LDS (the same float32-within-a-tile / float64-across-tiles summation as before, without the
// 2 x 31 cross-lane steps it took on the VALU).
template <int NT>
__device__ __forceinline__ void em_schur(const float *Eh, const float *Qs, const float *Bs, double *ysum, int R, int lane,
                                         double4_t (&acc)[NT * (NT + 1) / 2]) {
    const int kq = lane >> 4, li = lane & 15;
    float qv[16];
    const float *qp = Qs + kq;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) qv[ks] = qp[4 * ks];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        if (16 * ti < R) {
            // rows beyond the tile's E re-read its last row: their outputs are never emitted
            const float *ap = Eh + min(16 * ti + li, R - 1) * kLdsRowStride + kq;
            float af[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) af[ks] = ap[4 * ks];
            {
                float4_t yt = {0.0f, 0.0f, 0.0f, 0.0f};
                const float *bq = Bs + kq;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) yt = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks], bq[4 * ks], yt, 0, 0, 0);
                // f32 C/D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + 4 * kq + r;
                    if (li == 0 && row < R) atomicAdd(ysum + row, (double)yt[r]);
                }
            }
#if BT_EDGE_SCHUR_CHUNK == 0
            double av[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) av[ks] = (double)af[ks] * (double)qv[ks];
#else
            float aq[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) aq[ks] = af[ks] * qv[ks];
#endif
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj) {
                __builtin_amdgcn_sched_barrier(0);          // one output tile's operands at a time
                const float *bp = Eh + min(16 * tj + li, R - 1) * kLdsRowStride + kq;
                float bv[16];
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) bv[ks] = bp[4 * ks];
#if BT_EDGE_SCHUR_CHUNK == 0
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
                    acc[ti * (ti + 1) / 2 + tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], (double)bv[ks], acc[ti * (ti + 1) / 2 + tj], 0, 0, 0);
#else
#pragma unroll
                for (int k0 = 0; k0 < 16; k0 += BT_EDGE_SCHUR_CHUNK) {
                    float4_t c = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int ks = k0; ks < k0 + BT_EDGE_SCHUR_CHUNK; ++ks) c = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[ks], bv[ks], c, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[ti * (ti + 1) / 2 + tj][r] += (double)c[r];
                }
#endif
            }
        }
    }
}

#ifndef BT_EDGE_FULL_WAVES
#define BT_EDGE_FULL_WAVES 2
#endif
#ifndef BT_EDGE_DEPTH
#define BT_EDGE_DEPTH 3
#endif
constexpr int kEmDepth = BT_EDGE_DEPTH;      // iterations of gathered operands in flight per wave
constexpr int kEmLead = BT_EDGE_DEPTH;       // and the edge ids run this many iterations ahead of the gathers through them (the
                                             // memory counter retires loads in order: an id needed right after its load would
                                             // drain the whole pipeline every iteration)
constexpr int kEmIds = kEmDepth + kEmLead;

// LGS: log2 of the slots per track when every tile of the plan has the same (the lane-group reductions are then straight-line
// code), -1: read per tile
template <int MODE, int NT, int LGS, bool PROF = false>
__global__ __launch_bounds__(64, MODE == kEmFull ? BT_EDGE_FULL_WAVES : 4) void k_edge(PlanDev pd, StepArgs a, int tiles_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    long long pf[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tc = PROF ? clock64() : 0, tn;
#define BT_PF(i) do { if (PROF) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tn = clock64(); pf[i] += tn - tc; tc = tn; __builtin_amdgcn_sched_barrier(0); } } while (0)
    const int mtp = pd.max_tile_pairs > 0 ? pd.max_tile_pairs : 1;
    const int Rmax = 6 * pd.max_cams;
    constexpr int GS = MODE == kEmUpd ? kEmUpdGeo : kPairGeomFloats;
    // LDS carve-up (wave-private)
    float *geo = lds;                                                  // [mtp][GS]
    float *ptab = geo + mtp * GS;                                      // [64][4]: x, y, disparity, local source camera of every track of the tile
    float *cw = ptab + 256;                                            // [2][64]: C, w per track (kEmUpd: [0][64] = patch index)
    float *Eh = cw + 128;                                              // [Rmax][66]: local E
    float *Qs = Eh + (MODE == kEmFull ? Rmax * kLdsRowStride : 0);
    int *gidx = reinterpret_cast<int *>(Qs + (MODE == kEmFull ? 64 : 0));
    int *gpl = gidx + (MODE == kEmFull ? ((Rmax + 3) & ~3) : 0);
    double *pacc = reinterpret_cast<double *>(gpl + (MODE == kEmFull ? ((mtp + 3) & ~3) : 0));   // [mtp][32]
    double *ysum = pacc + (MODE == kEmFull ? mtp * 32 : 0);                                       // [64]: the wave's sums of E Q w' by local row

    // (workgroup -> tile range stays plain: handing each XCD a contiguous block of ranges, which helps k_tile and k_stream,
    //  made this kernel 4 % slower at 8.4M edges)
    const int wg_ = blockIdx.x;
    const int t_begin = wg_ * tiles_per_wave, t_end = min(pd.T, t_begin + tiles_per_wave);
    if (t_begin >= t_end) return;

    constexpr int NACC = NT * (NT + 1) / 2;
    double4_t sacc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) sacc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
    bool acc_live = false;
    int Racc = 0, np_acc = 0;

    auto flush_schur = [&]() {
        if (MODE != kEmFull || !acc_live) return;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj) {
                const int t = ti * (ti + 1) / 2 + tj;
                const int col = 16 * tj + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = BT_EDGE_SCHUR_CHUNK ? 16 * ti + 4 * (lane >> 4) + r : 16 * ti + (lane >> 4) + 4 * r;
                    const double val = sacc[t][r];
                    sacc[t][r] = 0.0;
                    if (row < Racc && col < Racc) {
                        const int gr = gidx[row], gc = gidx[col];
                        if (gr >= gc) atomicAdd(&a.S[(size_t)gr * pd.D + gc], -val);
                    }
                }
            }
        if (lane < Racc) {
            const double v = ysum[lane];
            ysum[lane] = 0.0;
            atomicAdd(&a.y[gidx[lane]], -v);
        }
        acc_live = false;
    };
    auto flush_pairs = [&]() {
        if (MODE != kEmFull) return;
        for (int p0 = 0; p0 < np_acc; p0 += 2) {
            const int p = p0 + (lane >> 5), vi = lane & 31;
            if (p < np_acc && vi < 27) {
                double *src = pacc + p * 32 + vi;
                const double val = *src;
                if (val != 0.0) atomicAdd(&a.pairacc[(size_t)gpl[p] * kPairAccStride + vi], val);
                *src = 0.0;
            }
        }
        np_acc = 0;
    };
    if (MODE == kEmFull) {
        for (int i = lane; i < mtp * 32; i += 64) pacc[i] = 0.0;
        ysum[lane] = 0.0;
    }

    // ---- the flat iteration stream of this wave: [gi, gi_end)
    EmRec rec = em_load_rec(pd, t_begin);
    EmRec rec_n = t_begin + 1 < t_end ? em_load_rec(pd, t_begin + 1) : rec;
    int gi = rec.it0;
    int gi_end;
    { const EmRec last = em_load_rec(pd, t_end - 1); gi_end = last.it0 + last.nit; }
    auto gather = [&](int e, float &tu, float &tv, float &w0, float &w1) {
        tu = tv = w0 = w1 = 0.0f;
        if (e >= 0) {
            const unsigned to = (unsigned)e * (unsigned)a.tstride;               // (launch_edge checks that byte offsets fit 32 bits)
            tu = a.targets[to]; tv = a.targets[to + 1u];
            const float2 w = reinterpret_cast<const float2 *>(a.weights)[(unsigned)e];
            w0 = w.x; w1 = w.y;
        }
    };
    // operand pipeline, kEmDepth iterations deep: the HBM round trip under load is several iterations long, and the bytes a wave
    // keeps in flight are what its share of the bandwidth is made of.  Slot 0 = the current iteration.
    int e_q[kEmIds];
    float tu_q[kEmDepth], tv_q[kEmDepth], w0_q[kEmDepth], w1_q[kEmDepth];
#pragma unroll
    for (int k = 0; k < kEmIds; ++k) e_q[k] = gi + k < gi_end ? pd.it_edge[(unsigned)(gi + k) * kLanes + (unsigned)lane] : -1;
#pragma unroll
    for (int k = 0; k < kEmDepth; ++k) gather(e_q[k], tu_q[k], tv_q[k], w0_q[k], w1_q[k]);

    // ---- per-tile context of the first tile (lane = track of the tile)
    int kx_c = pd.tile_kx[(unsigned)t_begin * kLanes + (unsigned)lane];
    unsigned la_c = pd.tile_la[(unsigned)t_begin * kLanes + (unsigned)lane];
    unsigned si_c = pd.tile_sinfo[(unsigned)t_begin * kLanes + (unsigned)(lane & ((1 << rec.lgS) - 1))];
    float px = 0.0f, py = 0.0f, pdisp = 0.0f, mono_v = 0.0f;
    if (kx_c >= 0) {
        px = a.patches[3u * (unsigned)kx_c]; py = a.patches[3u * (unsigned)kx_c + 1u]; pdisp = a.patches[3u * (unsigned)kx_c + 2u];
        if (MODE != kEmUpd) mono_v = a.mono[(unsigned)kx_c * (unsigned)a.mstride];
    }

#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int flags = tile == t_begin ? 0 : rec.flags;
        const int R = 6 * rec.ncam;
        const bool has_next = tile + 1 < t_end;
        const int lgS = LGS >= 0 ? LGS : rec.lgS, S1 = (1 << lgS) - 1, G = kLanes >> lgS;
        // ---- new cameras / new pair list
        if (MODE == kEmFull && !(flags & 1)) {
            flush_schur();
            const int *cams = pd.tile_cams + rec.cam0;
            for (int i = lane; i < R; i += 64) gidx[i] = 6 * cams[i / 6] + i % 6;
        }
        if (!(flags & 2)) {
            flush_pairs();
            for (int p = lane; p < rec.npair; p += 64) {
                const int gp = pd.tile_pairs[rec.pair0 + p];
                float *g = geo + p * GS;
                if (MODE == kEmUpd) {
                    const int ia = pd.pair_i[gp] - pd.fixedp, ib = pd.pair_j[gp] - pd.fixedp;
                    float gg[kPairGeomFloats];
                    const float4 *src = reinterpret_cast<const float4 *>(a.pairgeo + (size_t)gp * kPairGeomFloats);
#pragma unroll
                    for (int c = 0; c < kPairGeomFloats / 4; ++c) { const float4 t4 = src[c]; gg[4*c] = t4.x; gg[4*c + 1] = t4.y; gg[4*c + 2] = t4.z; gg[4*c + 3] = t4.w; }
                    float xi[6] = {0, 0, 0, 0, 0, 0}, xj[6] = {0, 0, 0, 0, 0, 0};
                    if (ia >= 0) for (int c = 0; c < 6; ++c) xi[c] = a.dx[6 * ia + c];
                    if (ib >= 0) for (int c = 0; c < 6; ++c) xj[c] = a.dx[6 * ib + c];
                    float Rt[3], Rp[3];                                  // Ad(Gij)(tau, phi) = (R tau + t x (R phi), R phi)   (se3.h:58-67)
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        Rt[r] = gg[3*r] * xi[0] + gg[3*r + 1] * xi[1] + gg[3*r + 2] * xi[2];
                        Rp[r] = gg[3*r] * xi[3] + gg[3*r + 1] * xi[4] + gg[3*r + 2] * xi[5];
                    }
#pragma unroll
                    for (int c = 0; c < kPairGeomFloats; ++c) g[c] = gg[c];
                    g[20] = xj[0] - (Rt[0] + gg[10] * Rp[2] - gg[11] * Rp[1]);
                    g[21] = xj[1] - (Rt[1] + gg[11] * Rp[0] - gg[9]  * Rp[2]);
                    g[22] = xj[2] - (Rt[2] + gg[9]  * Rp[1] - gg[10] * Rp[0]);
                    g[23] = xj[3] - Rp[0]; g[24] = xj[4] - Rp[1]; g[25] = xj[5] - Rp[2];
                    g[26] = 0.0f; g[27] = 0.0f;
                } else {
                    const int ij = pd.tile_ij[(unsigned)tile * (unsigned)mtp + (unsigned)p];
                    pair_geometry<float, BT_WPT_MIXED != 0>(a.poses, a.intr, ij & 0xffff, ij >> 16, g);
                    if (MODE == kEmFull) {
                        gpl[p] = gp;
                        float4 *dst = reinterpret_cast<float4 *>(a.pairgeo + (size_t)gp * kPairGeomFloats);
                        const float4 *src = reinterpret_cast<const float4 *>(g);
#pragma unroll
                        for (int c = 0; c < kPairGeomFloats / 4; ++c) dst[c] = src[c];
                    }
                }
            }
            np_acc = rec.npair;
        }
        // the tracks' patches where every lane of their group can read them
        reinterpret_cast<float4 *>(ptab)[lane] = make_float4(px, py, pdisp, __uint_as_float(la_c));
        if (MODE == kEmUpd) reinterpret_cast<int *>(cw)[lane] = kx_c;
        if (MODE == kEmFull) {
            float4 *z = reinterpret_cast<float4 *>(Eh);
            for (int i = lane; i < (R * kLdsRowStride + 3) / 4; i += 64) z[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        // this lane's slot of the tile: target camera, pair, repeated observation
        const unsigned si = si_c;
        const unsigned lb = si & 0xffu, lp = (si >> 8) & 0xffu;
        const bool rep = (si >> 16) & 1u, used = (si >> 17) & 1u;
        const int tl = lane >> lgS;
        const bool lead = (lane & S1) == 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- next tile's context: requested now, lands under this tile's iterations
        int kx_n = -1;
        unsigned la_n = 0xffu, si_n = 0u;
        float px_n = 0.0f, py_n = 0.0f, pd_n = 0.0f, mono_n = 0.0f;
        EmRec rec_nn = rec_n;
        if (tile + 2 < t_end) rec_nn = em_load_rec(pd, tile + 2);
        if (has_next) {
            kx_n = pd.tile_kx[(unsigned)(tile + 1) * kLanes + (unsigned)lane];
            la_n = pd.tile_la[(unsigned)(tile + 1) * kLanes + (unsigned)lane];
            si_n = pd.tile_sinfo[(unsigned)(tile + 1) * kLanes + (unsigned)(lane & ((1 << rec_n.lgS) - 1))];
        }
        BT_PF(0);

        float pa[26];
#pragma unroll
        for (int i = 0; i < 26; ++i) pa[i] = 0.0f;
#pragma unroll 1
        for (int it = 0; it < rec.nit; ++it, ++gi) {
            // ---- pipeline: gather iteration gi + kEmDepth through the edge ids loaded last time, load the ids of the one after
            const int e = e_q[0];
            const float tu = tu_q[0], tv = tv_q[0], w0 = w0_q[0], w1 = w1_q[0];
            float tu_x, tv_x, w0_x, w1_x;
            gather(e_q[kEmDepth], tu_x, tv_x, w0_x, w1_x);
            const int e_nn = gi + kEmIds < gi_end ? pd.it_edge[(unsigned)(gi + kEmIds) * kLanes + (unsigned)lane] : -1;
            if (it == 0 && has_next && kx_n >= 0) {          // (its index has arrived by now: the first iteration's arithmetic lies between)
                px_n = a.patches[3u * (unsigned)kx_n]; py_n = a.patches[3u * (unsigned)kx_n + 1u]; pd_n = a.patches[3u * (unsigned)kx_n + 2u];
                if (MODE != kEmUpd) mono_n = a.mono[(unsigned)kx_n * (unsigned)a.mstride];
            }

            BT_PF(8);
            const bool act = e >= 0;
            const int track = it * G + tl;
            const float4 pt = reinterpret_cast<const float4 *>(ptab)[track];
            const unsigned la = __float_as_uint(pt.w);
            float g[GS];
            {
                const float4 *g4 = reinterpret_cast<const float4 *>(geo + (size_t)lp * GS);
#pragma unroll
                for (int c = 0; c < GS / 4; ++c) {
                    const float4 t4 = g4[c];
                    g[4*c] = t4.x; g[4*c + 1] = t4.y; g[4*c + 2] = t4.z; g[4*c + 3] = t4.w;
                }
            }
            BT_PF(9);
            EdgeQ q;
            BT_WPT_EDGE_EVAL(g, pt.x, pt.y, pt.z, tu, tv, w0, w1, a, q);
            if (!act) { q.W0 = 0.0f; q.W1 = 0.0f; q.r0 = 0.0f; q.r1 = 0.0f; }
            BT_PF(10);
            if (MODE == kEmUpd) {
                float dv[1] = {0.0f};
                if (act) {
                    const float d0 = q.a0 * g[20] + q.a2 * g[22] + q.a3 * g[23] + q.a4 * g[24] + q.a5 * g[25];
                    const float d1 = q.b1 * g[21] + q.b2 * g[22] + q.b3 * g[23] + q.b4 * g[24] + q.b5 * g[25];
                    dv[0] = q.W0 * q.jz0 * d0 + q.W1 * q.jz1 * d1;
                }
                group_sum(dv, lgS);
                if (lead && track < rec.ntrk) {
                    const float2 qw = a.qw[(unsigned)rec.trk0 + (unsigned)track];
                    const unsigned kx = (unsigned)reinterpret_cast<const int *>(cw)[track];
                    float dd = pt.z + qw.x * (qw.y - dv[0]);                        // ba.py:328, :333
                    dd = dd < 1e-3f ? 1e-3f : dd;
                    dd = dd > 10.0f ? 10.0f : dd;
                    a.patches_out[3u * kx] = pt.x; a.patches_out[3u * kx + 1u] = pt.y; a.patches_out[3u * kx + 2u] = dd;
                }
            } else if (MODE == kEmSO) {
                float sv[2] = { q.W0 * q.jz0 * q.jz0 + q.W1 * q.jz1 * q.jz1, q.W0 * q.jz0 * q.r0 + q.W1 * q.jz1 * q.r1 };   // ba.py:287,292
                group_sum(sv, lgS);
                if (lead && track < rec.ntrk) { cw[track] = sv[0]; cw[64 + track] = sv[1]; }
            } else {
                const float wa0 = q.W0 * q.a0, wa2 = q.W0 * q.a2, wa3 = q.W0 * q.a3, wa4 = q.W0 * q.a4, wa5 = q.W0 * q.a5;
                const float wb1 = q.W1 * q.b1, wb2 = q.W1 * q.b2, wb3 = q.W1 * q.b3, wb4 = q.W1 * q.b4, wb5 = q.W1 * q.b5;
                // Ej = Jj^T W Jz (ba.py:263) and Ei = -Ad^T Ej
                const float Ej[6] = { wa0 * q.jz0, wb1 * q.jz1, fmaf(wa2, q.jz0, wb2 * q.jz1), fmaf(wa3, q.jz0, wb3 * q.jz1),
                                      fmaf(wa4, q.jz0, wb4 * q.jz1), fmaf(wa5, q.jz0, wb5 * q.jz1) };
                // the track's sums over its S lanes: C, w (ba.py:287,292) and its source-camera E
                float sv[8] = { q.W0 * q.jz0 * q.jz0 + q.W1 * q.jz1 * q.jz1, q.W0 * q.jz0 * q.r0 + q.W1 * q.jz1 * q.r1,
                                0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
                if (act && la != 0xffu) {
                    // o_tau = R^T e_tau ; o_phi = R^T (e_tau x t + e_phi)      (se3.h:58-67)
                    const float cx = Ej[1]*g[11] - Ej[2]*g[10] + Ej[3];
                    const float cy = Ej[2]*g[9]  - Ej[0]*g[11] + Ej[4];
                    const float cz = Ej[0]*g[10] - Ej[1]*g[9]  + Ej[5];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        sv[2 + c] = -(g[c]*Ej[0] + g[3 + c]*Ej[1] + g[6 + c]*Ej[2]);
                        sv[5 + c] = -(g[c]*cx + g[3 + c]*cy + g[6 + c]*cz);
                    }
                }
                BT_PF(11);
                group_sum(sv, lgS);
                BT_PF(12);
                // target-camera E of this edge: one lane per (track, camera) unless the plan marks a repeat
                if (act && lb != 0xffu) {
                    float *row = Eh + lb * 6 * kLdsRowStride + track;
                    if (rep) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) atomicAdd(row + c * kLdsRowStride, Ej[c]);
                    } else {
#pragma unroll
                        for (int c = 0; c < 6; ++c) row[c * kLdsRowStride] = Ej[c];
                    }
                }
                if (lead && track < rec.ntrk) {
                    cw[track] = sv[0]; cw[64 + track] = sv[1];
                    if (la != 0xffu) {
                        float *row = Eh + la * 6 * kLdsRowStride + track;
                        if (pd.em_self) {        // (behind the stores above in LDS order: a self edge's target rows are these)
#pragma unroll
                            for (int c = 0; c < 6; ++c) row[c * kLdsRowStride] += sv[2 + c];
                        } else {                 // a track's source-camera row is written here and nowhere else
#pragma unroll
                            for (int c = 0; c < 6; ++c) row[c * kLdsRowStride] = sv[2 + c];
                        }
                    }
                }
                BT_PF(13);
                // per-pair sums Bjj (21, row-major upper triangle) and gj (6) (ba.py:260,266), per lane: the lane's pair is
                // the same in every iteration of the tile
                pa[0] = fmaf(wa0, q.a0, pa[0]);   pa[1] = fmaf(wa0, q.a2, pa[1]);   pa[2] = fmaf(wa0, q.a3, pa[2]);
                pa[3] = fmaf(wa0, q.a4, pa[3]);   pa[4] = fmaf(wa0, q.a5, pa[4]);
                pa[5] = fmaf(wb1, q.b1, pa[5]);   pa[6] = fmaf(wb1, q.b2, pa[6]);   pa[7] = fmaf(wb1, q.b3, pa[7]);
                pa[8] = fmaf(wb1, q.b4, pa[8]);   pa[9] = fmaf(wb1, q.b5, pa[9]);
                pa[10] = fmaf(wa2, q.a2, fmaf(wb2, q.b2, pa[10])); pa[11] = fmaf(wa2, q.a3, fmaf(wb2, q.b3, pa[11]));
                pa[12] = fmaf(wa2, q.a4, fmaf(wb2, q.b4, pa[12])); pa[13] = fmaf(wa2, q.a5, fmaf(wb2, q.b5, pa[13]));
                pa[14] = fmaf(wa3, q.a3, fmaf(wb3, q.b3, pa[14])); pa[15] = fmaf(wa3, q.a4, fmaf(wb3, q.b4, pa[15]));
                pa[16] = fmaf(wa3, q.a5, fmaf(wb3, q.b5, pa[16]));
                pa[17] = fmaf(wa4, q.a4, fmaf(wb4, q.b4, pa[17])); pa[18] = fmaf(wa4, q.a5, fmaf(wb4, q.b5, pa[18]));
                pa[19] = fmaf(wa5, q.a5, fmaf(wb5, q.b5, pa[19]));
                pa[20] = fmaf(wa0, q.r0, pa[20]); pa[21] = fmaf(wb1, q.r1, pa[21]);
                pa[22] = fmaf(wa2, q.r0, fmaf(wb2, q.r1, pa[22])); pa[23] = fmaf(wa3, q.r0, fmaf(wb3, q.r1, pa[23]));
                pa[24] = fmaf(wa4, q.r0, fmaf(wb4, q.r1, pa[24])); pa[25] = fmaf(wa5, q.r0, fmaf(wb5, q.r1, pa[25]));
            }
            BT_PF(14);
            // rotate the pipeline
#pragma unroll
            for (int k = 0; k + 1 < kEmIds; ++k) e_q[k] = e_q[k + 1];
            e_q[kEmIds - 1] = e_nn;
#pragma unroll
            for (int k = 0; k + 1 < kEmDepth; ++k) { tu_q[k] = tu_q[k + 1]; tv_q[k] = tv_q[k + 1]; w0_q[k] = w0_q[k + 1]; w1_q[k] = w1_q[k + 1]; }
            tu_q[kEmDepth - 1] = tu_x; tv_q[kEmDepth - 1] = tv_x; w0_q[kEmDepth - 1] = w0_x; w1_q[kEmDepth - 1] = w1_x;
        }
        BT_PF(1);

        const bool has_trk = lane < rec.ntrk;
        if (MODE != kEmUpd) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (MODE == kEmFull) {
                // the lanes' pair sums: over the lanes with the same slot, then lane s (< S) adds the sums of slot s's pair to
                // the wave's float64 sums (LDS atomics: two slots may hold the same pair)
                stride_sum(pa, lgS);
                if (lane <= S1 && used) {
                    double *dst = pacc + lp * 32;
#pragma unroll
                    for (int i = 0; i < 26; ++i) {
                        // element order of the 27-vector: Bjj row-major upper triangle with its structural zero at [0][1]
                        const int vi = i < 1 ? 0 : i + 1;
                        atomicAdd(dst + vi, (double)pa[i]);
                    }
                }
            }
            float Q = 0.0f, wp = 0.0f;                                            // ba.py:296-311
            if (has_trk) {
                const float C = cw[lane], wv = cw[64 + lane];
                const float pm = mono_v > 1e-2f ? 1.0f : 0.0f;
                float Ca = C + pm * a.alpha;
                Ca = Ca + (a.lmbda_trk ? a.lmbda_trk[(unsigned)pd.trk_off + (unsigned)rec.trk0 + (unsigned)lane] : a.lmbda);
                wp = wv - pm * a.alpha * (pdisp - mono_v);
                Q = 1.0f / Ca;
                a.qw[(unsigned)rec.trk0 + (unsigned)lane] = make_float2(Q, wp);
            }
            if (MODE == kEmFull) {
                Qs[lane] = Q;
                cw[lane] = Q * wp;                     // beta (this lane read its own C above; w of the tile is consumed)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                BT_PF(2);
                BT_PF(3);
                em_schur<NT>(Eh, Qs, cw, ysum, R, lane, sacc);
                acc_live = true; Racc = R;
                BT_PF(4);
            }
        }

        // ---- rotate the tile context
        if (has_next) {
            rec = rec_n; rec_n = rec_nn; kx_c = kx_n; la_c = la_n; si_c = si_n;
            px = px_n; py = py_n; pdisp = pd_n; mono_v = mono_n;
        }
    }
    flush_pairs();
    flush_schur();
    BT_PF(5);
    if (PROF && lane == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BT_PF(6);
        long long *o = reinterpret_cast<long long *>(a.status + 4) + (blockIdx.x == 0 ? 20 : 30);
        for (int i = 0; i < 8; ++i) o[i] = pf[i];
        o[8] = t_end - t_begin;
        long long *o2 = reinterpret_cast<long long *>(a.status + 4) + (blockIdx.x == 0 ? 40 : 50);
        for (int i = 0; i < 8; ++i) o2[i] = pf[8 + i];
    }
#undef BT_PF
}
#undef BT_DPPF

// ------------------------------------------------------------------ dispatch

static size_t edge_lds_bytes(const PlanDev &pd, int mode) {
    const size_t mtp = (size_t)(pd.max_tile_pairs > 0 ? pd.max_tile_pairs : 1);
    const size_t Rmax = (size_t)(6 * pd.max_cams);
    if (mode == kEmUpd) return (mtp * kEmUpdGeo + 256 + 128) * sizeof(float);
    if (mode == kEmSO) return (mtp * kPairGeomFloats + 256 + 128) * sizeof(float);
    return (mtp * kPairGeomFloats + 256 + 128 + Rmax * kLdsRowStride + 64 + ((Rmax + 3) & ~(size_t)3) + ((mtp + 3) & ~(size_t)3)) * sizeof(float) +
           (mtp * 32 + 64) * sizeof(double) + 16;
}

// k_edge takes graphs of many tiles, all slot-uniform (the plan's em_ok), whose tiles see at most 10 cameras (row
// tiles of the register accumulators) and 64 camera pairs (one lane per pair in the prologue)
// Measured on the benchmark generator (profiles/r02_kernel_choice.txt, whole-step times): k_tile is fastest up to ~1500 tiles,
// k_stream from 2048 to ~4096 (two waves per tile fill the chip sooner), k_edge from 8192 on (fewer instructions per edge once
// every SIMD has its two waves); where both apply, k_stream keeps the graphs below BT_EDGE_PREF_TILES.
bool edge_applies(const PlanDev &pd) {
    static const int off = std::getenv("BT_EDGE_OFF") ? std::atoi(std::getenv("BT_EDGE_OFF")) : 0;   // measurement only
    static const int pref = std::getenv("BT_EDGE_PREF_TILES") ? std::atoi(std::getenv("BT_EDGE_PREF_TILES")) : 6144;
    if (off || !pd.em_ok || pd.T < pd.em_min || pd.max_cams > 10 || pd.max_cams <= 0 || pd.max_tile_pairs > 64 || pd.max_tile_pairs <= 0)
        return false;
    return pd.T >= pref || !stream_applies(pd);
}

template <int MODE, int NT, int LGS, bool PROF = false>
static int launch_edge_t(const PlanDev &pd, const StepArgs &a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    const size_t lds = edge_lds_bytes(pd, MODE);
    static int per_cu = 0;
    static size_t per_cu_lds = 0;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return BT_EHIP;
        n_cu = prop.multiProcessorCount;
    }
    if (!per_cu || per_cu_lds != lds) {
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k_edge<MODE, NT, LGS, PROF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return BT_EHIP;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_edge<MODE, NT, LGS, PROF>, 64, lds) != hipSuccess || nb < 1) nb = 1;
        static const int cap = std::getenv("BT_EDGE_WAVES_PER_CU") ? std::atoi(std::getenv("BT_EDGE_WAVES_PER_CU")) : 0;   // measurement only
        if (cap > 0 && nb > cap) nb = cap;
        per_cu = nb; per_cu_lds = lds;
    }
    const int max_waves = n_cu * per_cu;
    const int tpw = (pd.T + max_waves - 1) / max_waves, nw = (pd.T + tpw - 1) / tpw;
    if (ev0) hipExtLaunchKernelGGL((k_edge<MODE, NT, LGS, PROF>), dim3(nw), dim3(64), lds, st, ev0, ev1, 0, pd, a, tpw);
    else hipLaunchKernelGGL((k_edge<MODE, NT, LGS, PROF>), dim3(nw), dim3(64), lds, st, pd, a, tpw);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

int launch_edge(const PlanDev &pd, const StepArgs &a, int mode, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    if ((unsigned long long)pd.e_all * (unsigned long long)a.tstride * 4ull >= (1ull << 32)) return BT_EUNSUPPORTED;   // 32-bit byte offsets into the targets
    if ((unsigned long long)pd.p_tot * (unsigned long long)a.mstride * 4ull >= (1ull << 32)) return BT_EUNSUPPORTED;
    const bool s8 = pd.em_lgs == 3;            // the 8-observation graphs of the benchmark generator
    if (mode == kEmSO) return s8 ? launch_edge_t<kEmSO, 1, 3>(pd, a, st, ev0, ev1) : launch_edge_t<kEmSO, 1, -1>(pd, a, st, ev0, ev1);
    if (mode == kEmUpd) return s8 ? launch_edge_t<kEmUpd, 1, 3>(pd, a, st, ev0, ev1) : launch_edge_t<kEmUpd, 1, -1>(pd, a, st, ev0, ev1);
    static const int e2 = std::getenv("BT_EDGE2") ? std::atoi(std::getenv("BT_EDGE2")) : 1;                 // measurement only
    if (e2 && !(a.dbg & 32)) return launch_edge2(pd, a, st, ev0, ev1);
    if (pd.max_cams <= 8 && s8 && (a.dbg & 32)) return launch_edge_t<kEmFull, 3, 3, true>(pd, a, st, ev0, ev1);
    if (pd.max_cams <= 8) return s8 ? launch_edge_t<kEmFull, 3, 3>(pd, a, st, ev0, ev1) : launch_edge_t<kEmFull, 3, -1>(pd, a, st, ev0, ev1);
    return launch_edge_t<kEmFull, 4, -1>(pd, a, st, ev0, ev1);
}

}  // namespace bt
