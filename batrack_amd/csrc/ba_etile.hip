// ba_etile.hip — k_etile: the Jacobian kernel of a tile in the PAIR-MAJOR layout (gfx950, wave64).
//
// One workgroup of 8 waves per tile of up to 64 tracks, like k_tile, but a wave's 64 lanes are 64 / S consecutive tracks x
// the S camera pairs of the tile (ba_plan.cpp "pair-major": lane = track_in_iteration * S + local pair), and the edges a
// track has with one pair — the caller appends a window's factors again every keyframe step, so repeated observations are
// the rule (batrack.py:399-410) — are the lane's D rounds.  What this removes from k_tile's slot loop:
//   * the wave-wide reduce-scatter of the 27 per-pair products (ba.py:260,266) per slot: a lane keeps ONE pair for the whole
//     tile, sums the products in registers over its rounds and iterations, and the lanes of a pair are added once per wave;
//   * every read-add-write of the local E: a lane owns its (track, target camera) element (one plain store of the sum over
//     its rounds), the track's source-camera row, C and w (ba.py:284-292) are a DPP reduction over the track's S lanes;
//   * the per-wave partials of (E source row, C, w), their barrier and the owner threads' merge: a track lives in one wave,
//     its Q = 1 / (C + prior + lambda) and w' (ba.py:296-311) are formed by the track's first lane right there.
// One barrier, then the tile's Schur product E Q E^T on v_mfma_f64_16x16x4_f64 exactly as in k_tile.  The per-edge maths is
// float64 on the float32 inputs by default (StepArgs::prec), float32 with BT_FORCE prec=f32.
// MODE kEtSO: structure-only steps, the whole step in this launch (the tracks' first lanes write the new disparities, the
// workgroups behind the tile workgroups do the rest of the buffer and the poses).  MODE kEtUpd: a step's last kernel — the
// depth back-substitution dZ = Q (w' - sum E^T dX) re-evaluated per edge (see k_update in ba_kernels.hip), the rest of the
// patch buffer, the pose retraction and the clearing of [S | y].
// Reference: ba.py:228-337, projective_ops.py:54-100.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>

#include "ba_edge.hpp"
#include "ba_kernels.hpp"
#include "dev_cache.hpp"
#include "probe.hpp"
#include "ba_update.hpp"

namespace bt {

enum { kEtFull = 0, kEtSO = 1, kEtUpd = 2 };
constexpr int kEtWaves = 8, kEtThreads = 64 * kEtWaves;
constexpr int kEtGeoUpd = 28;        // numbers per pair in LDS for kEtUpd: geometry (20), delta (6), padding

// ------------------------------------------------------------------ cross-lane sums, float and double
template <int CTRL, int BANK = 0xf, bool BOUND = true>
__device__ __forceinline__ float et_dpp(float old, float x) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(x), CTRL, 0xf, BANK, BOUND));
}
template <int CTRL, int BANK = 0xf, bool BOUND = true>
__device__ __forceinline__ double et_dpp(double old, double x) {
    const unsigned long long o = (unsigned long long)__double_as_longlong(old), v = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)v, CTRL, 0xf, BANK, BOUND);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(v >> 32), CTRL, 0xf, BANK, BOUND);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ float et_swap16(float x) {
    const uint2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float et_swap32(float x) {
    const uint2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ double et_mk(unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); }
__device__ __forceinline__ double et_swap16(double x) {
    const unsigned long long v = (unsigned long long)__double_as_longlong(x);
    const uint2_t rl = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    const uint2_t rh = __builtin_amdgcn_permlane16_swap((unsigned)(v >> 32), (unsigned)(v >> 32), false, false);
    return et_mk(rl.x, rh.x) + et_mk(rl.y, rh.y);
}
__device__ __forceinline__ double et_swap32(double x) {
    const unsigned long long v = (unsigned long long)__double_as_longlong(x);
    const uint2_t rl = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    const uint2_t rh = __builtin_amdgcn_permlane32_swap((unsigned)(v >> 32), (unsigned)(v >> 32), false, false);
    return et_mk(rl.x, rh.x) + et_mk(rl.y, rh.y);
}
// x + (x of lane ^ M)
template <int M, typename T>
__device__ __forceinline__ T et_xor_add(T x) {
    if (M == 1) return x + et_dpp<0xb1>((T)0, x);
    if (M == 2) return x + et_dpp<0x4e>((T)0, x);
    if (M == 4) {
        T t = et_dpp<0x104, 0x5, false>((T)0, x);             // row_shl:4 on the banks with bit 2 clear
        t = et_dpp<0x114, 0xa, false>(t, x);                   // row_shr:4 on the others
        return x + t;
    }
    if (M == 8) return x + et_dpp<0x128>((T)0, x);
    if (M == 16) return et_swap16(x);
    return et_swap32(x);
}
// sum of x over the groups of 2^lg adjacent lanes (every lane of a group gets the total); lg is wave-uniform.  Inside a row
// the partners are quad_perm (lane ^ 1, lane ^ 2), row_half_mirror (7 - lane of the 8) and row_mirror (15 - lane of the 16)
template <int N, typename T>
__device__ __forceinline__ void et_group_sum(T (&x)[N], int lg) {
    if (lg > 0) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] += et_dpp<0xb1>((T)0, x[i]); }
    if (lg > 1) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] += et_dpp<0x4e>((T)0, x[i]); }
    if (lg > 2) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] += et_dpp<0x141>((T)0, x[i]); }
    if (lg > 3) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] += et_dpp<0x140>((T)0, x[i]); }
    if (lg > 4) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] = et_swap16(x[i]); }
    if (lg > 5) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] = et_swap32(x[i]); }
}
// sum of x over the lanes with the same (lane mod 2^lg) (lane < 2^lg then holds the total of its residue class)
template <int N, typename T>
__device__ __forceinline__ void et_stride_sum(T (&x)[N], int lg) {
    if (lg <= 5) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] = et_xor_add<32>(x[i]); }
    if (lg <= 4) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] = et_xor_add<16>(x[i]); }
    if (lg <= 3) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] = et_xor_add<8>(x[i]); }
    if (lg <= 2) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] = et_xor_add<4>(x[i]); }
    if (lg <= 1) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] = et_xor_add<2>(x[i]); }
    if (lg <= 0) { _Pragma("unroll") for (int i = 0; i < N; ++i) x[i] = et_xor_add<1>(x[i]); }
}

// ------------------------------------------------------------------ k_etile
// Grid: kEtFull pd.T workgroups; kEtSO pd.T + the blocks of update_rest; kEtUpd tile_blocks + update_rest blocks + the
// blocks that clear [S | y] (as k_update).
template <int MODE, typename R, bool PROF = false, bool TWO = false>
__global__ __launch_bounds__(kEtThreads, TWO ? 1 : 2) void k_etile(PlanDev pd, StepArgs a, int do_poses, int tile_blocks, int first_zero_block) {
    if (MODE == kEtUpd && (int)blockIdx.x >= first_zero_block) {
        const size_t nz = (size_t)pd.D * pd.D + pd.D;
        const size_t i0 = ((size_t)(blockIdx.x - first_zero_block) * blockDim.x + threadIdx.x) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i0 + k < nz) a.S[i0 + k] = 0.0;
        return;
    }
    if (MODE != kEtFull && (int)blockIdx.x >= tile_blocks) {
        if (MODE == kEtSO) update_rest<true, true>(pd, a, ((int)blockIdx.x - tile_blocks) * (int)blockDim.x + (int)threadIdx.x, do_poses);
        else update_rest<false, true>(pd, a, ((int)blockIdx.x - tile_blocks) * (int)blockDim.x + (int)threadIdx.x, do_poses);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef typename Vec2<R>::type R2;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    BT_PROBE_ET_DECL();           // (measurement hooks: probe.hpp, tools/probes/wave_times.hpp)
    long long pf[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tc = PROF ? clock64() : 0, tn;
#define BT_PF(i) do { if (PROF) { __builtin_amdgcn_sched_barrier(0); tn = clock64(); pf[i] += tn - tc; tc = tn; __builtin_amdgcn_sched_barrier(0); } } while (0)
    constexpr int GS = MODE == kEtUpd ? kEtGeoUpd : kPairGeomFloats;
    const int mtp = pd.max_tile_pairs > 0 ? pd.max_tile_pairs : 1;
    const int R16max = MODE == kEtFull ? pd.max_rows16 : 0;
    // LDS: ppart [4][26][Smax] f64 (the waves' per-pair sums; LDS float64 atomics cost ~500 cycles per wave instruction, so
    // waves 0..3 store their sums side by side, waves 4..7 add theirs on top after a barrier, and the workgroup adds the four
    // slabs after the next one) | Eh [R16max][66] | Qs [128] (Q, then beta = Q w') | geo [mtp][GS] | gidx
    int Smax = 1;
    while (Smax < mtp) Smax <<= 1;
    double *ppart = reinterpret_cast<double *>(lds_raw);
    R *Eh = reinterpret_cast<R *>(ppart + (MODE == kEtFull ? (kEtWaves / 2) * 26 * Smax : 0));
    R *Qs = Eh + R16max * kLdsRowStride;
    R *geo = Qs + (MODE == kEtFull ? 128 : 0);
    int *gidx = reinterpret_cast<int *>(geo + (size_t)mtp * GS);

    // (an XCD's workgroups take a contiguous range of tiles, as in k_tile)
    const int nt_ = MODE == kEtFull ? pd.T : tile_blocks;
    const int tq_ = nt_ >> 3, tr_ = nt_ & 7, xcd_ = blockIdx.x & 7;
    const int tile = xcd_ * tq_ + min(xcd_, tr_) + ((int)blockIdx.x >> 3);

    const int4 rec = reinterpret_cast<const int4 *>(pd.pm_rec)[tile];
    const int round0 = rec.x, lgS = rec.y & 0xff, D = rec.y >> 8, nit = rec.z;
    const int S1 = (1 << lgS) - 1, G = kLanes >> lgS;
    const int ntrk = pd.tile_ntrk[tile], ncam = pd.tile_ncam[tile], np = pd.tile_npair[tile];
    const int Rw = 6 * ncam, R16 = MODE == kEtFull ? ((Rw + 15) >> 4) << 4 : 0;
    const int s = lane & S1, tl = lane >> lgS;
    const bool lead = s == 0;

    // ---- this wave's stream of rounds: k = 0 .. nr - 1  <->  (iteration wave + (k / D) * kEtWaves, round k % D)
    const int my_its = wave < nit ? (nit - wave + kEtWaves - 1) / kEtWaves : 0;
    const int nr = my_its * D;
    auto edge_of = [&](int k) -> int {
        if (k >= nr) return -1;
        const int it = wave + (k / D) * kEtWaves, d = k - (k / D) * D;
        return pd.pm_edge[(size_t)(round0 + it * D + d) * kLanes + lane];
    };
    auto gather = [&](int e, R &tu, R &tv, R &w0, R &w1) {
        tu = tv = w0 = w1 = (R)0;
        if (e >= 0) {
            const float *tp = a.targets + (size_t)e * a.tstride;
            tu = tp[0]; tv = tp[1];
            const float2 w = reinterpret_cast<const float2 *>(a.weights)[e];
            w0 = w.x; w1 = w.y;
        }
    };
    // first loads that need nothing but the tile index
    int e_cur = edge_of(0), e_nx = edge_of(1);
    const int ij0 = tid < np ? pd.tile_ij[(size_t)tile * mtp + tid] : 0;
    const unsigned lb = s < np ? pd.pm_lb[(size_t)tile * kLanes + s] : 0xffu;
    auto load_track = [&](int it, int &patch, unsigned &la, R &x, R &y, R &d, R &mono) {
        const int track = it * G + tl;
        patch = -1; la = 0xffu; x = y = d = mono = (R)0;
        if (it < nit && track < ntrk) {
            patch = pd.tile_kx[(size_t)tile * kLanes + track];
            la = pd.pm_la[(size_t)tile * kLanes + track];
            x = a.patches[3 * (size_t)patch]; y = a.patches[3 * (size_t)patch + 1]; d = a.patches[3 * (size_t)patch + 2];
            if (MODE != kEtUpd) mono = a.mono[(size_t)patch * a.mstride];
        }
    };
    int patch_c, patch_n;
    unsigned la_c, la_n;
    R px, py, pdisp, mono_v, px_n, py_n, pd_n, mono_n;
    load_track(wave, patch_c, la_c, px, py, pdisp, mono_v);

    // ---- prologue: rows of the reduced system, relative pose of the tile's camera pairs, clear accumulators
    if (MODE == kEtFull) {
        const int *cams = pd.tile_cams + pd.tile_cam0[tile];
        for (int i = tid; i < R16max; i += kEtThreads) gidx[i] = i < Rw ? 6 * cams[i / 6] + i % 6 : -1;
        for (int i = tid; i < R16 * kLdsRowStride; i += kEtThreads) Eh[i] = (R)0;
        if (tid < 128) Qs[tid] = (R)0;
    }
    for (int p = tid; p < np; p += kEtThreads) {
        R *g = geo + (size_t)p * GS;
        const int gp = pd.tile_pairs[pd.tile_pair0[tile] + p];
        if (MODE == kEtUpd) {
            // the geometry the Jacobian kernel left in the workspace, and delta = dX_j - Ad(Gij) dX_i of the pair
            const int ia = pd.pair_i[gp] - pd.fixedp, ib = pd.pair_j[gp] - pd.fixedp;
            R gg[kPairGeomFloats];
            const R2 *src = reinterpret_cast<const R2 *>(reinterpret_cast<const R *>(a.pairgeo) + (size_t)gp * kPairGeomFloats);
#pragma unroll
            for (int c = 0; c < kPairGeomFloats / 2; ++c) { const R2 t2 = src[c]; gg[2*c] = t2.x; gg[2*c + 1] = t2.y; }
            R xi[6] = {0, 0, 0, 0, 0, 0}, xj[6] = {0, 0, 0, 0, 0, 0};
            if (ia >= 0) for (int c = 0; c < 6; ++c) xi[c] = a.dx[6 * ia + c];
            if (ib >= 0) for (int c = 0; c < 6; ++c) xj[c] = a.dx[6 * ib + c];
            R Rt[3], Rp[3];                                      // Ad(Gij)(tau, phi) = (R tau + t x (R phi), R phi)   (se3.h:58-67)
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
            g[26] = (R)0; g[27] = (R)0;
        } else {
            const int ij = p == tid ? ij0 : pd.tile_ij[(size_t)tile * mtp + p];
            pair_geometry<R>(a.poses, a.intr, ij & 0xffff, ij >> 16, g);
            if (MODE == kEtFull) {                               // left for k_pair_finalize and the step's last kernel
                R2 *dst = reinterpret_cast<R2 *>(reinterpret_cast<R *>(a.pairgeo) + (size_t)gp * kPairGeomFloats);
                const R2 *src = reinterpret_cast<const R2 *>(g);
#pragma unroll
                for (int c = 0; c < kPairGeomFloats / 2; ++c) dst[c] = src[c];
            }
        }
    }
    R tu_c, tv_c, w0_c, w1_c;
    gather(e_cur, tu_c, tv_c, w0_c, w1_c);
    __syncthreads();
    BT_PF(0);
    BT_PROBE_ET_MARK(1);

    // this lane's pair: the same for the whole tile
    R g[GS];
    {
        const R *gs = geo + (size_t)(s < np ? s : 0) * GS;
#pragma unroll
        for (int c = 0; c < GS; ++c) g[c] = gs[c];
    }
    R pa[26];
#pragma unroll
    for (int i = 0; i < 26; ++i) pa[i] = (R)0;

    constexpr int NSV = MODE == kEtFull ? 8 : 2;                  // C, w (and the six numbers of the source-camera E row)
    R Ejacc[6] = {0, 0, 0, 0, 0, 0}, sv[NSV], dacc = 0;
#pragma unroll
    for (int c = 0; c < NSV; ++c) sv[c] = (R)0;
    bool any = false;
    int it = wave;
    if (nr > 0) load_track(wave + kEtWaves, patch_n, la_n, px_n, py_n, pd_n, mono_n);
    // ---- one edge of the lane's (track, pair): its share of the track's and of the pair's sums
    auto accum = [&](const EdgeQT<R> &q, bool act) {
        if (MODE == kEtUpd) {
            if (act) {
                const R d0 = q.a0 * g[20] + q.a2 * g[22] + q.a3 * g[23] + q.a4 * g[24] + q.a5 * g[25];
                const R d1 = q.b1 * g[21] + q.b2 * g[22] + q.b3 * g[23] + q.b4 * g[24] + q.b5 * g[25];
                dacc += q.W0 * q.jz0 * d0 + q.W1 * q.jz1 * d1;
            }
        } else {
            // C, w of the track (ba.py:287,292)
            sv[0] += q.W0 * q.jz0 * q.jz0 + q.W1 * q.jz1 * q.jz1;
            sv[1] += q.W0 * q.jz0 * q.r0 + q.W1 * q.jz1 * q.r1;
        }
        if constexpr (MODE == kEtFull) {
            const R wa0 = q.W0 * q.a0, wa2 = q.W0 * q.a2, wa3 = q.W0 * q.a3, wa4 = q.W0 * q.a4, wa5 = q.W0 * q.a5;
            const R wb1 = q.W1 * q.b1, wb2 = q.W1 * q.b2, wb3 = q.W1 * q.b3, wb4 = q.W1 * q.b4, wb5 = q.W1 * q.b5;
            // Ej = Jj^T W Jz (ba.py:263): summed over the lane's rounds; the source-camera row Ei = -Ad^T Ej is linear in it
            Ejacc[0] += wa0 * q.jz0; Ejacc[1] += wb1 * q.jz1;
            Ejacc[2] += fma_t(wa2, q.jz0, wb2 * q.jz1); Ejacc[3] += fma_t(wa3, q.jz0, wb3 * q.jz1);
            Ejacc[4] += fma_t(wa4, q.jz0, wb4 * q.jz1); Ejacc[5] += fma_t(wa5, q.jz0, wb5 * q.jz1);
            any |= act;
            // per-pair sums Bjj (21, row-major upper triangle) and gj (6) (ba.py:260,266), per lane
            pa[0] = fma_t(wa0, q.a0, pa[0]);   pa[1] = fma_t(wa0, q.a2, pa[1]);   pa[2] = fma_t(wa0, q.a3, pa[2]);
            pa[3] = fma_t(wa0, q.a4, pa[3]);   pa[4] = fma_t(wa0, q.a5, pa[4]);
            pa[5] = fma_t(wb1, q.b1, pa[5]);   pa[6] = fma_t(wb1, q.b2, pa[6]);   pa[7] = fma_t(wb1, q.b3, pa[7]);
            pa[8] = fma_t(wb1, q.b4, pa[8]);   pa[9] = fma_t(wb1, q.b5, pa[9]);
            pa[10] = fma_t(wa2, q.a2, fma_t(wb2, q.b2, pa[10])); pa[11] = fma_t(wa2, q.a3, fma_t(wb2, q.b3, pa[11]));
            pa[12] = fma_t(wa2, q.a4, fma_t(wb2, q.b4, pa[12])); pa[13] = fma_t(wa2, q.a5, fma_t(wb2, q.b5, pa[13]));
            pa[14] = fma_t(wa3, q.a3, fma_t(wb3, q.b3, pa[14])); pa[15] = fma_t(wa3, q.a4, fma_t(wb3, q.b4, pa[15]));
            pa[16] = fma_t(wa3, q.a5, fma_t(wb3, q.b5, pa[16]));
            pa[17] = fma_t(wa4, q.a4, fma_t(wb4, q.b4, pa[17])); pa[18] = fma_t(wa4, q.a5, fma_t(wb4, q.b5, pa[18]));
            pa[19] = fma_t(wa5, q.a5, fma_t(wb5, q.b5, pa[19]));
            pa[20] = fma_t(wa0, q.r0, pa[20]); pa[21] = fma_t(wb1, q.r1, pa[21]);
            pa[22] = fma_t(wa2, q.r0, fma_t(wb2, q.r1, pa[22])); pa[23] = fma_t(wa3, q.r0, fma_t(wb3, q.r1, pa[23]));
            pa[24] = fma_t(wa4, q.r0, fma_t(wb4, q.r1, pa[24])); pa[25] = fma_t(wa5, q.r0, fma_t(wb5, q.r1, pa[25]));
        }
    };
    // ---- the lane's rounds of an iteration are in: finish the iteration's tracks, take the next iteration's
    auto finish = [&](bool more) {
        const int track = it * G + tl;
        const bool has_trk = track < ntrk;
        if (MODE == kEtUpd) {
            R dv[1] = {dacc};
            et_group_sum(dv, lgS);
            if (lead && has_trk) {
                const R2 qw = reinterpret_cast<const R2 *>(a.qw)[pd.tile_trk0[tile] + track];
                float dd = (float)(pdisp + qw.x * (qw.y - dv[0]));                  // ba.py:328, :333
                dd = dd < 1e-3f ? 1e-3f : dd;
                dd = dd > 10.0f ? 10.0f : dd;
                a.patches_out[3 * (size_t)patch_c] = (float)px; a.patches_out[3 * (size_t)patch_c + 1] = (float)py; a.patches_out[3 * (size_t)patch_c + 2] = dd;
            }
            dacc = (R)0;
        } else {
            if constexpr (MODE == kEtFull) {
                if (any && la_c != 0xffu) {
                    // o_tau = R^T e_tau ; o_phi = R^T (e_tau x t + e_phi)      (se3.h:58-67)
                    const R cx = Ejacc[1]*g[11] - Ejacc[2]*g[10] + Ejacc[3];
                    const R cy = Ejacc[2]*g[9]  - Ejacc[0]*g[11] + Ejacc[4];
                    const R cz = Ejacc[0]*g[10] - Ejacc[1]*g[9]  + Ejacc[5];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        sv[2 + c] = -(g[c]*Ejacc[0] + g[3 + c]*Ejacc[1] + g[6 + c]*Ejacc[2]);
                        sv[5 + c] = -(g[c]*cx + g[3 + c]*cy + g[6 + c]*cz);
                    }
                }
                // this lane's (track, target camera) element of E: nobody else writes it
                if (any && lb != 0xffu) {
                    R *row = Eh + lb * 6 * kLdsRowStride + track;
#pragma unroll
                    for (int c = 0; c < 6; ++c) row[c * kLdsRowStride] = Ejacc[c];
                }
            }
            et_group_sum(sv, lgS);
            if (lead && has_trk) {                                                    // ba.py:296-311
                const int trk = pd.tile_trk0[tile] + track;
                const R pm = mono_v > (R)1e-2f ? (R)1 : (R)0;                         // (the prior is float32 data: compared as such)
                R Ca = sv[0] + pm * (R)a.alpha;
                Ca = Ca + (R)(a.lmbda_trk ? a.lmbda_trk[pd.trk_off + trk] : a.lmbda);
                const R wp = sv[1] - pm * (R)a.alpha * (pdisp - mono_v);
                const R Q = sizeof(R) == 8 ? (R)frcp((double)Ca) : (R)1 / Ca;      // (float64: seed + two Newton steps, < 1e-15; the IEEE divide is ~30 instructions)
                if constexpr (MODE == kEtSO) {                                        // ba.py:316-317, :333
                    R2 qw2; qw2.x = Q; qw2.y = wp;
                    reinterpret_cast<R2 *>(a.qw)[trk] = qw2;                          // (for a k_update<true> behind a split step)
                    float dd = (float)(pdisp + Q * wp);
                    dd = dd < 1e-3f ? 1e-3f : dd;
                    dd = dd > 10.0f ? 10.0f : dd;
                    a.patches_out[3 * (size_t)patch_c] = (float)px; a.patches_out[3 * (size_t)patch_c + 1] = (float)py; a.patches_out[3 * (size_t)patch_c + 2] = dd;
                } else if constexpr (MODE == kEtFull) {
                    R2 qw2; qw2.x = Q; qw2.y = wp;
                    reinterpret_cast<R2 *>(a.qw)[trk] = qw2;
                    Qs[track] = Q; Qs[64 + track] = Q * wp;
                    if (la_c != 0xffu) {
                        R *row = Eh + la_c * 6 * kLdsRowStride + track;
                        if (pd.em_self) {        // (behind this wave's stores above in LDS order: a self edge's target row is this one)
#pragma unroll
                            for (int c = 0; c < 6; ++c) row[c * kLdsRowStride] += sv[2 + c];
                        } else {                 // a track's source-camera row is written here and nowhere else
#pragma unroll
                            for (int c = 0; c < 6; ++c) row[c * kLdsRowStride] = sv[2 + c];
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NSV; ++c) sv[c] = (R)0;
#pragma unroll
            for (int c = 0; c < 6; ++c) Ejacc[c] = (R)0;
            any = false;
        }
        // next iteration's tracks (requested one iteration ago), and the request for the one after
        it += kEtWaves;
        patch_c = patch_n; la_c = la_n; px = px_n; py = py_n; pdisp = pd_n; mono_v = mono_n;
        if (more) load_track(it + kEtWaves, patch_n, la_n, px_n, py_n, pd_n, mono_n);
    };
    if constexpr (TWO) {
        // ---- two rounds of the lane per trip: the two edges' dependency chains (float64: two reciprocals by Newton steps,
        // the robust weight) run side by side in one wave — a tile of a sliding window is 8 waves x 3-4 rounds, its time is
        // the waves' latency.  Trip j = (iteration wave + (j / D2) * kEtWaves, rounds 2 h, 2 h + 1 with h = j mod D2).
        const int D2 = (D + 1) >> 1, ntrips = my_its * D2;
        auto edges_of = [&](int j, int &ea, int &eb) {
            ea = eb = -1;
            if (j >= ntrips) return;
            const int itx = j / D2, h = j - itx * D2, rd = round0 + (wave + itx * kEtWaves) * D + 2 * h;
            ea = pd.pm_edge[(size_t)rd * kLanes + lane];
            if (2 * h + 1 < D) eb = pd.pm_edge[(size_t)(rd + 1) * kLanes + lane];
        };
        int ea_c = e_cur, eb_c = D > 1 ? e_nx : -1, ea_n, eb_n;             // (trip 0 is rounds 0 and 1 of the first iteration)
        R tub_c, tvb_c, w0b_c, w1b_c;
        gather(eb_c, tub_c, tvb_c, w0b_c, w1b_c);
        edges_of(1, ea_n, eb_n);
        int h = 0;
#pragma unroll 1
        for (int j = 0; j < ntrips; ++j) {
            const int ea = ea_c, eb = eb_c;
            const R tua = tu_c, tva = tv_c, w0a = w0_c, w1a = w1_c, tub = tub_c, tvb = tvb_c, w0b = w0b_c, w1b = w1b_c;
            gather(ea_n, tu_c, tv_c, w0_c, w1_c);
            gather(eb_n, tub_c, tvb_c, w0b_c, w1b_c);
            ea_c = ea_n; eb_c = eb_n;
            edges_of(j + 2, ea_n, eb_n);
            EdgeQT<R> qa, qb;
            edge_eval<R>(g, px, py, pdisp, tua, tva, w0a, w1a, a, qa);
            edge_eval<R>(g, px, py, pdisp, tub, tvb, w0b, w1b, a, qb);
            if (ea < 0) { qa.W0 = (R)0; qa.W1 = (R)0; qa.r0 = (R)0; qa.r1 = (R)0; }
            if (eb < 0) { qb.W0 = (R)0; qb.W1 = (R)0; qb.r0 = (R)0; qb.r1 = (R)0; }
            accum(qa, ea >= 0);
            accum(qb, eb >= 0);
            if (++h < D2) continue;
            h = 0;
            finish(j + 1 < ntrips);
        }
    } else {
        int d = 0;
#pragma unroll 1
        for (int k = 0; k < nr; ++k) {
            // ---- pipeline: operands of round k were requested one round ago; request round k + 1's, and the edge ids of k + 2
            const int e = e_cur;
            const R tu = tu_c, tv = tv_c, w0 = w0_c, w1 = w1_c;
            R tu_x, tv_x, w0_x, w1_x;
            gather(e_nx, tu_x, tv_x, w0_x, w1_x);
            const int e_nn = edge_of(k + 2);
            const bool act = e >= 0;
            EdgeQT<R> q;
            edge_eval<R>(g, px, py, pdisp, tu, tv, w0, w1, a, q);
            if (!act) { q.W0 = (R)0; q.W1 = (R)0; q.r0 = (R)0; q.r1 = (R)0; }
            accum(q, act);
            // rotate the pipeline
            e_cur = e_nx; e_nx = e_nn;
            tu_c = tu_x; tv_c = tv_x; w0_c = w0_x; w1_c = w1_x;
            if (++d < D) continue;
            d = 0;
            finish(k + 1 < nr);
        }
    }
    BT_PF(1);
    BT_PROBE_ET_MARK(2);
    if (MODE != kEtFull) return;

    // ---- the lanes' pair sums: over the lanes with the same pair, then lane s (< S) adds them to the workgroup's float64 sums
    const int gp_l = (!pd.sp_ok && lane < np) ? pd.tile_pairs[pd.tile_pair0[tile] + lane] : 0;       // (np <= 64: one pair per lane)
    et_stride_sum(pa, lgS);
    BT_PF(6);
    {
        double *dst = ppart + (size_t)(wave & 3) * 26 * Smax + s;
        if (lane <= S1 && wave < 4) {
#pragma unroll
            for (int i = 0; i < 26; ++i) dst[i * Smax] = (double)pa[i];
        }
        __syncthreads();
        if (lane <= S1 && wave >= 4) {
#pragma unroll
            for (int i = 0; i < 26; ++i) dst[i * Smax] += (double)pa[i];
        }
    }
    __syncthreads();
    BT_PF(2);
    BT_PROBE_ET_MARK(3);

    // per-pair sums of the tile -> the workspace (k_pair_finalize turns them into B and v): atomics, or — sp_ok, where many
    // tiles share each pair — stored per tile for k_pair_finalize to add up
    const int nt = R16 >> 4, ntl = nt * (nt + 1) / 2;
    // (the tile's record in StepArgs::spart: products of the 16x16 tiles t = ti (ti + 1) / 2 + tj | E Q w' | pair sums [27][pairs]
    //  — at offsets fixed by the plan's largest tile, so that k_pair_finalize finds them without the tile's own sizes)
    const size_t sp_y = (size_t)(pd.max_rows16 >> 4) * ((pd.max_rows16 >> 4) + 1) / 2 * 256, sp_p = sp_y + pd.max_rows16;
    double *sp_t = a.spart + (size_t)tile * sp_tile_doubles(pd.max_rows16, pd.max_tile_pairs);
    // (element vi of all pairs by one wave: consecutive lanes read consecutive LDS words and store consecutive doubles —
    //  layout [vi][pair] inside the tile's record.  The atomics' pair index was loaded before the merge: a load inside this
    //  loop would wait, on the in-order counter, for the previous atomic's round trip as well)
    const int mtp_s = pd.max_tile_pairs > 0 ? pd.max_tile_pairs : 1;
    if (pd.sp_ok) {
        // the tile's E for the step's last kernel (k_etile_upd: dZ = Q (w' - E^T dX), ba.py:328, without the edges)
        double *es = a.esave + (((size_t)tile * pd.max_rows16) << pd.et_lgts);
        const int ts1 = (1 << pd.et_lgts) - 1;
        for (int idx = tid; idx < (Rw << pd.et_lgts); idx += kEtThreads)
            es[idx] = (double)Eh[(idx >> pd.et_lgts) * kLdsRowStride + (idx & ts1)];
    }
    double *spp = sp_t + sp_p + lane;
    double *pacc = a.pairacc + (size_t)gp_l * kPairAccStride;
    for (int vi = wave; vi < 27; vi += kEtWaves) {
        if (lane >= np) break;
        const int i = vi == 0 ? 0 : vi - 1;        // element order of the 27-vector: Bjj row-major upper triangle with its structural zero at [0][1]
        double val = 0.0;
        if (vi != 1) {
#pragma unroll
            for (int w = 0; w < kEtWaves / 2; ++w) val += ppart[((size_t)w * 26 + i) * Smax + lane];
        }
        if (pd.sp_ok) spp[(size_t)vi * mtp_s] = val;
        else if (val != 0.0) atomicAdd(&pacc[vi], val);
    }
    BT_PF(5);

    // ---- Schur product of the tile on the matrix cores (as k_tile): out[i][j] += sum_k Q_k Eh[i][k] Eh[j][k] over the 64
    // tracks, one 16x16 output tile per wave on v_mfma_f64_16x16x4_f64; the diagonal tiles' waves also emit E (Q w')
    for (int t = wave; t < ntl; t += kEtWaves) {
        int ti = 0, base = 0;
        while (base + ti + 1 <= t) { base += ti + 1; ++ti; }
        const int tj = t - base;
        const int li = lane & 15, kq = lane >> 4;
        const R *ar = Eh + (16 * ti + li) * kLdsRowStride + kq;
        const R *br = Eh + (16 * tj + li) * kLdsRowStride + kq;
        const R *qr = Qs + kq;
        // (a tile of 16 tracks needs 4 of the 16 k-steps: the columns behind its tracks are zero)
        const int nks = (ntrk + 3) >> 2;
        R av[16], bv[16], qv[16];
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 4)
            if (k0 < nks) {
#pragma unroll
                for (int ks = k0; ks < k0 + 4; ++ks) { av[ks] = ar[4 * ks]; bv[ks] = br[4 * ks]; qv[ks] = qr[4 * ks]; }
            } else {
#pragma unroll
                for (int ks = k0; ks < k0 + 4; ++ks) { av[ks] = (R)0; bv[ks] = (R)0; qv[ks] = (R)0; }
            }
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 4)
            if (k0 < nks) {
#pragma unroll
                for (int ks = k0; ks < k0 + 4; ++ks)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[ks] * (double)qv[ks], (double)bv[ks], acc, 0, 0, 0);
            }
        // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
        // sp_ok (all tiles share their cameras): the tile's product is stored, k_pair_finalize adds the tiles' products up —
        // hundreds of workgroups' float64 atomics on the same few thousand elements of S would serialise in the L2
        if (pd.sp_ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sp_t[(size_t)t * 256 + r * 64 + lane] = acc[r];
        } else {
            const int gc = gidx[16 * tj + li];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + kq + 4 * r;
                if (gc >= 0 && row < Rw) { const int gr = gidx[row]; if (gr >= gc) atomicAdd(&a.S[(size_t)gr * pd.D + gc], -acc[r]); }
            }
        }
        if (ti == tj) {
            double4_t yt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += 4)
                if (k0 < nks) {
#pragma unroll
                    for (int ks = k0; ks < k0 + 4; ++ks)
                        yt = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[ks], (double)qr[64 + 4 * ks], yt, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + kq + 4 * r;
                if (pd.sp_ok) { if (li == 0) sp_t[sp_y + row] = yt[r]; }
                else if (li == 0 && row < Rw) atomicAdd(&a.y[gidx[row]], -yt[r]);
            }
        }
    }
    BT_PF(3);
    BT_PROBE_ET_END(MODE == kEtFull, lane, wave, ((long long)ntrk << 32) | (long long)(D << 8 | nit));
    if (PROF && lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BT_PF(4);
        long long *o = reinterpret_cast<long long *>(a.status + 4) + (blockIdx.x == 0 ? 20 : 30);
        for (int i = 0; i < 10; ++i) o[i] = pf[i];
    }
#undef BT_PF
}

// ------------------------------------------------------------------ k_etile_upd
// Last kernel of a pose+structure step on plans with sp_ok: one wave per tile reads the E its k_etile<kEtFull> left
// (StepArgs::esave) and applies dZ = Q (w' - E^T dX) (ba.py:328, :333) — 6 ncam x ntrk products per tile instead of the
// tile's edges again.  Lanes = (row slice, track); behind the tile blocks the blocks of update_rest and of the clearing
// of [S | y], as k_update.
constexpr int kEuThreads = 256, kEuRows = 6 * kTileCamHard;
__global__ __launch_bounds__(kEuThreads) void k_etile_upd(PlanDev pd, StepArgs a, int do_poses, int tile_blocks, int first_zero_block) {
    if ((int)blockIdx.x >= first_zero_block) {
        const size_t nz = (size_t)pd.D * pd.D + pd.D;
        const size_t i0 = ((size_t)(blockIdx.x - first_zero_block) * blockDim.x + threadIdx.x) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i0 + k < nz) a.S[i0 + k] = 0.0;
        return;
    }
    if ((int)blockIdx.x >= tile_blocks) {
        update_rest<false, true>(pd, a, ((int)blockIdx.x - tile_blocks) * (int)blockDim.x + (int)threadIdx.x, do_poses);
        return;
    }
    __shared__ float sdx[kEuThreads / 64][kEuRows];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = (int)blockIdx.x * (kEuThreads / 64) + wave;
    if (tile >= pd.T) return;
    const int ntrk = pd.tile_ntrk[tile], Rw = 6 * pd.tile_ncam[tile];
    const int lgts = pd.et_lgts, k = lane & ((1 << lgts) - 1), sl = lane >> lgts, nsl = 64 >> lgts;
    // the track's numbers first: their latency runs under the rows
    int patch = -1;
    float px = 0.f, py = 0.f, pdisp = 0.f;
    double2 qw = {0.0, 0.0};
    if (sl == 0 && k < ntrk) {
        patch = pd.tile_kx[(size_t)tile * kLanes + k];
        px = a.patches[3 * (size_t)patch]; py = a.patches[3 * (size_t)patch + 1]; pdisp = a.patches[3 * (size_t)patch + 2];
        qw = reinterpret_cast<const double2 *>(a.qw)[pd.tile_trk0[tile] + k];
    }
    const int *cams = pd.tile_cams + pd.tile_cam0[tile];
    for (int r = lane; r < Rw; r += 64) sdx[wave][r] = a.dx[6 * cams[r / 6] + r % 6];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const double *E = a.esave + (((size_t)tile * pd.max_rows16) << lgts) + k;
    double acc = 0.0;
    for (int r0 = sl; r0 < Rw; r0 += 8 * nsl) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int r = r0 + u * nsl; v[u] = r < Rw ? E[(size_t)r << lgts] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int r = r0 + u * nsl; if (r < Rw) acc = fma(v[u], (double)sdx[wave][r], acc); }
    }
    // over the row slices: lanes with the same track are 1 << lgts apart
    for (int m = 32; m >= (1 << lgts); m >>= 1) acc += __shfl_xor(acc, m);
    if (patch >= 0) {
        float dd = (float)((double)pdisp + qw.x * (qw.y - acc));                  // ba.py:328, :333
        dd = dd < 1e-3f ? 1e-3f : dd;
        dd = dd > 10.0f ? 10.0f : dd;
        a.patches_out[3 * (size_t)patch] = px; a.patches_out[3 * (size_t)patch + 1] = py; a.patches_out[3 * (size_t)patch + 2] = dd;
    }
}

// ------------------------------------------------------------------ dispatch
static size_t etile_lds_bytes(const PlanDev &pd, int mode, size_t rsz) {
    const size_t mtp = (size_t)(pd.max_tile_pairs > 0 ? pd.max_tile_pairs : 1);
    if (mode == kEtUpd) return mtp * kEtGeoUpd * rsz + 64;
    if (mode == kEtSO) return mtp * kPairGeomFloats * rsz + 64;
    return etile_full_lds_bytes(pd.max_rows16, pd.max_tile_pairs, rsz);
}

constexpr size_t kEtLdsBudget = kEtileLdsBudget;

// 8: the tiles' E fits LDS as double, 4: only as float, 0: k_etile does not take this plan
int etile_precision_bytes(const PlanDev &pd) {
    if (pd.pm_ok != 2 || pd.T <= 0 || edge_applies(pd) || stream_applies(pd)) return 0;
    if (etile_lds_bytes(pd, kEtFull, sizeof(double)) <= kEtLdsBudget) return 8;
    return 0;                       // (k_tile then decides: float64 if ITS tile fits LDS as double, else float32)
}

template <int MODE, typename R, bool PROF = false, bool TWO = false>
static int launch_etile_t(const PlanDev &pd, const StepArgs &a, int do_poses, int extra_blocks, int zero_blocks, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    const size_t lds = etile_lds_bytes(pd, MODE, sizeof(R));
    static LdsLimit lds_limit;                     // per instantiation and device; only ever raised (several plans coexist)
    if (!lds_limit.ensure(reinterpret_cast<const void *>(&k_etile<MODE, R, PROF, TWO>), lds, pd.dev_id)) return BT_EHIP;
    const dim3 grid((unsigned)(pd.T + extra_blocks + zero_blocks)), blk(kEtThreads);
    if (ev0) hipExtLaunchKernelGGL((k_etile<MODE, R, PROF, TWO>), grid, blk, lds, st, ev0, ev1, 0, pd, a, do_poses, pd.T, pd.T + extra_blocks);
    else hipLaunchKernelGGL((k_etile<MODE, R, PROF, TWO>), grid, blk, lds, st, pd, a, do_poses, pd.T, pd.T + extra_blocks);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

// mode 0: pose+structure reduce; 1: the whole structure-only step (do_poses: copy the poses too); 2: a pose+structure step's
// last kernel.  extra_blocks / zero_blocks: the blocks of update_rest / of the clearing of [S | y] behind the tile blocks.
int launch_etile(const PlanDev &pd, const StepArgs &a, int mode, int do_poses, int extra_blocks, int zero_blocks, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    const bool dbl = a.prec != 0;
    if (mode == kEtUpd && dbl && pd.sp_ok && 6 * pd.max_cams <= kEuRows) {
        // (extra_blocks / zero_blocks were counted in blocks of kEtThreads threads)
        const int per = kEtThreads / kEuThreads, tb = (pd.T + kEuThreads / 64 - 1) / (kEuThreads / 64);
        const dim3 grid((unsigned)(tb + per * (extra_blocks + zero_blocks))), blk(kEuThreads);
        if (ev0) hipExtLaunchKernelGGL(k_etile_upd, grid, blk, 0, st, ev0, ev1, 0, pd, a, do_poses, tb, tb + per * extra_blocks);
        else hipLaunchKernelGGL(k_etile_upd, grid, blk, 0, st, pd, a, do_poses, tb, tb + per * extra_blocks);
        return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
    }
    if (mode == kEtSO) return dbl ? launch_etile_t<kEtSO, double>(pd, a, do_poses, extra_blocks, 0, st, ev0, ev1) : launch_etile_t<kEtSO, float>(pd, a, do_poses, extra_blocks, 0, st, ev0, ev1);
    if (mode == kEtUpd) return dbl ? launch_etile_t<kEtUpd, double>(pd, a, do_poses, extra_blocks, zero_blocks, st, ev0, ev1) : launch_etile_t<kEtUpd, float>(pd, a, do_poses, extra_blocks, zero_blocks, st, ev0, ev1);
    // float64: two rounds per trip (both variants need more than 128 registers: one workgroup per CU either way)
    constexpr bool two = true;
    if ((a.dbg & 32) && dbl && two) return launch_etile_t<kEtFull, double, true, true>(pd, a, 0, 0, 0, st, ev0, ev1);
    if (a.dbg & 32) return dbl ? launch_etile_t<kEtFull, double, true>(pd, a, 0, 0, 0, st, ev0, ev1) : launch_etile_t<kEtFull, float, true>(pd, a, 0, 0, 0, st, ev0, ev1);
    if (dbl && two) return launch_etile_t<kEtFull, double, false, true>(pd, a, 0, 0, 0, st, ev0, ev1);
    return dbl ? launch_etile_t<kEtFull, double>(pd, a, 0, 0, 0, st, ev0, ev1) : launch_etile_t<kEtFull, float>(pd, a, 0, 0, 0, st, ev0, ev1);
}

}  // namespace bt
