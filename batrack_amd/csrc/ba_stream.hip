// ba_stream.hip — wave-per-tile streaming kernels for graphs of many tiles (gfx950, wave64).
//
// The tile kernel of ba_kernels.hip spreads one tile of 64 tracks over a workgroup of 8-16 waves: right for
// graphs of a few hundred tiles, where a tile's latency is all there is.  On graphs of thousands of tiles the
// chain  tile tables -> poses / patches -> edge list -> targets -> barriers -> Schur product -> atomics  of ONE
// tile at a time keeps a CU's pipes idle.  Here every WAVE owns whole tiles and walks a contiguous range of them
// with no workgroup barrier anywhere (workgroup = one wave, its LDS is private):
//   * lane l = track l of the tile (as in k_tile); the wave runs ALL edge slots of the tile, in groups of kSG
//     slots whose operands (edge ids, 16-bit slot codes, targets, weights) are loaded in one burst — the first
//     group of tile t+1 while tile t is still being computed (tables at its top, the gathers before its Schur
//     product), so the tile-to-tile critical path holds no exposed memory round trip;
//   * local E in LDS (lane-private columns: plain read-add-write), per-pair sums in LDS doubles;
//   * the Schur product E Q E^T on v_mfma_f64_16x16x4_f64 with the accumulators kept in REGISTERS across
//     consecutive tiles with the same cameras (tracks of one frame), atomics only when the cameras change;
//     E Q w' by a wave reduce-scatter (the MFMA form would spend three 16x16 output tiles on one row);
//   * MODE kModeSO: structure-only steps (C, w, Q, w' only);  kModeUpd: the depth back-substitution of k_update
//     (ba_kernels.hip) for the same tile walk.
// Reference: ba.py:228-337, projective_ops.py:54-100.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>

#include "ba_edge.hpp"
#include "ba_kernels.hpp"

namespace bt {

constexpr int kSG = 8;                 // slots per operand group
enum { kModeFull = 0, kModeSO = 1, kModeUpd = 2 };
constexpr int kUpdGeoS = 28;           // floats per pair in LDS for kModeUpd: geometry (20), delta (6), padding

struct TileRec { int ntrk, ncam, npair, flags, slot0, nslot, cam0, pair0, trk0; };

__device__ __forceinline__ TileRec load_rec(const PlanDev &pd, int t) {
    const int4 r0 = reinterpret_cast<const int4 *>(pd.tile_rec)[2 * t], r1 = reinterpret_cast<const int4 *>(pd.tile_rec)[2 * t + 1];
    TileRec r;
    r.ntrk = r0.x & 0xff; r.ncam = (r0.x >> 8) & 0xff; r.npair = (r0.x >> 16) & 0xff; r.flags = (r0.x >> 24) & 0xff;
    r.slot0 = r0.y; r.nslot = r0.z; r.cam0 = r0.w; r.pair0 = r1.x; r.trk0 = r1.y;
    return r;
}

struct Grp { unsigned code[kSG]; float tu[kSG], tv[kSG], w0[kSG], w1[kSG]; };

__device__ __forceinline__ void load_tables(const PlanDev &pd, int slot_base, int nvalid, int lane, int (&e)[kSG], unsigned (&code)[kSG]) {
#pragma unroll
    for (int s = 0; s < kSG; ++s) {
        e[s] = -1; code[s] = 0xffu;
        if (s < nvalid) {
            const size_t idx = (size_t)(slot_base + s) * kLanes + lane;
            e[s] = pd.slot_edge[idx]; code[s] = pd.slot_code[idx];
        }
    }
}

// bit 16 of the code = the lane has an edge in this slot; inactive lanes read pair 0 with zero weights
__device__ __forceinline__ void load_gather(const StepArgs &a, const int (&e)[kSG], const unsigned (&code)[kSG], Grp &g) {
#pragma unroll
    for (int s = 0; s < kSG; ++s) {
        const bool act = e[s] >= 0;
        g.code[s] = act ? (code[s] | 0x10000u) : 0xffu;
        g.tu[s] = g.tv[s] = g.w0[s] = g.w1[s] = 0.0f;
        if (act) {
            const float *tp = a.targets + (size_t)e[s] * a.tstride;
            g.tu[s] = tp[0]; g.tv[s] = tp[1];
            const float2 w = reinterpret_cast<const float2 *>(a.weights)[e[s]];
            g.w0[s] = w.x; g.w1[s] = w.y;
        }
    }
}

template <int NT>
__device__ __forceinline__ void schur_acc(const float *Eh, const float *Qs, int R, int lane, double4_t (&acc)[NT * (NT + 1) / 2]) {
    const int kq = lane >> 4, li = lane & 15;
    float qv[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) qv[ks] = Qs[4 * ks + kq];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        if (16 * ti < R) {
            const int ra = min(16 * ti + li, R);            // rows beyond the tile's E read the w' row: their outputs are never emitted
            double av[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) av[ks] = (double)Eh[ra * kLdsRowStride + 4 * ks + kq] * (double)qv[ks];
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj) {
                const int rb = min(16 * tj + li, R);
                float bv[16];
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) bv[ks] = Eh[rb * kLdsRowStride + 4 * ks + kq];
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
                    acc[ti * (ti + 1) / 2 + tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], (double)bv[ks], acc[ti * (ti + 1) / 2 + tj], 0, 0, 0);
            }
        }
    }
}

// MODE, NT: row tiles (16 rows each) of the largest local E the instantiation accumulates (kModeFull only)
template <int MODE, int NT>
__global__ __launch_bounds__(64, MODE == kModeFull ? 2 : 3) void k_stream(PlanDev pd, StepArgs a, int tiles_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const int mtp = pd.max_tile_pairs > 0 ? pd.max_tile_pairs : 1;
    const int Rmax = 6 * pd.max_cams;
    // LDS carve-up (wave-private)
    float *geo = lds;                                                  // [mtp][20] (kModeUpd: [mtp][28])
    float *Eh = geo + mtp * (MODE == kModeUpd ? kUpdGeoS : kPairGeomFloats);   // [(Rmax + 1)][66], row Rmax... row R of the tile = w'
    float *Qs = Eh + (MODE == kModeFull ? (Rmax + 1) * kLdsRowStride : 0);
    int *gidx = reinterpret_cast<int *>(Qs + (MODE == kModeFull ? 64 : 0));
    int *gpl = gidx + (MODE == kModeFull ? ((Rmax + 3) & ~3) : 0);
    double *pacc = reinterpret_cast<double *>(gpl + (MODE == kModeFull ? ((mtp + 3) & ~3) : 0));   // [mtp][32]
    const int GS = MODE == kModeUpd ? kUpdGeoS : kPairGeomFloats;

    const int t_begin = blockIdx.x * tiles_per_wave, t_end = min(pd.T, t_begin + tiles_per_wave);
    if (t_begin >= t_end) return;

    constexpr int NACC = NT * (NT + 1) / 2;
    double4_t sacc[NACC];
    double yacc[2] = {0.0, 0.0};
#pragma unroll
    for (int t = 0; t < NACC; ++t) sacc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
    bool acc_live = false;
    int Racc = 0, np_acc = 0;

    auto flush_schur = [&]() {
        if (MODE != kModeFull || !acc_live) return;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj) {
                const int t = ti * (ti + 1) / 2 + tj;
                const int col = 16 * tj + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + (lane >> 4) + 4 * r;
                    const double val = sacc[t][r];
                    sacc[t][r] = 0.0;
                    if (row < Racc && col < Racc) {
                        const int gr = gidx[row], gc = gidx[col];
                        if (gr >= gc) atomicAdd(&a.S[(size_t)gr * pd.D + gc], -val);
                    }
                }
            }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int row = 32 * c + ((lane >> 1) & 31);
            if (!(lane & 1) && row < Racc) atomicAdd(&a.y[gidx[row]], -yacc[c]);
            yacc[c] = 0.0;
        }
        acc_live = false;
    };
    auto flush_pairs = [&]() {
        if (MODE != kModeFull) return;
        for (int p0 = 0; p0 < np_acc; p0 += 2) {
            const int p = p0 + (lane >> 5), vi = lane & 31;
            if (p < np_acc && vi < 27) {
                double *src = pacc + p * 32 + vi;
                atomicAdd(&a.pairacc[(size_t)gpl[p] * kPairAccStride + vi], *src);
                *src = 0.0;
            }
        }
        np_acc = 0;
    };
    if (MODE == kModeFull) {
        for (int i = lane; i < mtp * 32; i += 64) pacc[i] = 0.0;
    }

    // ---- operands of the first tile
    TileRec rec = load_rec(pd, t_begin);
    int kx_c = pd.tile_kx[(size_t)t_begin * kLanes + lane];
    unsigned la_c = pd.tile_la[(size_t)t_begin * kLanes + lane];
    Grp grp;
    {
        int e[kSG]; unsigned code[kSG];
        load_tables(pd, rec.slot0, min(rec.nslot, kSG), lane, e, code);
        load_gather(a, e, code, grp);
    }
    float px = 0.0f, py = 0.0f, pdisp = 0.0f, mono_v = 0.0f;
    if (kx_c >= 0) {
        px = a.patches[3 * kx_c]; py = a.patches[3 * kx_c + 1]; pdisp = a.patches[3 * kx_c + 2];
        if (MODE != kModeUpd) mono_v = a.mono[kx_c];
    }

#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int flags = tile == t_begin ? 0 : rec.flags;
        const int R = 6 * rec.ncam;
        const bool has_next = tile + 1 < t_end;
        // ---- new cameras / new pair list
        if (MODE == kModeFull && !(flags & 1)) {
            flush_schur();
            const int *cams = pd.tile_cams + rec.cam0;
            for (int i = lane; i < R; i += 64) gidx[i] = 6 * cams[i / 6] + i % 6;
        }
        if (!(flags & 2)) {
            flush_pairs();
            for (int p = lane; p < rec.npair; p += 64) {
                const int gp = pd.tile_pairs[rec.pair0 + p];
                float *g = geo + p * GS;
                if (MODE == kModeUpd) {
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
                    const int ij = pd.tile_ij[(size_t)tile * mtp + p];
                    pair_geometry(a.poses, a.intr, ij & 0xffff, ij >> 16, g);
                    if (MODE == kModeFull) {
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
        if (MODE == kModeFull) {
            for (int r = 0; r < R; ++r) Eh[r * kLdsRowStride + lane] = 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- the tile's slots
        float Cacc = 0.0f, wacc = 0.0f, Ei[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, dacc = 0.0f;
        const unsigned la = la_c;
#pragma unroll 1
        for (int sb = 0; sb < rec.nslot; sb += kSG) {
            if (sb > 0) {                        // (tiles of more than kSG slots: later groups are loaded in place)
                int e[kSG]; unsigned code[kSG];
                load_tables(pd, rec.slot0 + sb, min(rec.nslot - sb, kSG), lane, e, code);
                load_gather(a, e, code, grp);
            }
            const int ns = min(rec.nslot - sb, kSG);
#pragma unroll
            for (int s = 0; s < kSG; ++s) {
                if (s >= ns) break;
                __builtin_amdgcn_sched_barrier(0);      // one slot's working set at a time (the unrolled bodies would otherwise be interleaved)
                const unsigned code = grp.code[s];
                const bool act = (code >> 16) != 0;
                const unsigned lb = code & 0xffu, lp = (code >> 8) & 0xffu;
                float g[MODE == kModeUpd ? kUpdGeoS : kPairGeomFloats];
                {
                    const float4 *g4 = reinterpret_cast<const float4 *>(geo + (size_t)lp * GS);
#pragma unroll
                    for (int c = 0; c < (MODE == kModeUpd ? kUpdGeoS : kPairGeomFloats) / 4; ++c) {
                        const float4 t4 = g4[c];
                        g[4*c] = t4.x; g[4*c + 1] = t4.y; g[4*c + 2] = t4.z; g[4*c + 3] = t4.w;
                    }
                }
                EdgeQ q;
                edge_eval(g, px, py, pdisp, grp.tu[s], grp.tv[s], grp.w0[s], grp.w1[s], a, q);
                if (!act) { q.W0 = 0.0f; q.W1 = 0.0f; q.r0 = 0.0f; q.r1 = 0.0f; }
                if (MODE == kModeUpd) {
                    if (act) {
                        const float d0 = q.a0 * g[20] + q.a2 * g[22] + q.a3 * g[23] + q.a4 * g[24] + q.a5 * g[25];
                        const float d1 = q.b1 * g[21] + q.b2 * g[22] + q.b3 * g[23] + q.b4 * g[24] + q.b5 * g[25];
                        dacc += q.W0 * q.jz0 * d0 + q.W1 * q.jz1 * d1;
                    }
                    continue;
                }
                // C, w of the track (ba.py:287,292)
                Cacc += q.W0 * q.jz0 * q.jz0 + q.W1 * q.jz1 * q.jz1;
                wacc += q.W0 * q.jz0 * q.r0 + q.W1 * q.jz1 * q.r1;
                if (MODE == kModeSO) continue;

                const float wa0 = q.W0 * q.a0, wa2 = q.W0 * q.a2, wa3 = q.W0 * q.a3, wa4 = q.W0 * q.a4, wa5 = q.W0 * q.a5;
                const float wb1 = q.W1 * q.b1, wb2 = q.W1 * q.b2, wb3 = q.W1 * q.b3, wb4 = q.W1 * q.b4, wb5 = q.W1 * q.b5;
                // Ej = Jj^T W Jz (ba.py:263) and Ei = -Ad^T Ej
                const float Ej[6] = { wa0 * q.jz0, wb1 * q.jz1, fmaf(wa2, q.jz0, wb2 * q.jz1), fmaf(wa3, q.jz0, wb3 * q.jz1),
                                      fmaf(wa4, q.jz0, wb4 * q.jz1), fmaf(wa5, q.jz0, wb5 * q.jz1) };
                if (act && lb != 0xffu) {
                    float *row = Eh + lb * 6 * kLdsRowStride + lane;
                    float old[6];
#pragma unroll
                    for (int c = 0; c < 6; ++c) old[c] = row[c * kLdsRowStride];
#pragma unroll
                    for (int c = 0; c < 6; ++c) row[c * kLdsRowStride] = old[c] + Ej[c];
                }
                if (act && la != 0xffu) {
                    // o_tau = R^T e_tau ; o_phi = R^T (e_tau x t + e_phi)      (se3.h:58-67)
                    const float cx = Ej[1]*g[11] - Ej[2]*g[10] + Ej[3];
                    const float cy = Ej[2]*g[9]  - Ej[0]*g[11] + Ej[4];
                    const float cz = Ej[0]*g[10] - Ej[1]*g[9]  + Ej[5];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        Ei[c]     -= g[c]*Ej[0] + g[3 + c]*Ej[1] + g[6 + c]*Ej[2];
                        Ei[3 + c] -= g[c]*cx + g[3 + c]*cy + g[6 + c]*cz;
                    }
                }
                // per-pair sums: Bjj (21, row-major upper triangle) and gj (6)   (ba.py:260,266); one pass per
                // distinct pair of the slot (a single pass on regular graphs)
                unsigned long long todo = __ballot(act);
                while (todo) {
                    const int leader = __ffsll((long long)todo) - 1;
                    const unsigned p0 = __shfl(lp, leader);
                    const float m = (act && lp == p0) ? 1.0f : 0.0f;
                    const float ma0 = m * wa0, mb1 = m * wb1, ma2 = m * wa2, mb2 = m * wb2, ma3 = m * wa3, mb3 = m * wb3,
                                ma4 = m * wa4, mb4 = m * wb4, ma5 = m * wa5, mb5 = m * wb5;
                    float v[32];
                    v[0] = ma0 * q.a0;  v[1] = 0.0f;        v[2] = ma0 * q.a2;  v[3] = ma0 * q.a3;
                    v[4] = ma0 * q.a4;  v[5] = ma0 * q.a5;
                    v[6] = mb1 * q.b1;  v[7] = mb1 * q.b2;  v[8] = mb1 * q.b3;  v[9] = mb1 * q.b4;  v[10] = mb1 * q.b5;
                    v[11] = fmaf(ma2, q.a2, mb2 * q.b2); v[12] = fmaf(ma2, q.a3, mb2 * q.b3);
                    v[13] = fmaf(ma2, q.a4, mb2 * q.b4); v[14] = fmaf(ma2, q.a5, mb2 * q.b5);
                    v[15] = fmaf(ma3, q.a3, mb3 * q.b3); v[16] = fmaf(ma3, q.a4, mb3 * q.b4); v[17] = fmaf(ma3, q.a5, mb3 * q.b5);
                    v[18] = fmaf(ma4, q.a4, mb4 * q.b4); v[19] = fmaf(ma4, q.a5, mb4 * q.b5);
                    v[20] = fmaf(ma5, q.a5, mb5 * q.b5);
                    v[21] = ma0 * q.r0; v[22] = mb1 * q.r1;
                    v[23] = fmaf(ma2, q.r0, mb2 * q.r1); v[24] = fmaf(ma3, q.r0, mb3 * q.r1);
                    v[25] = fmaf(ma4, q.r0, mb4 * q.r1); v[26] = fmaf(ma5, q.r0, mb5 * q.r1);
                    v[27] = v[28] = v[29] = v[30] = v[31] = 0.0f;
                    wave_reduce_scatter32(v, lane);
                    const int vi = (lane >> 1) & 31;
                    if (!(lane & 1) && vi < 27) pacc[p0 * 32 + vi] += (double)v[0];
                    todo &= ~__ballot(act && lp == p0);
                }
            }
        }

        // ---- next tile, stage A: its record, patch index and slot tables are requested now and arrive under the tile's
        // finish; stage B (the gathers through them) is issued before the Schur product and lands under it
        TileRec rec_n = rec;
        int kx_n = -1;
        unsigned la_n = 0xffu;
        int e_n[kSG];
        unsigned code_n[kSG];
        if (has_next) {
            rec_n = load_rec(pd, tile + 1);
            kx_n = pd.tile_kx[(size_t)(tile + 1) * kLanes + lane];
            la_n = pd.tile_la[(size_t)(tile + 1) * kLanes + lane];
            load_tables(pd, rec_n.slot0, min(rec_n.nslot, kSG), lane, e_n, code_n);
        }
        Grp grp_n;
        float px_n = 0.0f, py_n = 0.0f, pd_n = 0.0f, mono_n = 0.0f;
        auto stage_b = [&]() {
            if (has_next) {
                load_gather(a, e_n, code_n, grp_n);
                if (kx_n >= 0) {
                    px_n = a.patches[3 * kx_n]; py_n = a.patches[3 * kx_n + 1]; pd_n = a.patches[3 * kx_n + 2];
                    if (MODE != kModeUpd) mono_n = a.mono[kx_n];
                }
            }
        };

        const bool has_trk = lane < rec.ntrk;
        if (MODE == kModeUpd) {
            if (has_trk) {
                const float2 qw = a.qw[rec.trk0 + lane];
                float dd = pdisp + qw.x * (qw.y - dacc);                         // ba.py:328, :333
                dd = dd < 1e-3f ? 1e-3f : dd;
                dd = dd > 10.0f ? 10.0f : dd;
                a.patches_out[3 * kx_c] = px; a.patches_out[3 * kx_c + 1] = py; a.patches_out[3 * kx_c + 2] = dd;
            }
            stage_b();
        } else {
            float Q = 0.0f, wp = 0.0f;                                            // ba.py:296-311
            if (has_trk) {
                const float pm = mono_v > 1e-2f ? 1.0f : 0.0f;
                float Ca = Cacc + pm * a.alpha;
                Ca = Ca + a.lmbda;
                wp = wacc - pm * a.alpha * (pdisp - mono_v);
                Q = 1.0f / Ca;
                a.qw[rec.trk0 + lane] = make_float2(Q, wp);
            }
            if (MODE == kModeFull) {
                if (la != 0xffu) {
                    float *row = Eh + la * 6 * kLdsRowStride + lane;
#pragma unroll
                    for (int c = 0; c < 6; ++c) row[c * kLdsRowStride] += Ei[c];
                }
                Qs[lane] = Q;
                Eh[R * kLdsRowStride + lane] = wp;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // E Q w' (the Schur term of y): every lane scales its own column, 32 rows per reduce-scatter
                const float beta = Q * wp;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (32 * c < R) {
                        float v[32];
#pragma unroll
                        for (int i = 0; i < 32; ++i) v[i] = 32 * c + i < R ? beta * Eh[(32 * c + i) * kLdsRowStride + lane] : 0.0f;
                        wave_reduce_scatter32(v, lane);
                        yacc[c] += (double)v[0];
                    }
                }
                stage_b();
                schur_acc<NT>(Eh, Qs, R, lane, sacc);
                acc_live = true; Racc = R;
            } else {
                stage_b();
            }
        }

        // ---- rotate
        if (has_next) {
            rec = rec_n; kx_c = kx_n; la_c = la_n; grp = grp_n;
            px = px_n; py = py_n; pdisp = pd_n; mono_v = mono_n;
        }
    }
    flush_pairs();
    flush_schur();
}

// ------------------------------------------------------------------ dispatch
static int stream_threshold() {
    static const int t = std::getenv("BT_STREAM_MIN_TILES") ? std::atoi(std::getenv("BT_STREAM_MIN_TILES")) : 2048;   // measurement only
    return t;
}

static size_t stream_lds_bytes(const PlanDev &pd, int mode) {
    const size_t mtp = (size_t)(pd.max_tile_pairs > 0 ? pd.max_tile_pairs : 1);
    const size_t Rmax = (size_t)(6 * pd.max_cams);
    if (mode == kModeUpd) return mtp * kUpdGeoS * sizeof(float);
    if (mode == kModeSO) return mtp * kPairGeomFloats * sizeof(float);
    return (mtp * kPairGeomFloats + (Rmax + 1) * kLdsRowStride + 64 + ((Rmax + 3) & ~(size_t)3) + ((mtp + 3) & ~(size_t)3)) * sizeof(float) +
           mtp * 32 * sizeof(double) + 16;
}

// The streaming kernels take graphs of many tiles whose tiles see at most 10 cameras (row tiles of the register
// accumulators) and 32 camera pairs (one lane per pair in the prologue, LDS of the per-pair sums).
bool stream_applies(const PlanDev &pd) {
    return pd.T >= stream_threshold() && pd.max_cams <= 10 && pd.max_tile_pairs <= 32 && pd.max_tile_pairs > 0;
}

template <int MODE, int NT>
static int launch_stream_t(const PlanDev &pd, const StepArgs &a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    const size_t lds = stream_lds_bytes(pd, MODE);
    static int per_cu[4] = {0, 0, 0, 0};
    static size_t per_cu_lds[4] = {0, 0, 0, 0};
    static int n_cu = 0;
    const int slot = MODE == kModeFull ? (NT == 3 ? 0 : 1) : MODE + 1;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return BT_EHIP;
        n_cu = prop.multiProcessorCount;
    }
    if (!per_cu[slot] || per_cu_lds[slot] != lds) {
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stream<MODE, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return BT_EHIP;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_stream<MODE, NT>, 64, lds) != hipSuccess || nb < 1) nb = 1;
        static const int cap = std::getenv("BT_STREAM_WAVES_PER_CU") ? std::atoi(std::getenv("BT_STREAM_WAVES_PER_CU")) : 0;   // measurement only
        if (cap > 0 && nb > cap) nb = cap;
        per_cu[slot] = nb; per_cu_lds[slot] = lds;
    }
    const int max_waves = n_cu * per_cu[slot];
    const int tpw = (pd.T + max_waves - 1) / max_waves, nw = (pd.T + tpw - 1) / tpw;
    if (ev0) hipExtLaunchKernelGGL((k_stream<MODE, NT>), dim3(nw), dim3(64), lds, st, ev0, ev1, 0, pd, a, tpw);
    else hipLaunchKernelGGL((k_stream<MODE, NT>), dim3(nw), dim3(64), lds, st, pd, a, tpw);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

int launch_stream(const PlanDev &pd, const StepArgs &a, int mode, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    if (mode == kModeSO) return launch_stream_t<kModeSO, 1>(pd, a, st, ev0, ev1);
    if (mode == kModeUpd) return launch_stream_t<kModeUpd, 1>(pd, a, st, ev0, ev1);
    if (pd.max_cams <= 8) return launch_stream_t<kModeFull, 3>(pd, a, st, ev0, ev1);
    return launch_stream_t<kModeFull, 4>(pd, a, st, ev0, ev1);
}

}  // namespace bt
