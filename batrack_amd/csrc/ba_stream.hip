// ba_stream.hip — k_stream: tiles streamed through two-wave workgroups, for graphs of 2048 .. ~6000 tiles (gfx950, wave64).
//
// The tile kernel of ba_kernels.hip spreads one tile of 64 tracks over a workgroup of 8-16 waves: right for
// graphs of a few hundred tiles, where a tile's latency is all there is.  On graphs of thousands of tiles the
// chain  tile tables -> poses / patches -> edge list -> targets -> barriers -> Schur product -> atomics  of ONE
// tile at a time keeps a CU's pipes idle.  Here a workgroup is TWO waves that own whole tiles and walk a contiguous
// range of them (LDS private to the pair, one barrier per tile where their partial sums merge):
//   * lane l = track l of the tile (as in k_tile); each wave runs half of the tile's edge slots, in groups of kSG
//     slots whose operands (edge ids, 16-bit slot codes, targets, weights) are loaded in one burst — the first
//     group of tile t+1 while tile t is still being computed (tables at its top, the gathers before its Schur
//     product), so the tile-to-tile critical path holds no exposed memory round trip;
//   * local E in LDS (lane-private columns: plain read-add-write; the planner flags the tiles where the two waves'
//     slot ranges split a run of repeated observations), per-pair sums per wave in LDS doubles;
//   * the Schur product E Q E^T on v_mfma_f64_16x16x4_f64, its row tiles split between the two waves, with the
//     accumulators kept in REGISTERS across consecutive tiles with the same cameras (tracks of one frame), atomics
//     only when the cameras change; E Q w' by a wave reduce-scatter;
//   * MODE kModeSO: structure-only steps (C, w, Q, w' only);  kModeUpd: the depth back-substitution of k_update
//     (ba_kernels.hip) for the same tile walk.
// Which graphs take this kernel: stream_applies() below and edge_applies() in ba_stream3.hip (profiles/r02_kernel_choice.txt).
// Reference: ba.py:228-337, projective_ops.py:54-100.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>

#include "ba_edge.hpp"
#include "ba_kernels.hpp"
#include "dev_cache.hpp"

namespace bt {

#ifndef BT_STREAM_FULL_WAVES
#define BT_STREAM_FULL_WAVES 2
#endif
constexpr int kSG = 4;                 // slots per operand group (a wave's share of an 8-observation tile)
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
            const unsigned idx = (unsigned)(slot_base + s) * kLanes + (unsigned)lane;     // 32-bit offsets: scalar base + vector offset addressing
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
            const unsigned to = (unsigned)e[s] * (unsigned)a.tstride;               // (launch_stream checks that byte offsets fit 32 bits)
            g.tu[s] = a.targets[to]; g.tv[s] = a.targets[to + 1u];
            const float2 w = reinterpret_cast<const float2 *>(a.weights)[(unsigned)e[s]];
            g.w0[s] = w.x; g.w1[s] = w.y;
        }
    }
}

// One row tile (16 rows) of the tile's Schur product: out tiles (TI, 0..TI) += A(TI) diag(Q) B(tj)^T over the 64 tracks
template <int TI>
__device__ __forceinline__ void schur_row(const float *Eh, const float (&qv)[16], int R, int lane, double4_t *acc) {
    if (16 * TI >= R) return;
    // operand pointers = one lane-dependent base each, the k step as a constant offset (folds into the ds_read)
    const int kq = lane >> 4, li = lane & 15;
    const float *ap = Eh + min(16 * TI + li, R) * kLdsRowStride + kq;     // rows beyond the tile's E read row R: their outputs are never emitted
    double av[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) av[ks] = (double)ap[4 * ks] * (double)qv[ks];
#pragma unroll
    for (int tj = 0; tj <= TI; ++tj) {
        __builtin_amdgcn_sched_barrier(0);              // one output tile's operands at a time
        const float *bp = Eh + min(16 * tj + li, R) * kLdsRowStride + kq;
        float bv[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) bv[ks] = bp[4 * ks];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            acc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], (double)bv[ks], acc[tj], 0, 0, 0);
    }
}

template <int TI>
__device__ __forceinline__ void flush_row(const PlanDev &pd, const StepArgs &a, const int *gidx, int Racc, int lane, double4_t *acc) {
#pragma unroll
    for (int tj = 0; tj <= TI; ++tj) {
        const int col = 16 * tj + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * TI + (lane >> 4) + 4 * r;
            const double val = acc[tj][r];
            acc[tj][r] = 0.0;
            if (row < Racc && col < Racc) {
                const int gr = gidx[row], gc = gidx[col];
                if (gr >= gc) atomicAdd(&a.S[(size_t)gr * pd.D + gc], -val);
            }
        }
    }
}

// Row tiles of the Schur product per wave (NT row tiles in all; a row tile TI has TI + 1 output tiles):
//   NT = 1: wave 0 {0}          NT = 2: wave 0 {1}, wave 1 {0}
//   NT = 3: wave 0 {2}, wave 1 {0, 1}                  (3 + 3 output tiles)
//   NT = 4: wave 0 {3, 0}, wave 1 {1, 2}               (5 + 5)
// acc[] of a wave: its row tiles' output tiles one after the other.
template <int NT, int WAVE, int PART>
__device__ __forceinline__ void schur_mine(const float *Eh, const float *Qs, int R, int lane, double4_t *acc) {
    float qv[16];
    const float *qp = Qs + (lane >> 4);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) qv[ks] = qp[4 * ks];
    // PART 0: the first row tile of the wave (the gathers of the next tile are issued behind it), PART 1: the rest
    if (WAVE == 0) {
        if (PART == 0) schur_row<NT - 1>(Eh, qv, R, lane, acc);
        if (PART == 1 && NT == 4) schur_row<0>(Eh, qv, R, lane, acc + NT);
    } else {
        if (NT == 2 && PART == 0) schur_row<0>(Eh, qv, R, lane, acc);
        if (NT == 3) { if (PART == 0) schur_row<0>(Eh, qv, R, lane, acc); else schur_row<1>(Eh, qv, R, lane, acc + 1); }
        if (NT == 4) { if (PART == 0) schur_row<1>(Eh, qv, R, lane, acc); else schur_row<2>(Eh, qv, R, lane, acc + 2); }
    }
}
template <int NT, int WAVE>
__device__ __forceinline__ void flush_mine(const PlanDev &pd, const StepArgs &a, const int *gidx, int Racc, int lane, double4_t *acc) {
    if (WAVE == 0) {
        flush_row<NT - 1>(pd, a, gidx, Racc, lane, acc);
        if (NT == 4) flush_row<0>(pd, a, gidx, Racc, lane, acc + NT);
    } else {
        if (NT == 2) flush_row<0>(pd, a, gidx, Racc, lane, acc);
        if (NT == 3) { flush_row<0>(pd, a, gidx, Racc, lane, acc); flush_row<1>(pd, a, gidx, Racc, lane, acc + 1); }
        if (NT == 4) { flush_row<1>(pd, a, gidx, Racc, lane, acc); flush_row<2>(pd, a, gidx, Racc, lane, acc + 2); }
    }
}
constexpr int stream_nacc(int nt) { return nt == 4 ? 5 : nt == 3 ? 3 : nt == 2 ? 2 : 1; }

// MODE, NT: row tiles (16 rows each) of the largest local E the instantiation accumulates (kModeFull only)
template <int MODE, int NT, bool PROF = false>
__global__ __launch_bounds__(128, MODE == kModeFull ? BT_STREAM_FULL_WAVES : 4) void k_stream(PlanDev pd, StepArgs a, int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc = PROF ? clock64() : 0, tn;
#define BT_PF(i) do { if (PROF) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tn = clock64(); pf[i] += tn - tc; tc = tn; __builtin_amdgcn_sched_barrier(0); } } while (0)
    const int mtp = pd.max_tile_pairs > 0 ? pd.max_tile_pairs : 1;
    const int Rmax = 6 * pd.max_cams;
    constexpr int GS = MODE == kModeUpd ? kUpdGeoS : kPairGeomFloats;
    // LDS carve-up (private to the two waves of the tile)
    float *geo = lds;                                                  // [mtp][GS]
    float *part = geo + mtp * GS;                                      // [2 waves][8][64]: C, w, source-camera E partials (kModeUpd: depth sums)
    float *Eh = part + 1024;                                           // [(Rmax + 1)][66]: local E, row R of the tile = w'
    float *Qs = Eh + (MODE == kModeFull ? (Rmax + 1) * kLdsRowStride : 0);
    int *gidx = reinterpret_cast<int *>(Qs + (MODE == kModeFull ? 128 : 0));   // Qs: one copy of Q per wave
    int *gpl = gidx + (MODE == kModeFull ? ((Rmax + 3) & ~3) : 0);
    double *pacc = reinterpret_cast<double *>(gpl + (MODE == kModeFull ? ((mtp + 3) & ~3) : 0)) + wave * mtp * 32;   // [2 waves][mtp][32]: this wave's

    // workgroups are dealt round-robin to the 8 XCDs: an XCD's workgroups take neighbouring tile ranges (shared cameras, pair
    // geometry and rows of S stay in one L2): 48.7 -> 46.0 us at 2048 tiles
    const int wq_ = gridDim.x >> 3, wr_ = gridDim.x & 7, wx_ = blockIdx.x & 7;
    const int wg_ = wx_ * wq_ + min(wx_, wr_) + ((int)blockIdx.x >> 3);
    const int t_begin = wg_ * tiles_per_wg, t_end = min(pd.T, t_begin + tiles_per_wg);
    if (t_begin >= t_end) return;

    constexpr int NACC = stream_nacc(NT);
    double4_t sacc[NACC];
    double yacc = 0.0;
#pragma unroll
    for (int t = 0; t < NACC; ++t) sacc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
    bool acc_live = false;
    int Racc = 0, np_acc = 0;

    auto flush_schur = [&]() {
        if (MODE != kModeFull || !acc_live) return;
        if (wave == 0) flush_mine<NT, 0>(pd, a, gidx, Racc, lane, sacc);
        else           flush_mine<NT, 1>(pd, a, gidx, Racc, lane, sacc);
        const int row = 32 * wave + ((lane >> 1) & 31);            // E Q w': rows 0..31 on wave 0, 32..63 on wave 1
        if (!(lane & 1) && row < Racc) atomicAdd(&a.y[gidx[row]], -yacc);
        yacc = 0.0;
        acc_live = false;
    };
    auto flush_pairs = [&]() {                                      // each wave its own sums, two pairs per pass
        if (MODE != kModeFull) return;
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
    if (MODE == kModeFull) {
        for (int i = lane; i < mtp * 32; i += 64) pacc[i] = 0.0;
    }

    // ---- this wave's share of a tile's slots: a contiguous chunk (repeated observations of one (track, camera)
    // are consecutive slots)
    auto chunk_of = [&](const TileRec &r, int &s0, int &s1) {
        const int ch = (r.nslot + 1) >> 1;
        s0 = wave * ch; s1 = min(r.nslot, s0 + ch);
    };

    // ---- operands of the first tile
    TileRec rec = load_rec(pd, t_begin);
    TileRec rec_n = t_begin + 1 < t_end ? load_rec(pd, t_begin + 1) : rec;       // records run one tile ahead of the other operands
    int kx_c = pd.tile_kx[(unsigned)t_begin * kLanes + (unsigned)lane];
    unsigned la_c = pd.tile_la[(unsigned)t_begin * kLanes + (unsigned)lane];
    Grp grp;
    {
        int s0, s1;
        chunk_of(rec, s0, s1);
        int e[kSG]; unsigned code[kSG];
        load_tables(pd, rec.slot0 + s0, min(max(s1 - s0, 0), kSG), lane, e, code);
        load_gather(a, e, code, grp);
    }
    float px = 0.0f, py = 0.0f, pdisp = 0.0f, mono_v = 0.0f;
    if (kx_c >= 0) {
        px = a.patches[3u * (unsigned)kx_c]; py = a.patches[3u * (unsigned)kx_c + 1u]; pdisp = a.patches[3u * (unsigned)kx_c + 2u];
        if (MODE != kModeUpd) mono_v = a.mono[(unsigned)kx_c * (unsigned)a.mstride];
    }

#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        // (the lane index is re-derived per tile behind an opaque barrier: the dozens of lane-dependent LDS offsets of the
        // unrolled code below are then recomputed where they are used instead of living in registers across the loop)
        const int lane_outer = lane;
        int lane = lane_outer;
        asm volatile("" : "+v"(lane));
        const int flags_t = rec.flags;
        const int flags = tile == t_begin ? 0 : flags_t;
        const int R = 6 * rec.ncam;
        const bool has_next = tile + 1 < t_end;
        // ---- new cameras / new pair list (the barrier at the end of the previous tile has passed: nobody reads the old ones)
        if (MODE == kModeFull && !(flags & 1)) {
            flush_schur();
            const int *cams = pd.tile_cams + rec.cam0;
            for (int i = tid; i < R; i += 128) gidx[i] = 6 * cams[i / 6] + i % 6;
        }
        if (!(flags & 2)) {
            flush_pairs();
            for (int p = tid; p < rec.npair; p += 128) {
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
                    const int ij = pd.tile_ij[(unsigned)tile * (unsigned)mtp + (unsigned)p];
                    pair_geometry<float, BT_WPT_MIXED != 0>(a.poses, a.intr, ij & 0xffff, ij >> 16, g);
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
            float4 *z = reinterpret_cast<float4 *>(Eh);
            for (int i = tid; i < (R * kLdsRowStride + 3) / 4; i += 128) z[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        __syncthreads();
        BT_PF(0);

        // ---- this wave's slots of the tile
        float Cacc = 0.0f, wacc = 0.0f, Ei[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, dacc = 0.0f;
        const unsigned la = la_c;
        int s0, s1;
        chunk_of(rec, s0, s1);
#pragma unroll 1
        for (int sb = s0; sb < s1; sb += kSG) {
            if (sb > s0) {                       // (chunks of more than kSG slots: later groups are loaded in place)
                int e[kSG]; unsigned code[kSG];
                load_tables(pd, rec.slot0 + sb, min(s1 - sb, kSG), lane, e, code);
                load_gather(a, e, code, grp);
            }
            const int ns = min(s1 - sb, kSG);
#pragma unroll
            for (int s = 0; s < kSG; ++s) {
                if (s >= ns) break;
                __builtin_amdgcn_sched_barrier(0);      // one slot's working set at a time (the unrolled bodies would otherwise be interleaved)
                const unsigned code = grp.code[s];
                const bool act = (code >> 16) != 0;
                const unsigned lb = code & 0xffu, lp = (code >> 8) & 0xffu;
                float g[GS];
                {
                    const float4 *g4 = reinterpret_cast<const float4 *>(geo + (size_t)lp * GS);
#pragma unroll
                    for (int c = 0; c < GS / 4; ++c) {
                        const float4 t4 = g4[c];
                        g[4*c] = t4.x; g[4*c + 1] = t4.y; g[4*c + 2] = t4.z; g[4*c + 3] = t4.w;
                    }
                }
                EdgeQ q;
                BT_WPT_EDGE_EVAL(g, px, py, pdisp, grp.tu[s], grp.tv[s], grp.w0[s], grp.w1[s], a, q);
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
                    if (flags_t & 4) {                // a repeated (track, camera) observation spans both waves' chunks somewhere in this tile
#pragma unroll
                        for (int c = 0; c < 6; ++c) atomicAdd(row + c * kLdsRowStride, Ej[c]);
                    } else {
                        float old[6];
#pragma unroll
                        for (int c = 0; c < 6; ++c) old[c] = row[c * kLdsRowStride];
#pragma unroll
                        for (int c = 0; c < 6; ++c) row[c * kLdsRowStride] = old[c] + Ej[c];
                    }
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
        BT_PF(1);

        // ---- next tile: stage A (its patch index and this wave's slot tables) is requested once the tile's own
        // sums are done, stage B (the gathers through the tables) behind the first part of the Schur product, under
        // whose rest they land; the tile records run one tile further ahead
        int kx_n = -1;
        unsigned la_n = 0xffu;
        int e_n[kSG];
        unsigned code_n[kSG];
        TileRec rec_nn = rec_n;
        auto stage_a = [&]() {
            if (tile + 2 < t_end) rec_nn = load_rec(pd, tile + 2);
            if (has_next) {
                kx_n = pd.tile_kx[(unsigned)(tile + 1) * kLanes + (unsigned)lane];
                la_n = pd.tile_la[(unsigned)(tile + 1) * kLanes + (unsigned)lane];
                int n0, n1;
                chunk_of(rec_n, n0, n1);
                load_tables(pd, rec_n.slot0 + n0, min(max(n1 - n0, 0), kSG), lane, e_n, code_n);
            }
        };
        Grp grp_n;
        float px_n = 0.0f, py_n = 0.0f, pd_n = 0.0f, mono_n = 0.0f;
        auto stage_b = [&]() {
            if (has_next) {
                load_gather(a, e_n, code_n, grp_n);
                if (kx_n >= 0) {
                    px_n = a.patches[3u * (unsigned)kx_n]; py_n = a.patches[3u * (unsigned)kx_n + 1u]; pd_n = a.patches[3u * (unsigned)kx_n + 2u];
                    if (MODE != kModeUpd) mono_n = a.mono[(unsigned)kx_n * (unsigned)a.mstride];
                }
            }
        };

        // ---- the two waves' partial sums meet in LDS
        const bool has_trk = lane < rec.ntrk;
        if (MODE == kModeUpd) {
            part[wave * 64 + lane] = dacc;
        } else {
            float *mine = part + wave * 512 + lane;
            mine[0] = Cacc; mine[64] = wacc;
            if (MODE == kModeFull) {
#pragma unroll
                for (int c = 0; c < 6; ++c) mine[(2 + c) * 64] = Ei[c];
            }
        }
        __syncthreads();
        BT_PF(2);
        if (MODE == kModeUpd) {
            if (wave == 0 && has_trk) {
                const float tot = part[lane] + part[64 + lane];
                const float2 qw = a.qw[(unsigned)rec.trk0 + (unsigned)lane];
                float dd = pdisp + qw.x * (qw.y - tot);                          // ba.py:328, :333
                dd = dd < 1e-3f ? 1e-3f : dd;
                dd = dd > 10.0f ? 10.0f : dd;
                a.patches_out[3u * (unsigned)kx_c] = px; a.patches_out[3u * (unsigned)kx_c + 1u] = py; a.patches_out[3u * (unsigned)kx_c + 2u] = dd;
            }
            stage_a();
            stage_b();
        } else {
            // both waves form Q and w' from the two partials (same values: either may write them)          ba.py:296-311
            const float C = part[lane] + part[512 + lane], wv = part[64 + lane] + part[576 + lane];
            if (MODE == kModeFull && la != 0xffu) {
                // the track's source-camera rows: components 0-2 merged by wave 0, 3-5 by wave 1 (after the barrier no
                // slot touches E any more; self edges, whose target rows are these, were added before it)
                float *row = Eh + (la * 6 + 3 * wave) * kLdsRowStride + lane;
                const float *p0 = part + (2 + 3 * wave) * 64 + lane;
#pragma unroll
                for (int c = 0; c < 3; ++c) row[c * kLdsRowStride] += p0[c * 64] + p0[512 + c * 64];
            }
            float Q = 0.0f, wp = 0.0f;
            if (has_trk) {
                const float pm = mono_v > 1e-2f ? 1.0f : 0.0f;
                float Ca = C + pm * a.alpha;
                Ca = Ca + (a.lmbda_trk ? a.lmbda_trk[(unsigned)pd.trk_off + (unsigned)rec.trk0 + (unsigned)lane] : a.lmbda);
                wp = wv - pm * a.alpha * (pdisp - mono_v);
                Q = 1.0f / Ca;
                if (wave == 0) a.qw[(unsigned)rec.trk0 + (unsigned)lane] = make_float2(Q, wp);
            }
            if (MODE == kModeFull) {
                __syncthreads();                   // E complete
                // E Q w' (the Schur term of y): every lane scales its own column; wave w reduces rows 32 w .. 32 w + 31
                const float beta = Q * wp;
                if (32 * wave < R) {
                    float v[32];
#pragma unroll
                    for (int i = 0; i < 32; ++i) v[i] = 32 * wave + i < R ? beta * Eh[(32 * wave + i) * kLdsRowStride + lane] : 0.0f;
                    wave_reduce_scatter32(v, lane);
                    yacc += (double)v[0];
                }
                BT_PF(3);
                stage_a();
                // the wave's own copy of Q for the operand loads (written and read by this wave only)
                float *Qw = Qs + wave * 64;
                Qw[lane] = Q;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (wave == 0) schur_mine<NT, 0, 0>(Eh, Qw, R, lane, sacc);
                else           schur_mine<NT, 1, 0>(Eh, Qw, R, lane, sacc);
                if (wave == 0) schur_mine<NT, 0, 1>(Eh, Qw, R, lane, sacc);
                else           schur_mine<NT, 1, 1>(Eh, Qw, R, lane, sacc);
                BT_PF(4);
                stage_b();            // (behind the Schur product: its operands and the gathers' results do not fit the registers together)
                acc_live = true; Racc = R;
                BT_PF(5);
            } else {
                stage_a();
                stage_b();
            }
        }
        __syncthreads();                         // the next tile clears / rewrites Eh, part, geo

        // ---- rotate
        if (has_next) {
            rec = rec_n; rec_n = rec_nn; kx_c = kx_n; la_c = la_n; grp = grp_n;
            px = px_n; py = py_n; pdisp = pd_n; mono_v = mono_n;
        }
    }
    flush_pairs();
    flush_schur();
    BT_PF(6);
    if (PROF && lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BT_PF(7);
        long long *o = reinterpret_cast<long long *>(a.status + 4) + (blockIdx.x == 0 ? 20 : 30);
        for (int i = 0; i < 8; ++i) o[i] = pf[i];
        o[8] = t_end - t_begin;
    }
#undef BT_PF
}

// ------------------------------------------------------------------ dispatch

static size_t stream_lds_bytes(const PlanDev &pd, int mode) {
    const size_t mtp = (size_t)(pd.max_tile_pairs > 0 ? pd.max_tile_pairs : 1);
    const size_t Rmax = (size_t)(6 * pd.max_cams);
    if (mode == kModeUpd) return (mtp * kUpdGeoS + 1024) * sizeof(float);
    if (mode == kModeSO) return (mtp * kPairGeomFloats + 1024) * sizeof(float);
    return (mtp * kPairGeomFloats + 1024 + (Rmax + 1) * kLdsRowStride + 128 + ((Rmax + 3) & ~(size_t)3) + ((mtp + 3) & ~(size_t)3)) * sizeof(float) +
           2 * mtp * 32 * sizeof(double) + 16;
}

// The streaming kernels take graphs of many tiles whose tiles see at most 10 cameras (row tiles of the register
// accumulators) and 32 camera pairs (one lane per pair in the prologue, LDS of the per-pair sums).
bool stream_applies(const PlanDev &pd) {
    return pd.st_ok != 0 && pd.T >= pd.st_min && pd.max_cams <= 10 && pd.max_tile_pairs <= 32 && pd.max_tile_pairs > 0;
}

template <int MODE, int NT, bool PROF = false>
static int launch_stream_t(const PlanDev &pd, const StepArgs &a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    const size_t lds = stream_lds_bytes(pd, MODE);
    // (per device: the LDS limit of a kernel and what fits a CU are properties of the device the launch goes to)
    static LdsLimit lds_limit;
    static std::atomic<int> per_cu_c[kMaxDevices];
    static std::atomic<size_t> per_cu_lds[kMaxDevices];
    DevProps dp;
    if (!device_props(&dp, pd.dev_id)) return BT_EHIP;
    const int n_cu = dp.n_cu, dslot = dp.dev >= 0 && dp.dev < kMaxDevices ? dp.dev : 0;
    int per_cu = per_cu_lds[dslot].load(std::memory_order_acquire) == lds ? per_cu_c[dslot].load(std::memory_order_relaxed) : 0;
    if (!per_cu) {
        if (!lds_limit.ensure(reinterpret_cast<const void *>(&k_stream<MODE, NT, PROF>), lds, pd.dev_id)) return BT_EHIP;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_stream<MODE, NT, PROF>, 128, lds) != hipSuccess || nb < 1) nb = 1;
        per_cu = nb;
        per_cu_c[dslot].store(nb, std::memory_order_relaxed); per_cu_lds[dslot].store(lds, std::memory_order_release);
    }
    const int max_waves = n_cu * per_cu;
    const int tpw = (pd.T + max_waves - 1) / max_waves, nw = (pd.T + tpw - 1) / tpw;
    if (ev0) hipExtLaunchKernelGGL((k_stream<MODE, NT, PROF>), dim3(nw), dim3(128), lds, st, ev0, ev1, 0, pd, a, tpw);
    else hipLaunchKernelGGL((k_stream<MODE, NT, PROF>), dim3(nw), dim3(128), lds, st, pd, a, tpw);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

int launch_stream(const PlanDev &pd, const StepArgs &a, int mode, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    if ((unsigned long long)pd.e_all * (unsigned long long)a.tstride * 4ull >= (1ull << 32)) return BT_EUNSUPPORTED;   // 32-bit byte offsets into the targets
    if ((unsigned long long)pd.p_tot * (unsigned long long)a.mstride * 4ull >= (1ull << 32)) return BT_EUNSUPPORTED;
    if (mode == kModeSO) return launch_stream_t<kModeSO, 1>(pd, a, st, ev0, ev1);
    if (mode == kModeUpd) return launch_stream_t<kModeUpd, 1>(pd, a, st, ev0, ev1);
    if (pd.max_cams <= 8 && (a.dbg & 32)) return launch_stream_t<kModeFull, 3, true>(pd, a, st, ev0, ev1);
    if (pd.max_cams <= 8) return launch_stream_t<kModeFull, 3>(pd, a, st, ev0, ev1);
    return launch_stream_t<kModeFull, 4>(pd, a, st, ev0, ev1);
}

}  // namespace bt
