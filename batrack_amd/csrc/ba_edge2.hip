// ba_edge2.hip — k_edge2: the pose+structure Jacobian kernel for graphs of many SLOT-UNIFORM tiles, TWO EDGES PER LANE
// (gfx950, wave64).  Round 5: what bounded round 4's one-edge-per-lane k_edge (deleted) was the number of vector instructions a wave has to issue
// (~430 per edge-lane, profiles/r04_pmc_sq_e8m.txt), not bytes.  Same edge-major tables (it_edge, tile_sinfo, tile_rec), but
//   * a STEP is two iterations of the table: a lane handles the same slot of TWO tracks (track t and t + G of the step's 2G
//     tracks).  Both edges meet the same camera pair, so they share the pair's geometry (one set of LDS reads) and the 26
//     per-pair sums, and every float32 operation of the Jacobians, robust weights and products is ONE packed instruction for
//     the two (v_pk_fma_f32 / v_pk_mul_f32 on explicit 2-vectors; SLP auto-vectorisation is off for this library);
//   * the float64 reprojection reads the pair's geometry from LDS AS DOUBLES (no 20 conversions per edge) and the track's
//     normalised source coordinates (X0, Y0) = ((x - cx_i) / fx_i, (y - cy_i) / fy_i), which depend on the track alone, are
//     formed once per track at the tile's top (the tracks of a slot-uniform tile share their source camera);
//   * the per-pair sums stay in (packed float32) registers across the consecutive tiles that keep the lanes' pairs — a wave
//     walks 8+ tiles of one source frame — and go to the float64 accumulators in global memory every kPaFlushTiles tiles;
//   * a step holds ALL S slots of its 2G tracks, so their rows of E, their C and w are complete when the step's edges are
//     done: Q and w' are formed right there by the tracks' first lanes, and E Q E^T of those tracks goes onto the matrix pipe
//     in the same step (v_mfma_f32_16x16x4_f32, K = the step's 16 tracks, from ZERO accumulators: float32 chains longer than
//     that cost the update its fifth digit — DESIGN.md §4), the step's products added to float64 sums in LDS that run across
//     the steps and tiles of one camera set (kSchurFlushTiles at most).  The local E is [rows][16 tracks] (3.8 KB instead of
//     13), there is no per-tile phase — no merge barrier, no pass over the tile's tracks, no clearing of E (every element of a
//     step's E is stored by exactly one lane, zeros where an edge is missing; repeated observations of a (track, camera) are
//     added up across their lanes first);
//   * the kernel is as long as its slowest wave (tools/gpu_wave_times.py): the two waves of a SIMD take turns at the higher
//     issue priority (the arbiter's default serves the older one first: 83-96 against 110-120 us for the same 8 tiles), the
//     launch's first waves take the last tiles, workgroups of 4 waves add their sums up before the atomics;
//   * the gathered operands of a step live in one of two register sets by the step's parity and are re-loaded in place two
//     steps ahead, the edge ids likewise: no register is moved, no load is waited for before its use.
// Reference: ba.py:228-337, projective_ops.py:54-100.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "ba_edge2.hpp"
#include "dev_cache.hpp"
#include "probe.hpp"

namespace bt {
namespace e2 {

#ifndef BT_EDGE2_PA_TILES
#define BT_EDGE2_PA_TILES 16
#endif
// tiles a lane's float32 pair sums run before they are added to the float64 sums (measured on the benchmark graphs: 4 / 8 / 16
// tiles give the same S, y and dX against the oracle; every flush is 26 wave-wide atomics on lines other waves hit as well)
constexpr int kPaFlushTiles = BT_EDGE2_PA_TILES;
#ifndef BT_EDGE2_SCHUR_TILES
#define BT_EDGE2_SCHUR_TILES 4096
#endif
// tiles of one camera set after which the float64 Schur sums in LDS go to S (they are exact enough to run for ever; the bound
// only keeps a flush from being the wave's very last act in graphs of one camera set)
constexpr int kSchurFlushTiles = BT_EDGE2_SCHUR_TILES;

// floats per row of the local E: the step's tracks (2G, at least the 16 of one k-sweep of the matrix pipe) + 4 (rows 16-byte
// aligned for the b128 operand reads, neighbouring rows on different banks)
__host__ __device__ constexpr int e_row(int lgS) { return (lgS >= 3 ? 16 : lgS == 2 ? 32 : 64) + 4; }

// E Q E^T and E (Q w') of a step's tracks, for a tile whose E has NTL row tiles of 16 (compile time: straight-line code).
// Lane (li, kq) supplies, for k-step (v, c), the element of track 16 v + 4 kq + c (one b128 read per 16 tracks: the k order of
// a matrix product is free); rows beyond the tile's E re-read its last row, their outputs are never emitted.  The products of
// the step's (at most 16 per k-sweep) tracks are summed on the matrix pipe's float32 accumulators from zero, k-step by k-step
// ACROSS the output tiles (a product into the accumulator of the one before it waits for that one's eight passes), and the
// step's sums are added to FLOAT64 sums in LDS (lane-private 32 bytes per output tile).  Float32 across a tile's 64 tracks —
// let alone across tiles — is not enough: S = B - E Q E^T cancels in the directions the solve amplifies (measured on the
// benchmark graphs: dX 4e-5 from the oracle with 64-track float32 chains, 3e-6 .. 1e-5 with 16-track ones — S, y alike 2e-7).
// `filler`: independent vector work of the caller placed in the same scheduling region as the matrix products, for the scheduler
// to put between them (none at present: see the caller).
template <int NTL, int NT, int KQN, int ROW, typename F>
__device__ __forceinline__ void schur_rows(const float *Eh, const float *Qs, const float *Bs, double *sacc, float (&yacc)[NT], int R, int nv, int lane, F &&filler) {
    const int kq = lane >> 4, li = lane & 15;
    constexpr int NA = NTL * (NTL + 1) / 2;
    // (layout of the float64 sums: [tile][half][lane] of 16-byte pieces, consecutive lanes on consecutive banks)
    double2 *dp = reinterpret_cast<double2 *>(sacc) + lane;
    static_assert(KQN == 16 || NTL >= 0, "");
    float4_t c[NA];
#pragma unroll
    for (int t = 0; t < NA; ++t) c[t] = float4_t{0.0f, 0.0f, 0.0f, 0.0f};
    // the float64 sums of the first half of the output tiles are requested before the products (they arrive under them), those
    // of the second half before the first half is folded: no fold waits for an LDS round trip of its own
    constexpr int NH = (NA + 1) / 2;
    double2 sm[NH][2];
#pragma unroll
    for (int t = 0; t < NH; ++t) { sm[t][0] = dp[t * 128]; sm[t][1] = dp[t * 128 + 64]; }
    BT_E2_SB;
#pragma unroll
    for (int v = 0; v < KQN / 16; ++v) {
        if (v < nv) {
            const float4 q4 = *reinterpret_cast<const float4 *>(Qs + 16 * v + 4 * kq), be4 = *reinterpret_cast<const float4 *>(Bs + 16 * v + 4 * kq);
            float af[NTL][4], aq[NTL][4];
#pragma unroll
            for (int ti = 0; ti < NTL; ++ti) {
                const float4 a4 = *reinterpret_cast<const float4 *>(Eh + min(16 * ti + li, R - 1) * ROW + 16 * v + 4 * kq);
                af[ti][0] = a4.x; af[ti][1] = a4.y; af[ti][2] = a4.z; af[ti][3] = a4.w;
                const f2 q01 = f2{a4.x, a4.y} * f2{q4.x, q4.y}, q23 = f2{a4.z, a4.w} * f2{q4.z, q4.w};
                aq[ti][0] = q01.x; aq[ti][1] = q01.y; aq[ti][2] = q23.x; aq[ti][3] = q23.y;
            }
            // the products, k-step by k-step across the output tiles of a row group (rows ti < NTL - 1, then the last row): while
            // the matrix pipe works on the second group the first group's sums are folded on the vector pipe
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
                const int ta = grp == 0 ? 0 : NTL - 1, tb = grp == 0 ? NTL - 1 : NTL;
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int ti = ta; ti < tb; ++ti)
#pragma unroll
                        for (int tj = 0; tj <= ti; ++tj)
                            c[ti * (ti + 1) / 2 + tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[ti][k], af[tj][k], c[ti * (ti + 1) / 2 + tj], 0, 0, 0);
            }
            if (v == 0) filler();
#pragma unroll
            for (int ti = 0; ti < NTL; ++ti) {
                // E (Q w'): the lane's four tracks, then the four lane groups (all of them end with the row's sum)
                const f2 y2 = fma2(f2{af[ti][0], af[ti][1]}, f2{be4.x, be4.y}, f2{af[ti][2], af[ti][3]} * f2{be4.z, be4.w});
                yacc[ti] += xor_add<32>(xor_add<16>(y2.x + y2.y));
            }
        }
    }
    // the step's sums to the float64 sums
    double2 sn[NA - NH > 0 ? NA - NH : 1][2];
#pragma unroll
    for (int t = NH; t < NA; ++t) { sn[t - NH][0] = dp[t * 128]; sn[t - NH][1] = dp[t * 128 + 64]; }
    BT_E2_SB;
#pragma unroll
    for (int t = 0; t < NH; ++t) {
        sm[t][0].x += (double)c[t][0]; sm[t][0].y += (double)c[t][1]; sm[t][1].x += (double)c[t][2]; sm[t][1].y += (double)c[t][3];
        dp[t * 128] = sm[t][0]; dp[t * 128 + 64] = sm[t][1];
    }
#pragma unroll
    for (int t = NH; t < NA; ++t) {
        sn[t - NH][0].x += (double)c[t][0]; sn[t - NH][0].y += (double)c[t][1]; sn[t - NH][1].x += (double)c[t][2]; sn[t - NH][1].y += (double)c[t][3];
        dp[t * 128] = sn[t - NH][0]; dp[t * 128 + 64] = sn[t - NH][1];
    }
}

// (BT_E2_PF(i), BT_PROBE_E2_*: measurement hooks, empty in the product build — probe.hpp)


// LGS: log2 of the slots per track when every tile of the plan has the same (straight-line lane-group reductions, E rows of
// 16 tracks), -1: read per tile (E rows for the 64 tracks a step of S <= 2 slots holds)
//
// Workgroup = the W waves of one CU (W * 64 threads, one workgroup per CU).  Every wave walks its own tile range with its own
// slice of LDS and never waits for another — until the end: what a wave has summed (E Q E^T and E Q w' in LDS, the pair sums in
// registers) would go to [S | y] and the pair accumulators as ~1400 float64 atomics, and two thousand waves ending together
// are throughput-bound on them (measured at 8.4M edges: 170 us with them, 135 without).  So the waves of a workgroup — ranges of
// consecutive tiles, as a rule of one source frame: the same cameras, the same pairs — add their sums up in LDS first (a
// binary tree, partners whose cameras / pairs differ keep theirs) and the atomics are issued for the tree's roots only.
template <int NT, int LGS, int LOSS>
__global__ __launch_bounds__(512, 1) void k_edge2(PlanDev pd, StepArgs a, int tiles_per_wave, int wave_doubles) {
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
    const int gw = blockIdx.x * nwv + wv;                              // this wave among the launch's
    double *lds_d = lds_all + (size_t)wv * wave_doubles;
    const int mtp = pd.max_tile_pairs > 0 ? pd.max_tile_pairs : 1;
    const int Rmax = 6 * pd.max_cams;
    constexpr int kRow = LGS >= 0 ? e_row(LGS) : e_row(0);
    constexpr int kQn = kRow - 4;                                      // tracks a step can hold
    // LDS carve-up (wave-private).  The per-lane arrays first, at compile-time offsets (one address register per element size,
    // the rest in the instructions' offset fields); what the plan sizes (E rows, pairs) behind them.
    double *ptD = lds_d;                                               // [64][2]: X0, Y0 of every track of the tile
    double *srcK = ptD + 128;                                          // 1/fx_i, 1/fy_i, cx_i, cy_i of the tile's source camera
    double *ysum = srcK + 4;                                           // [64]: E (Q w') since the last flush by local row
    double *sacc = ysum + 64;                                          // [NACC][64][4]: E Q E^T since the last flush (lane-private rows of the output tiles)
    float *ptF = reinterpret_cast<float *>(sacc + NT * (NT + 1) / 2 * 256);   // [64][2]: disparity, depth prior of every track of the tile
    float *Qs = ptF + 128;                                             // [kQn]: Q of the step's tracks
    float *Bs = Qs + kQn;                                              // [kQn]: beta = Q w'
    float *Eh = Bs + kQn;                                              // [Rmax][kRow]: E of the step's tracks
    int *gidx = reinterpret_cast<int *>(Eh + Rmax * kRow);             // [Rmax rounded up to 4]
    double *geoD = reinterpret_cast<double *>(gidx + ((Rmax + 3) & ~3));   // [mtp][kGeoDS]
    float *geoF = reinterpret_cast<float *>(geoD + mtp * kGeoDS);      // [mtp][kGeoFS]
    float *ptL = geoF + mtp * kGeoFS;                                  // [64]: per-track lmbda (ba.py:299-300), only where the caller passes one

    // An opaque copy of the lane id for code that runs once per tile: addresses formed from it are formed THERE (a shift and an
    // add) and die there; formed from `lane` they are hoisted out of the tile loop as loop invariants, spilled for want of
    // registers, and every reload is a scratch load — a wait for everything the prefetch has in flight.
    auto olane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    // (a wave without tiles — the launch's last workgroup may hold some — runs the prologue on tile 0, skips the loop, and
    //  takes part in the workgroup's barriers at the end with nothing to add)
    // Wave priority.  A SIMD holds two of these waves, and its arbiter serves the OLDER one first whenever both have an instruction
    // ready: measured (tools/gpu_wave_times.py, 8.4M edges) the first wave of a SIMD walks its 8 tiles in 83-96 us, the second in
    // 110-120 — and the kernel ends with the last wave.  So the two take turns: a wave learns from an arrival counter of its SIMD
    // whether it is the first or the second, and raises its priority (s_setprio) in every other slice of the CU's clock, the slices in
    // which its partner lowers it.
    int arrive_rank = 0;
    if (a.priv) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4) /* HW_ID: simd 5:4, cu 11:8, sh 12, se 15:13 */, xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) /* XCC_ID */;
        const unsigned idx = ((xcc & 7u) << 10) | (((hw >> 13) & 7u) << 7) | (((hw >> 12) & 1u) << 6) | (((hw >> 8) & 15u) << 2) | ((hw >> 4) & 3u);
        int *arr = reinterpret_cast<int *>(a.priv + priv_copy_doubles((size_t)pd.D, (size_t)pd.P));
        if (lane == 0) arrive_rank = atomicAdd(arr + idx, 1);
    }
    // The launch's first waves take the LAST tiles: they are the first on their SIMDs and stay a few per cent ahead even with the
    // priorities taking turns, and the tile list of a growing map ends with its newest frames — the ones whose windows are clipped
    // (repeated and self observations: the tiles that take longest; 132 -> 129 us at 8.4M edges of the benchmark generator,
    // profiles/r05_edge2_wave_times.txt).
#ifdef BT_E2_FORWARD      /* measurement */
    const int gwt = gw;
#else
    const int gwt = (int)(gridDim.x * nwv) - 1 - gw;
#endif
    const bool has_work = gwt * tiles_per_wave < pd.T;
    const int t_begin = has_work ? gwt * tiles_per_wave : 0, t_end = has_work ? min(pd.T, t_begin + tiles_per_wave) : 0;
    BT_PROBE_E2_DECL();

    // E Q E^T since the last flush lives in LDS, as float64 (schur_rows); E (Q w') of the current tile in float32 registers,
    // added to float64 sums in LDS at the tile's end
    // this wave's copies of y and of the per-pair sums (ba_plan.hpp: kPrivY)
    // (by workgroup: only the roots of the workgroups' trees issue atomics)
    double *ypriv = a.priv ? a.priv + (size_t)(blockIdx.x & (kPrivY - 1)) * pd.D : a.y;
    double *ppriv = a.priv ? a.priv + (size_t)kPrivY * pd.D + (size_t)(blockIdx.x & (kPrivP - 1)) * pd.P * kPairAccStride : a.pairacc;
    constexpr int NACC = NT * (NT + 1) / 2;
    float yacc[NT];                          // lane (li, kq) holds row 16 ti + li (all kq alike)
#pragma unroll
    for (int t = 0; t < 2 * NACC; ++t) reinterpret_cast<double2 *>(sacc)[t * 64 + lane] = make_double2(0.0, 0.0);
    ysum[lane] = 0.0;
#pragma unroll
    for (int t = 0; t < NT; ++t) yacc[t] = 0.0f;
    int acc_tiles = 0, Racc = 0;

    auto flush_schur = [&]() {
        if (acc_tiles == 0) return;
        const int ln = olane();
        // global rows of this lane's elements: the lane's column of every column tile and its four rows of every row tile
        // (f32 C/D layout: col = lane & 15, row = 4 * (lane >> 4) + reg), read once — one LDS round trip for the flush
        int gc[NT], gr[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = 16 * t + (ln & 15);
            gc[t] = col < Racc ? gidx[col] : -1;
            const int4 r4 = *reinterpret_cast<const int4 *>(gidx + min(16 * t + 4 * (ln >> 4), ((Racc + 3) & ~3) - 4));
            const int row = 16 * t + 4 * (ln >> 4);
            gr[t][0] = row < Racc ? r4.x : -1; gr[t][1] = row + 1 < Racc ? r4.y : -1;
            gr[t][2] = row + 2 < Racc ? r4.z : -1; gr[t][3] = row + 3 < Racc ? r4.w : -1;
        }
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj) {
                const int t = ti * (ti + 1) / 2 + tj;
                double2 *cp = reinterpret_cast<double2 *>(sacc) + t * 128 + ln;
                const double2 c01 = cp[0], c23 = cp[64];
                cp[0] = make_double2(0.0, 0.0); cp[64] = make_double2(0.0, 0.0);
                const double cr[4] = {c01.x, c01.y, c23.x, c23.y};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (gc[tj] >= 0 && gr[ti][r] >= gc[tj]) atomicAdd(&a.S[(size_t)gr[ti][r] * pd.D + gc[tj]], -cr[r]);
            }
        if (ln < Racc) {
            const double v = ysum[ln];
            ysum[ln] = 0.0;
            atomicAdd(&ypriv[gidx[ln]], -v);
        }
        acc_tiles = 0;
    };

    // per-pair sums Bjj (21, row-major upper triangle, structural zero at [0][1] skipped) and gj (6) (ba.py:260,266) of this
    // lane's slot, both edges of the lane side by side
    f2 pa[26];
#pragma unroll
    for (int i = 0; i < 26; ++i) pa[i] = f2{0.0f, 0.0f};
    int pa_tiles = 0, pa_lgS = 0, pa_gp = -1;          // tiles summed into pa, their slots per track, the global pair of lane's slot
    auto flush_pairs = [&]() {
        if (pa_tiles == 0) return;
        float f[26];
#pragma unroll
        for (int i = 0; i < 26; ++i) { f[i] = pa[i].x + pa[i].y; pa[i] = f2{0.0f, 0.0f}; }
        stride_sum(f, pa_lgS);
        if (olane() < (1 << pa_lgS) && pa_gp >= 0) {
            double *dst = ppriv + (size_t)pa_gp * kPairAccStride;
#pragma unroll
            for (int i = 0; i < 26; ++i) {
                const int vi = i < 1 ? 0 : i + 1;        // element order of the 27-vector: the zero at [0][1] stays
                atomicAdd(dst + vi, (double)f[i]);
            }
        }
        pa_tiles = 0;
    };

    {   // columns of E a step does not hold (fewer than 16 tracks per step) are never written: zero once
        float4 *z = reinterpret_cast<float4 *>(Eh);
        for (int i = lane; i < (Rmax * kRow) / 4; i += 64) z[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (lane < kQn) { Qs[lane] = 0.0f; Bs[lane] = 0.0f; }
    }

    // ---- the wave's stream of steps: the planner gives every tile an even number of 64-edge iterations (ba_plan.cpp:
    // em_iterations), so step J of the stream is the iterations (2J, 2J + 1) of it_edge — both inside one tile, one edge of each
    // per lane — and the prefetch runs over it with no knowledge of the tiles.  The operands of a step (targets, weights of its
    // two edges) live in one of TWO register sets, by the step's parity: a step consumes its set in its first third and the
    // gathers of the step after next are issued into the same registers right there — no register is copied and no load is
    // waited for before its step (a rotating queue has to MOVE the youngest load's destination at the end of every step, which
    // is a wait for everything in flight).  Edge ids of missing edges (-1) are clamped to edge 0 for the gathers (branch-free)
    // and kept as two flag bits.
    Rec rec = load_rec(pd, t_begin);
    Rec rec_n = t_begin + 1 < t_end ? load_rec(pd, t_begin + 1) : rec;
    int gj = rec.it0 >> 1, gj_end;
    { const Rec last = load_rec(pd, max(t_end - 1, 0)); gj_end = (last.it0 + last.nit + 1) >> 1; }
    auto load_ids = [&](int j, int &ea, int &eb) {             // (beyond the wave's last step: that step's ids again, never used)
        const unsigned f = (unsigned)min(j, gj_end - 1) * 2u * kLanes + (unsigned)lane;
        ea = pd.it_edge[f]; eb = pd.it_edge[f + kLanes];
    };
    // gather through the ids (ea, eb), then replace them by the ids of step jn.  The id loads are ISSUED FIRST (the counter of
    // outstanding loads retires in order: whoever waits for ids issued behind the gathers waits for the gathers as well), after
    // the addresses of the gathers have been formed from the old ids.
    auto gather = [&](int &ea, int &eb, int jn, f2 &tu, f2 &tv, f2 &w0, f2 &w1, int &fl) {
        const unsigned ua = (unsigned)max(ea, 0), ub = (unsigned)max(eb, 0);
        const unsigned ta = ua * (unsigned)a.tstride, tb = ub * (unsigned)a.tstride;     // (launch_edge2 checks that byte offsets fit 32 bits)
        const float *pa_ = a.targets + ta, *pb_ = a.targets + tb;
        const float2 *wa_ = reinterpret_cast<const float2 *>(a.weights) + ua, *wb_ = reinterpret_cast<const float2 *>(a.weights) + ub;
        fl = (ea >= 0 ? 1 : 0) | (eb >= 0 ? 2 : 0);
        asm volatile("" : "+v"(fl));         // (formed NOW: left symbolic, the compiler keeps the old ids alive for two steps instead)
        BT_E2_SB;
        load_ids(jn, ea, eb);
        BT_E2_SB;
        tu.x = pa_[0]; tv.x = pa_[1];
        tu.y = pb_[0]; tv.y = pb_[1];
        const float2 wa = *wa_, wb = *wb_;
        w0 = f2{wa.x, wb.x}; w1 = f2{wa.y, wb.y};
    };
    // register sets by the parity of the step: operands, flags, and the ids of the step TWO steps on (same parity)
    f2 tu_q[2], tv_q[2], w0_q[2], w1_q[2];
    int fl_q[2], ea_q[2], eb_q[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        load_ids(gj + k, ea_q[k], eb_q[k]);
        gather(ea_q[k], eb_q[k], gj + k + 2, tu_q[k], tv_q[k], w0_q[k], w1_q[k], fl_q[k]);
    }
    int par = 0;                                               // parity of the stream's next step = its register set

    // ---- per-tile context of the first tile (lane = track of the tile)
    int kx_c = pd.tile_kx[(unsigned)t_begin * kLanes + (unsigned)lane];
    unsigned la_c = pd.tile_la[(unsigned)t_begin * kLanes + (unsigned)lane];
    unsigned si_c = pd.tile_sinfo[(unsigned)t_begin * kLanes + (unsigned)(lane & ((1 << rec.lgS) - 1))];
    float px, py, pdisp, mono_v, lm_v = a.lmbda;
    {
        const unsigned kq = (unsigned)max(kx_c, 0);
        px = a.patches[3u * kq]; py = a.patches[3u * kq + 1u]; pdisp = a.patches[3u * kq + 2u];
        mono_v = a.mono[kq * (unsigned)a.mstride];
        if (a.lmbda_trk) lm_v = a.lmbda_trk[(unsigned)pd.trk_off + (unsigned)min(rec.trk0 + lane, pd.m - 1)];
    }
    unsigned si_prev = 0xffffffffu;
    int e_lgS = LGS >= 0 ? LGS : -1;                           // slots per track of the tiles that wrote the local E last
    const double b0 = (double)a.b0, b1 = (double)a.b1, b2 = (double)a.b2, b3 = (double)a.b3;

    const int rank_s = __builtin_amdgcn_readfirstlane(arrive_rank) & 1;      // (the counter's answer has long arrived)
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int flags = tile == t_begin ? 0 : rec.flags;
        const int R = 6 * rec.ncam;
        const bool has_next = tile + 1 < t_end;
        const int lgS = LGS >= 0 ? LGS : rec.lgS, S1 = (1 << lgS) - 1, G = kLanes >> lgS;
        const unsigned si = si_c;
        const unsigned lb = si & 0xffu, lp = (si >> 8) & 0xffu;
        const bool rep = (si >> 16) & 1u, used = (si >> 17) & 1u;
        BT_E2_PF(10);
        // ---- new cameras / new pair list / another pair in some lane's slot
        if (!(flags & 1) || acc_tiles >= kSchurFlushTiles) {
            flush_schur();
            const int *cams = pd.tile_cams + rec.cam0;
            for (int i = olane(); i < R; i += 64) gidx[i] = 6 * cams[i / 6] + i % 6;
        }
        const bool same_slots = (flags & 2) && lgS == pa_lgS && __builtin_amdgcn_ballot_w64((si ^ si_prev) & 0x2ff00u) == 0;
        if (!same_slots || pa_tiles >= kPaFlushTiles) flush_pairs();
        if (!same_slots) {
            pa_lgS = lgS;
            pa_gp = used ? pd.tile_pairs[rec.pair0 + (int)lp] : -1;
        }
        si_prev = si;
        BT_E2_PF(11);
        if (!(flags & 2)) {
            for (int p = olane(); p < rec.npair; p += 64) {
                const int gp = pd.tile_pairs[rec.pair0 + p];
                const int ij = pd.tile_ij[(unsigned)tile * (unsigned)mtp + (unsigned)p];
                double g[kPairGeomFloats];
                pair_geometry<double, true>(a.poses, a.intr, ij & 0xffff, ij >> 16, g);
                double *gd = geoD + p * kGeoDS;
                // R, t ROUNDED to float32 here as well: the Jacobians below and the Ad sandwich of k_pair_finalize use the float32
                // geometry, and a residual taken at the exact geometry with Jacobians at the rounded one is an INCONSISTENT
                // system — the same 6e-8 per pair, coherent over the pair's thousands of edges, in J but not in r: y = J^T W r
                // then errs along directions cond(S) amplifies (measured: S, y 3e-8 from the oracle and dX 5e-5; with both at the
                // rounded geometry the step is the exact step of a problem 6e-8 away: S, y 2e-7 and dX 2e-6 .. 7e-6)
#pragma unroll
                for (int c = 0; c < 12; ++c) gd[c] = (double)(float)g[c];
#pragma unroll
                for (int c = 0; c < 4; ++c) gd[12 + c] = g[16 + c];
                float *gf = geoF + p * kGeoFS;
                gf[0] = (float)g[9]; gf[1] = (float)g[10]; gf[2] = (float)g[11]; gf[3] = (float)g[16];
#pragma unroll
                for (int c = 0; c < 9; ++c) gf[4 + c] = (float)g[c];
                gf[13] = (float)g[17]; gf[14] = 0.0f; gf[15] = 0.0f;
                // what k_pair_finalize and the step's last kernel read (ba_edge.hpp: pair_geometry<float, true>)
                float4 *dst = reinterpret_cast<float4 *>(a.pairgeo + (size_t)gp * kPairGeomFloats);
#pragma unroll
                for (int c = 0; c < kPairGeomFloats / 4; ++c)
                    dst[c] = make_float4((float)g[4 * c], (float)g[4 * c + 1], (float)g[4 * c + 2], (float)g[4 * c + 3]);
                if (p == 0) { srcK[0] = frcp(g[12]); srcK[1] = frcp(g[13]); srcK[2] = g[14]; srcK[3] = g[15]; }
            }
            BT_E2_WAVE_SYNC();
        }
        // the tracks' normalised source coordinates (projective_ops.py:19-29), disparities, priors where every lane can read them
        {
            const double X0 = ((double)px - srcK[2]) * srcK[0], Y0 = ((double)py - srcK[3]) * srcK[1];
            const int ln = olane();
            reinterpret_cast<double2 *>(ptD)[ln] = make_double2(X0, Y0);
            reinterpret_cast<float2 *>(ptF)[ln] = make_float2(pdisp, mono_v);
            if (a.lmbda_trk) ptL[ln] = lm_v;
        }
        if (LGS < 0 && lgS != e_lgS) {
            // a step of this tile holds another number of tracks: the columns beyond them must not keep the last tile's values
            float4 *z = reinterpret_cast<float4 *>(Eh);
            for (int i = olane(); i < (Rmax * kRow) / 4; i += 64) z[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            e_lgS = lgS;
        }
        const bool any_rep = __builtin_amdgcn_ballot_w64(rep && lane <= S1) != 0;
        // Repeated observations (several slots of a track with the same target camera: the slots are sorted by pair, so they are
        // neighbours): their E entries are added up across the lanes — a segmented suffix sum over the run, distances 1, 2, 4, 8 —
        // and the run's first lane stores the sum.  (LDS float atomics, what round 4 used, cost ~250 cycles per wave instruction:
        // the tiles of the benchmark graphs' border frames, whose clipped targets repeat, took twice as long as the others.)
        // (seg_dist: the largest of the distances 1, 2, 4, 8 at which some lane finds a lane of its run; 0: the repeats are all of
        //  fixed cameras, nothing to add up)
        int seg_dist = 0;
        if (any_rep && lgS <= 4) {
            const int slot = olane() & S1;
            const int key = used && lb != 0xffu ? (int)lb : -1 - slot;
            // (the lane moves first, with every lane active: a DPP read of a lane that a short-circuited condition has switched off
            //  does not return that lane's value)
            const int k1 = __builtin_amdgcn_update_dpp(-999, key, 0x101, 0xf, 0xf, false), k2 = __builtin_amdgcn_update_dpp(-999, key, 0x102, 0xf, 0xf, false);
            const int k4 = __builtin_amdgcn_update_dpp(-999, key, 0x104, 0xf, 0xf, false), k8 = __builtin_amdgcn_update_dpp(-999, key, 0x108, 0xf, 0xf, false);
            const bool h1 = (slot + 1 <= S1) & (k1 == key), h2 = (slot + 2 <= S1) & (k2 == key), h4 = (slot + 4 <= S1) & (k4 == key), h8 = (slot + 8 <= S1) & (k8 == key);
            seg_dist = __builtin_amdgcn_ballot_w64(h8) ? 8 : __builtin_amdgcn_ballot_w64(h4) ? 4 : __builtin_amdgcn_ballot_w64(h2) ? 2 : __builtin_amdgcn_ballot_w64(h1) ? 1 : 0;
        }
        const bool seg = seg_dist > 0, rep_add = any_rep && lgS > 4;
        const unsigned la = __builtin_amdgcn_readfirstlane(la_c);          // the tile's tracks share their source camera
        const bool self_tile = la != 0xffu && __builtin_amdgcn_ballot_w64(used && lane <= S1 && lb == la) != 0;   // some slot's target IS the source (ii == jj)
        const bool hasB = G < 64;                                          // (S = 1: a step is the tile's 64 tracks, no second half)
        const int tl = lane >> lgS;
        const bool lead = (lane & S1) == 0;
        const int ntrk = rec.ntrk, trk0 = rec.trk0;
        const int nv = (min(2 * G, kQn) + 15) >> 4;                        // k-sweeps of 16 tracks per step
        const int ntl = (R + 15) >> 4;                                     // row tiles of the tile's E
        BT_E2_WAVE_SYNC();
        BT_E2_PF(12);
        // ---- next tile's context: requested now, lands under this tile's steps
        int kx_n = -1;
        unsigned la_n = 0xffu, si_n = 0u;
        float px_n = 0.0f, py_n = 0.0f, pd_n = 0.0f, mono_n = 0.0f, lm_n = a.lmbda;
        RawRec raw_nn = load_raw(pd, min(tile + 2, t_end - 1));
        if (has_next) {
            kx_n = pd.tile_kx[(unsigned)(tile + 1) * kLanes + (unsigned)lane];
            la_n = pd.tile_la[(unsigned)(tile + 1) * kLanes + (unsigned)lane];
            si_n = pd.tile_sinfo[(unsigned)(tile + 1) * kLanes + (unsigned)(lane & ((1 << rec_n.lgS) - 1))];
            if (a.lmbda_trk) lm_n = a.lmbda_trk[(unsigned)pd.trk_off + (unsigned)min(rec_n.trk0 + lane, pd.m - 1)];
        }

        const int nit2 = (rec.nit + 1) >> 1;
        BT_E2_PF(0);
        auto step = [&](auto pc, int it) {
            constexpr int P = decltype(pc)::value;
#ifndef BT_E2_NO_PRIO
            // (see "wave priority" at the top; slices of the CU's clock, so that exactly one of a SIMD's two waves is raised at any time)
            // (BT_E2_PRIO_YOUNG eighths of the time to the wave that arrived second on its SIMD, the rest to the first)
            {
                const unsigned ph = ((unsigned)clock64() >> (BT_E2_PRIO_SHIFT - 2)) & 7u;
                if (rank_s ? ph < BT_E2_PRIO_YOUNG : ph >= BT_E2_PRIO_YOUNG) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            }
#endif
            const f2 tu = tu_q[P], tv = tv_q[P];
            const int fl = fl_q[P];
            const int trA = it * 2 * G + tl, trB = (trA + G) & 63;
            const double2 xyA = reinterpret_cast<const double2 *>(ptD)[trA], xyB = reinterpret_cast<const double2 *>(ptD)[trB];
            const float2 tkA = reinterpret_cast<const float2 *>(ptF)[trA], tkB = reinterpret_cast<const float2 *>(ptF)[trB];
            const f2 d = f2{tkA.x, tkB.x};
            double gd[kGeoD];
            {
                const double2 *g2 = reinterpret_cast<const double2 *>(geoD + (size_t)lp * kGeoDS);
#pragma unroll
                for (int c = 0; c < kGeoD / 2; ++c) { const double2 t2 = g2[c]; gd[2 * c] = t2.x; gd[2 * c + 1] = t2.y; }
            }
            const Proj pA = project(gd, xyA.x, xyA.y, d.x, tu.x, tv.x, (fl & 1) != 0, b0, b1, b2, b3);
            const Proj pB = project(gd, xyB.x, xyB.y, d.y, tu.y, tv.y, hasB && (fl & 2) != 0, b0, b1, b2, b3);
            BT_E2_PF(1);
            float gf[kGeoF];
            {
                const float4 *g4 = reinterpret_cast<const float4 *>(geoF + (size_t)lp * kGeoFS);
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float4 t4 = g4[c]; gf[4 * c] = t4.x; gf[4 * c + 1] = t4.y; gf[4 * c + 2] = t4.z; gf[4 * c + 3] = t4.w; }
            }
            // ---- Jacobians, robust weights and products for the two edges at once (projective_ops.py:80-98, ba.py:247-266)
            const f2 X = f2{pA.X, pB.X}, Y = f2{pA.Y, pB.Y}, Z = f2{pA.Z, pB.Z};
            const f2 vld = f2{pA.ok ? 1.0f : 0.0f, pB.ok ? 1.0f : 0.0f};
            const f2 dj = f2{fabsf(Z.x) > 0.2f ? frcp(Z.x) : 0.0f, fabsf(Z.y) > 0.2f ? frcp(Z.y) : 0.0f};
            const f2 t0 = splat(gf[0]), t1 = splat(gf[1]), t2 = splat(gf[2]);
            const f2 A = splat(gf[3]) * dj, C = splat(gf[13]) * dj;
            const f2 Bc = -(A * (X * dj)), Dc = -(C * (Y * dj));
            const f2 a0 = d * A, a2 = d * Bc, a3 = Bc * Y, a4 = fma2(A, Z, -(Bc * X)), a5 = -(A * Y);
            const f2 b1_ = d * C, b2_ = d * Dc, b3_ = fma2(Dc, Y, -(C * Z)), b4_ = -(Dc * X), b5_ = C * X;
            const f2 jz0 = fma2(A, t0, Bc * t2), jz1 = fma2(C, t1, Dc * t2);
            const f2 r0u = f2{pA.r0, pB.r0}, r1u = f2{pA.r1, pB.r1};
            const f2 s0 = r0u * r0u, s1 = r1u * r1u;
            const f2 rw0 = f2{robust1<LOSS>(s0.x), robust1<LOSS>(s0.y)};
            const f2 rw1 = f2{robust1<LOSS>(s1.x), robust1<LOSS>(s1.y)};
            const f2 W0 = vld * (w0_q[P] * rw0), W1 = vld * (w1_q[P] * rw1);
            // ---- this step's operands are consumed: the gathers of the step after next go into the same registers, through the
            // ids loaded two steps ago, and the ids of the step two steps behind that one are requested
            BT_E2_PF(2);
            BT_E2_SB;
            // The next tile's context arrives through loads issued a variable number of steps ago, and the compiler's count of
            // loads in flight does not survive a loop of variable length: whoever touches such a value first waits for EVERY
            // load in flight.  So they are touched HERE, where that wait is cheapest — the youngest loads in flight are the
            // previous step's gathers, a whole step old: the patch indices in the tile's first step (to request the patches),
            // everything else in its last step.  At the tile's top nothing is left to wait for.
            if (it == 0 && has_next) {
                asm volatile("" : "+v"(kx_n));
                const unsigned kq = (unsigned)max(kx_n, 0);
                px_n = a.patches[3u * kq]; py_n = a.patches[3u * kq + 1u]; pd_n = a.patches[3u * kq + 2u];
                mono_n = a.mono[kq * (unsigned)a.mstride];
            }
            if (it == nit2 - 1 && has_next) {
                asm volatile("" : "+v"(px_n), "+v"(py_n), "+v"(pd_n), "+v"(mono_n), "+v"(lm_n), "+v"(la_n), "+v"(si_n));
                asm volatile("" : "+v"(raw_nn.r0.x), "+v"(raw_nn.r0.w), "+v"(raw_nn.r1.x), "+v"(raw_nn.r1.y), "+v"(raw_nn.r1.z), "+v"(raw_nn.r1.w));
            }
            gather(ea_q[P], eb_q[P], gj + 4, tu_q[P], tv_q[P], w0_q[P], w1_q[P], fl_q[P]);
            BT_E2_SB;
            BT_E2_PF(3);
            const f2 r0 = vld * r0u, r1 = vld * r1u;
            const f2 wa0 = W0 * a0, wa2 = W0 * a2, wa3 = W0 * a3, wa4 = W0 * a4, wa5 = W0 * a5;
            const f2 wb1 = W1 * b1_, wb2 = W1 * b2_, wb3 = W1 * b3_, wb4 = W1 * b4_, wb5 = W1 * b5_;
            // Ej = Jj^T W Jz (ba.py:263)
            f2 Ej[6] = { wa0 * jz0, wb1 * jz1, fma2(wa2, jz0, wb2 * jz1), fma2(wa3, jz0, wb3 * jz1),
                               fma2(wa4, jz0, wb4 * jz1), fma2(wa5, jz0, wb5 * jz1) };
            // the tracks' sums over their S lanes: C, w (ba.py:287,292) and the source-camera E, Ei = -Ad^T Ej
            const f2 wj0 = W0 * jz0, wj1 = W1 * jz1;
            f2 sv[8];
            sv[0] = fma2(wj0, jz0, wj1 * jz1);
            sv[1] = fma2(wj0, r0, wj1 * r1);
            if (la != 0xffu) {
                // o_tau = R^T e_tau ; o_phi = R^T (e_tau x t + e_phi)      (se3.h:58-67)
                const f2 cx = fma2(Ej[1], t2, fma2(-Ej[2], t1, Ej[3]));
                const f2 cy = fma2(Ej[2], t0, fma2(-Ej[0], t2, Ej[4]));
                const f2 cz = fma2(Ej[0], t1, fma2(-Ej[1], t0, Ej[5]));
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f2 Rc0 = splat(gf[4 + c]), Rc1 = splat(gf[7 + c]), Rc2 = splat(gf[10 + c]);
                    sv[2 + c] = -fma2(Rc0, Ej[0], fma2(Rc1, Ej[1], Rc2 * Ej[2]));
                    sv[5 + c] = -fma2(Rc0, cx, fma2(Rc1, cy, Rc2 * cz));
                }
                if (self_tile) {
                    // a self edge (ii == jj): its target rows ARE the track's source rows — it joins the sum instead of being
                    // stored (a store here and a read-add-write by the track's first lane were six LDS round trips in a row)
                    const f2 sm = splat(used && lb == la ? 1.0f : 0.0f);
#pragma unroll
                    for (int c = 0; c < 6; ++c) sv[2 + c] = fma2(sm, Ej[c], sv[2 + c]);
                }
                group_sum2(sv, lgS);
            } else {
                f2 s2[2] = {sv[0], sv[1]};
                group_sum2(s2, lgS);
                sv[0] = s2[0]; sv[1] = s2[1];
#pragma unroll
                for (int c = 2; c < 8; ++c) sv[c] = f2{0.0f, 0.0f};
            }
            BT_E2_PF(4);
            // ---- the step's E.  Column of a track = its index in the step (first halves 0 .. G-1, second halves G .. 2G-1).
            // Target-camera rows: one lane per (track, camera) — a plain store, zeros where there is no edge; where the plan marks
            // repeated observations the run's first lane stores the run's sum (`seg`, at the tile's top)
            if (seg) {
                // bit d: the lane 2^d above belongs to this lane's run; bit 4: the lane is its run's first (formed here, per step, in
                // the tiles that need it: a register held across the tile costs the others a spill)
                const int slot = lane & S1;
                const int key = used && lb != 0xffu ? (int)lb : -1 - slot;                      // (no two lanes without a camera match)
                const int k1 = __builtin_amdgcn_update_dpp(-999, key, 0x101, 0xf, 0xf, false), k2 = __builtin_amdgcn_update_dpp(-999, key, 0x102, 0xf, 0xf, false);
                const int k4 = __builtin_amdgcn_update_dpp(-999, key, 0x104, 0xf, 0xf, false), k8 = __builtin_amdgcn_update_dpp(-999, key, 0x108, 0xf, 0xf, false);
                const int kp = __builtin_amdgcn_update_dpp(-999, key, 0x111, 0xf, 0xf, false);   // row_shr:1: the lane below
                const unsigned seg_bits = (slot + 1 <= S1 && k1 == key ? 1u : 0u) | (slot + 2 <= S1 && k2 == key ? 2u : 0u) | (slot + 4 <= S1 && k4 == key ? 4u : 0u) |
                                          (slot + 8 <= S1 && k8 == key ? 8u : 0u) | (slot == 0 || kp != key ? 16u : 0u);
                auto shl = [&](f2 v, auto ctrl) {
                    return f2{__uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v.x), decltype(ctrl)::value, 0xf, 0xf, true)),
                              __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v.y), decltype(ctrl)::value, 0xf, 0xf, true))};
                };
#pragma unroll
                for (int c = 0; c < 6; ++c) Ej[c] = fma2(splat(seg_bits & 1u ? 1.0f : 0.0f), shl(Ej[c], IC<0x101>()), Ej[c]);
                if (seg_dist > 1) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) Ej[c] = fma2(splat(seg_bits & 2u ? 1.0f : 0.0f), shl(Ej[c], IC<0x102>()), Ej[c]);
                }
                if (seg_dist > 2) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) Ej[c] = fma2(splat(seg_bits & 4u ? 1.0f : 0.0f), shl(Ej[c], IC<0x104>()), Ej[c]);
                }
                if (seg_dist > 4) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) Ej[c] = fma2(splat(seg_bits & 8u ? 1.0f : 0.0f), shl(Ej[c], IC<0x108>()), Ej[c]);
                }
                if (used && lb != 0xffu && lb != la && (seg_bits & 16u)) {
                    float *row = Eh + lb * 6 * kRow + tl;
#pragma unroll
                    for (int c = 0; c < 6; ++c) { row[c * kRow] = Ej[c].x; if (hasB) row[c * kRow + G] = Ej[c].y; }
                }
            } else {
                if (rep_add) {                      // (more than 16 slots per track: the step's E is cleared first and the repeats add)
                    float4 *z = reinterpret_cast<float4 *>(Eh);
                    for (int i = lane; i < (R * kRow + 3) / 4; i += 64) z[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
                if (used && lb != 0xffu && lb != la) {
                    float *row = Eh + lb * 6 * kRow + tl;
                    if (rep && rep_add) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) { atomicAdd(row + c * kRow, Ej[c].x); if (hasB) atomicAdd(row + c * kRow + G, Ej[c].y); }
                    } else {
#pragma unroll
                        for (int c = 0; c < 6; ++c) { row[c * kRow] = Ej[c].x; if (hasB) row[c * kRow + G] = Ej[c].y; }
                    }
                }
            }
            if (lead) {
                if (la != 0xffu) {
                    float *row = Eh + la * 6 * kRow + tl;          // a track's source-camera row is written here and nowhere else
#pragma unroll
                    for (int c = 0; c < 6; ++c) { row[c * kRow] = sv[2 + c].x; if (hasB) row[c * kRow + G] = sv[2 + c].y; }
                }
                // Q = 1 / (C + pi alpha + lmbda), w' = w - pi alpha (d - d_mono) (ba.py:296-311) of the lane's two tracks;
                // tracks beyond the tile's last have no edge (E column 0): Q = 0 keeps their products finite
                const f2 mono = f2{tkA.y, tkB.y}, lmb = a.lmbda_trk ? f2{ptL[trA], ptL[trB]} : splat(a.lmbda);
                const f2 pm = f2{mono.x > 1e-2f ? a.alpha : 0.0f, mono.y > 1e-2f ? a.alpha : 0.0f};
                const f2 Ca = sv[0] + pm + lmb;
                const f2 wp = sv[1] - pm * (d - mono);
                const int gA = it * 2 * G + tl, gB = gA + G;                 // (unwrapped: a second half beyond the tile has no track)
                const f2 Qv = f2{gA < ntrk ? rcp_f32(Ca.x) : 0.0f, hasB && gB < ntrk ? rcp_f32(Ca.y) : 0.0f};
                const f2 be = Qv * wp;
                Qs[tl] = Qv.x; Bs[tl] = be.x;
                if (hasB) { Qs[tl + G] = Qv.y; Bs[tl + G] = be.y; }
                if (gA < ntrk) a.qw[(unsigned)trk0 + (unsigned)gA] = make_float2(Qv.x, wp.x);
                if (hasB && gB < ntrk) a.qw[(unsigned)trk0 + (unsigned)gB] = make_float2(Qv.y, wp.y);
            }
            BT_E2_PF(5);
            auto pair_sums = [&]() {
                // per-pair sums, the lane's two edges side by side
                pa[0] = fma2(wa0, a0, pa[0]);   pa[1] = fma2(wa0, a2, pa[1]);   pa[2] = fma2(wa0, a3, pa[2]);
                pa[3] = fma2(wa0, a4, pa[3]);   pa[4] = fma2(wa0, a5, pa[4]);
                pa[5] = fma2(wb1, b1_, pa[5]);  pa[6] = fma2(wb1, b2_, pa[6]);  pa[7] = fma2(wb1, b3_, pa[7]);
                pa[8] = fma2(wb1, b4_, pa[8]);  pa[9] = fma2(wb1, b5_, pa[9]);
                pa[10] = fma2(wa2, a2, fma2(wb2, b2_, pa[10])); pa[11] = fma2(wa2, a3, fma2(wb2, b3_, pa[11]));
                pa[12] = fma2(wa2, a4, fma2(wb2, b4_, pa[12])); pa[13] = fma2(wa2, a5, fma2(wb2, b5_, pa[13]));
                pa[14] = fma2(wa3, a3, fma2(wb3, b3_, pa[14])); pa[15] = fma2(wa3, a4, fma2(wb3, b4_, pa[15]));
                pa[16] = fma2(wa3, a5, fma2(wb3, b5_, pa[16]));
                pa[17] = fma2(wa4, a4, fma2(wb4, b4_, pa[17])); pa[18] = fma2(wa4, a5, fma2(wb4, b5_, pa[18]));
                pa[19] = fma2(wa5, a5, fma2(wb5, b5_, pa[19]));
                pa[20] = fma2(wa0, r0, pa[20]); pa[21] = fma2(wb1, r1, pa[21]);
                pa[22] = fma2(wa2, r0, fma2(wb2, r1, pa[22])); pa[23] = fma2(wa3, r0, fma2(wb3, r1, pa[23]));
                pa[24] = fma2(wa4, r0, fma2(wb4, r1, pa[24])); pa[25] = fma2(wa5, r0, fma2(wb5, r1, pa[25]));
            };
            // (Placing these 42 products between the matrix products of schur_rows — a wave that issues 24 of those back to back
            //  waits ~30 cycles at each — keeps their 44 operand registers alive across the E stores: measured in the compiler's
            //  report as 264 bytes of scratch per lane.  They stay here.)
            pair_sums();
            BT_E2_PF(6);
            // ---- E Q E^T and E (Q w') of the step's tracks (schur_rows)
            BT_E2_WAVE_SYNC();
            if (ntl == NT) schur_rows<NT, NT, kQn, kRow>(Eh, Qs, Bs, sacc, yacc, R, nv, lane, [] {});
            else if (NT > 1 && ntl == NT - 1) schur_rows<(NT > 1 ? NT - 1 : 1), NT, kQn, kRow>(Eh, Qs, Bs, sacc, yacc, R, nv, lane, [] {});
            else if (NT > 2 && ntl == NT - 2) schur_rows<(NT > 2 ? NT - 2 : 1), NT, kQn, kRow>(Eh, Qs, Bs, sacc, yacc, R, nv, lane, [] {});
            else schur_rows<1, NT, kQn, kRow>(Eh, Qs, Bs, sacc, yacc, R, nv, lane, [] {});
            BT_E2_WAVE_SYNC();               // (the next step's stores into E stay behind these reads)
            BT_E2_PF(7);
            ++gj;
        };
        {
            int it = 0;
            if (par) { step(IC<1>{}, it); ++it; }
#pragma unroll 1
            for (; it + 1 < nit2; it += 2) { step(IC<0>{}, it); step(IC<1>{}, it + 1); }
            if (it < nit2) { step(IC<0>{}, it); par = 1; } else par = 0;
        }
        ++pa_tiles;
        ++acc_tiles; Racc = R;
        {   // the tile's E (Q w') to the float64 sums (the lanes of the first group hold the rows' totals)
            const int ln = olane();
            double ys[NT];
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) ys[ti] = ysum[min(16 * ti + (ln & 15), 63)];
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                if (ln < 16 && 16 * ti + ln < R) ysum[16 * ti + ln] = ys[ti] + (double)yacc[ti];
                yacc[ti] = 0.0f;
            }
        }

        // ---- rotate the tile context
        if (has_next) {
            rec = rec_n; rec_n = decode_rec(raw_nn); kx_c = kx_n; la_c = la_n; si_c = si_n;
            px = px_n; py = py_n; pdisp = pd_n; mono_v = mono_n; lm_v = lm_n;
        }
    }
    BT_PROBE_E2_TILES_DONE();
    // ---- the end: the workgroup's waves add up what they hold, then the roots of the tree issue the atomics (see the top)
    {
        const int S_p = 1 << pa_lgS;
        const bool pvalid = pa_tiles > 0;
        float f[26];
#pragma unroll
        for (int i = 0; i < 26; ++i) f[i] = pa[i].x + pa[i].y;
        if (pvalid) stride_sum(f, pa_lgS);
        // pair sums through LDS where the S lanes' 26 doubles fit the local E's place (it is dead now)
        const bool pcomb = pvalid && (size_t)S_p * 26 * sizeof(double) <= (size_t)Rmax * kRow * sizeof(float) && (size_t)S_p <= (size_t)2 * kQn;
        int *hdr = reinterpret_cast<int *>(srcK);                       // [0] rows of the Schur sums (0: none)  [1] lg S + 1 of the pair sums (0: none)  [2], [3] taken by a partner
        int *psig = reinterpret_cast<int *>(Qs);                        // the lanes' global pairs
        double *pbuf = reinterpret_cast<double *>(Eh);                  // [S][26]
        const int ln = olane();
        if (ln == 0) { hdr[0] = acc_tiles > 0 ? Racc : 0; hdr[1] = pcomb ? pa_lgS + 1 : 0; hdr[2] = 0; hdr[3] = 0; }
        if (pcomb && ln < S_p) {
            psig[ln] = pa_gp;
#pragma unroll
            for (int i = 0; i < 26; ++i) pbuf[ln * 26 + i] = (double)f[i];
        }
        __syncthreads();
        for (int stride = 1; stride < nwv; stride <<= 1) {
            if ((wv & (2 * stride - 1)) == 0 && wv + stride < nwv) {
                const ptrdiff_t po = (ptrdiff_t)stride * wave_doubles;     // the partner's slice of LDS, in doubles
                const int *phdr = reinterpret_cast<const int *>(srcK + po);
                int *phdr_w = reinterpret_cast<int *>(srcK + po);
                const int myR = hdr[0], myP = hdr[1];
                if (myR != 0 && phdr[0] == myR && phdr[2] == 0) {
                    const int *pgidx = reinterpret_cast<const int *>(reinterpret_cast<const double *>(gidx) + po);
                    const bool same = __builtin_amdgcn_ballot_w64(ln < myR && gidx[min(ln, myR - 1)] != pgidx[min(ln, myR - 1)]) == 0;
                    if (same) {
                        double2 *m = reinterpret_cast<double2 *>(sacc) + ln;
                        const double2 *q = reinterpret_cast<const double2 *>(sacc + po) + ln;
#pragma unroll
                        for (int t = 0; t < 2 * NACC; ++t) { double2 x = m[t * 64]; const double2 y2 = q[t * 64]; x.x += y2.x; x.y += y2.y; m[t * 64] = x; }
                        ysum[ln] += (ysum + po)[ln];
                        if (ln == 0) phdr_w[2] = 1;
                    }
                }
                if (myP != 0 && phdr[1] == myP && phdr[3] == 0) {
                    const int *ppsig = reinterpret_cast<const int *>(reinterpret_cast<const double *>(psig) + po);
                    const int Sq = 1 << (myP - 1);
                    const bool same = __builtin_amdgcn_ballot_w64(ln < Sq && psig[min(ln, Sq - 1)] != ppsig[min(ln, Sq - 1)]) == 0;
                    if (same) {
                        const double *q = pbuf + po;
                        for (int i = ln; i < Sq * 26; i += 64) pbuf[i] += q[i];
                        if (ln == 0) phdr_w[3] = 1;
                    }
                }
            }
            __syncthreads();
        }
        if (hdr[2] == 0) flush_schur();
        if (pvalid && (!pcomb || hdr[3] == 0) && ln < S_p && pa_gp >= 0) {
            double *dst = ppriv + (size_t)pa_gp * kPairAccStride;
#pragma unroll
            for (int i = 0; i < 26; ++i) {
                const int vi = i < 1 ? 0 : i + 1;
                atomicAdd(dst + vi, pcomb ? pbuf[ln * 26 + i] : (double)f[i]);
            }
        }
    }
    BT_PROBE_E2_END(gw, (int)(gridDim.x * nwv));
}
#undef BT_DPPF

static size_t lds_bytes(const PlanDev &pd, int lgs, bool with_lmbda_trk) {
    const size_t mtp = (size_t)(pd.max_tile_pairs > 0 ? pd.max_tile_pairs : 1);
    const size_t Rmax = (size_t)(6 * pd.max_cams);
    const size_t row = (size_t)(lgs >= 0 ? e_row(lgs) : e_row(0));
    const size_t nt = pd.max_cams <= 8 ? 3 : 4;
    return (128 + 4 + 64 + nt * (nt + 1) / 2 * 256 + mtp * kGeoDS) * sizeof(double) +
           (128 + 2 * (row - 4) + Rmax * row + ((Rmax + 3) & ~(size_t)3) + mtp * kGeoFS + (with_lmbda_trk ? 64 : 0)) * sizeof(float);
}

template <int NT, int LGS, int LOSS>
static int launch_t(const PlanDev &pd, const StepArgs &a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    const size_t lds_w = (lds_bytes(pd, LGS, a.lmbda_trk != nullptr) + 255) & ~(size_t)255;     // one wave's slice
    DevProps dp;
    if (!device_props(&dp, pd.dev_id)) return BT_EHIP;
    const int n_cu = dp.n_cu;
    const size_t lds_cu = dp.lds_cu;
    // waves per workgroup = per CU: two per SIMD (the kernel's 256 registers) unless their LDS slices do not fit
    int W = (int)std::min<size_t>(8, lds_cu / lds_w);
    if (W < 1) return BT_EUNSUPPORTED;
    const int per_cu = W;                                              // waves per CU
    // waves per workgroup: 4.  The waves of a workgroup add their sums up before the atomics; larger workgroups halve the atomics
    // again but wait for more partners.  Measured with the waves' priorities taking turns (profiles/r05_edge2_workgroup.txt; us at
    // 8192 / 16384 / 32768 tiles): 2 waves 87 / 139 / 245, 4 waves 77 / 130 / 240, 8 waves (the whole CU) 76 / 133 / 248; at one tile
    // per wave (2048 tiles) 48 / 36 / 43.
    const int wg = 4;
    if (wg < W) W = wg;
    const size_t lds = lds_w * (size_t)W;
    static LdsLimit lds_limit;
    if (!lds_limit.ensure(reinterpret_cast<const void *>(&k_edge2<NT, LGS, LOSS>), lds, pd.dev_id)) return BT_EHIP;
    if (a.dbg & 128) {                                                  // measurement: what the runtime says fits a CU
        static bool said = false;
        if (!said) {
            said = true;
            for (int w = 1; w <= 8; w *= 2) {
                int nb_ = -1;
                const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, k_edge2<NT, LGS, LOSS>, 64 * w, lds_w * (size_t)w);
                fprintf(stderr, "k_edge2: %d waves per workgroup, %zu bytes of LDS: %d workgroups per CU (%s); n_cu %d, LDS per CU %zu\n", w,
                        lds_w * (size_t)w, nb_, hipGetErrorString(e), n_cu, lds_cu);
            }
        }
    }
    const int max_waves = n_cu * per_cu;
    const int tpw = (pd.T + max_waves - 1) / max_waves, nw = (pd.T + tpw - 1) / tpw, nb = (nw + W - 1) / W;
    if (ev0) hipExtLaunchKernelGGL((k_edge2<NT, LGS, LOSS>), dim3(nb), dim3(64 * W), lds, st, ev0, ev1, 0, pd, a, tpw, (int)(lds_w / sizeof(double)));
    else hipLaunchKernelGGL((k_edge2<NT, LGS, LOSS>), dim3(nb), dim3(64 * W), lds, st, pd, a, tpw, (int)(lds_w / sizeof(double)));
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

}  // namespace e2

// pose+structure reduce of a plan the edge-major layout applies to (edge_applies, ba_plan.hpp)
int launch_edge2(const PlanDev &pd, const StepArgs &a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    if ((unsigned long long)pd.e_all * (unsigned long long)a.tstride * 4ull >= (1ull << 32)) return BT_EUNSUPPORTED;   // 32-bit byte offsets into the targets
    if ((unsigned long long)pd.p_tot * (unsigned long long)a.mstride * 4ull >= (1ull << 32)) return BT_EUNSUPPORTED;
    const bool s8 = pd.em_lgs == 3;            // the 8-observation graphs of the benchmark generator
#define BT_E2_LOSS(NT, LGS)                                                                             \
    (a.loss == BT_LOSS_HUBER ? e2::launch_t<NT, LGS, BT_LOSS_HUBER>(pd, a, st, ev0, ev1)                \
     : a.loss == BT_LOSS_CAUCHY ? e2::launch_t<NT, LGS, BT_LOSS_CAUCHY>(pd, a, st, ev0, ev1)            \
                                : e2::launch_t<NT, LGS, BT_LOSS_TRIVIAL>(pd, a, st, ev0, ev1))
    if (pd.max_cams <= 8) return s8 ? BT_E2_LOSS(3, 3) : BT_E2_LOSS(3, -1);
    return BT_E2_LOSS(4, -1);
#undef BT_E2_LOSS
}

}  // namespace bt
