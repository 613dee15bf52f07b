// ba_plan.hpp — structure of one BA edge list, shared by the host planner
// (ba_plan.cpp), the kernels (ba_kernels.hip) and the C ABI (ba_api.cpp).
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/batrack_ba.h"

namespace bt {

constexpr int kLanes = 64;          // tracks per wave tile (one lane = one track)
constexpr int kTileCamSoft = 16;    // close a tile when its camera union would exceed this
constexpr int kTileCamHard = 64;    // a track that sees more free cameras than this sits in no tile: a LOOSE track (ba_loose.hip)
constexpr int kTileCamF64 = 32;     // ... and so does one that sees more than THIS, where a plan has only a few of them (kFewHubs): a tile of 32
constexpr int kFewHubs = 64;        //     cameras keeps its E in LDS as double, one of 33 .. 64 would turn the whole plan's per-edge maths to float32
constexpr int kMaxFree = 255;       // free poses the block-sparse solvers take (8-bit pose numbers in their tables)
constexpr int kMaxFreeWide = 2048;  // ... and the dense solver of larger systems (ba_dense.hip: the right-hand side lives in LDS)
constexpr int kPairAccStride = 32;  // doubles per camera pair (27 used: 21 Bjj + 6 gj)
// Private copies of y and of the per-pair sums for k_edge2 (ba_edge2.hip): thousands of waves end with atomics on the SAME few
// cache lines (y is 6n doubles in all; a pair's 27 sums are hit by every wave of its source frame) and atomics on one line are
// served one after the other — at 8.4M edges they were 35 of the kernel's 170 us.  A wave adds to copy (its index mod the count);
// k_pair_finalize adds the copies up and clears them.
constexpr int kPrivY = 16, kPrivP = 4;
// behind them: one arrival counter per SIMD of the chip (k_edge2: which of a SIMD's two waves am I — see "wave priority" there)
constexpr int kPrivArrive = 8192;   // ints: xcc (3 bits) | se (3) | sh (1) | cu (4) | simd (2)
constexpr int kPairGeomFloats = 20; // R(9) t(3) Ki(4) Kj(4)
constexpr int kLdsRowStride = 66;
constexpr int kMaxLevelCols = 4;
// LDS a solver workgroup may ask for (160 KB a CU), and what the block-sparse factor with its tables takes there at `elem` bytes
// per number: the planner sends systems whose factor does not fit as double to the dense solver (ba_plan.cpp: `wide`).
constexpr size_t kLdsBudget = 160 * 1024 - 512;
inline size_t solve_lds_bytes_raw(size_t nnzb, size_t D, size_t nupd, size_t n, size_t nlev, size_t ndp, size_t elem) {
    size_t b = (nnzb * 36 + 2 * D) * elem;                                 // Lw, z, zt
    b = (b + 15) / 16 * 16;
    b += nupd * 3 * sizeof(unsigned short);                                // update triples
    b = (b + 15) / 16 * 16;
    // row_idx, col_ptr, upd_ptr, upd_next, dp_ptr, lvl_ptr, lvl_cols, dp
    b += (nnzb + 4 * (n + 1) + nlev + 1 + n + ndp) * sizeof(int);
    b = (b + 15) / 16 * 16 + nlev * kMaxLevelCols * 8 * sizeof(int);       // lvl_meta
    return b + 64;
}
constexpr int kSpGroupMax = 32;     // tiles per group of k_pair_finalize's positional sums of the tiles' Schur products
constexpr int kMaxTilePairs = 192;  // distinct camera pairs per tile whose geometry is kept in LDS    // columns of the reduced system factored concurrently (one critical wave each)   // floats per local E row (bank-conflict-free)

// doubles per tile of StepArgs::spart (plans with sp_ok): the tile's Schur product (ntl lower 16x16 tiles of 256), E Q w'
// (max_rows16) and its per-pair sums (max_tile_pairs x 32: 27 used)
#ifdef __HIPCC__
#define BT_HD __host__ __device__
#else
#define BT_HD
#endif
// doubles of the private region (WsLayout::priv): the copies of y, of the per-pair sums, the arrival counters
BT_HD inline size_t priv_copy_doubles(size_t D, size_t pairs) { return (size_t)kPrivY * D + (size_t)kPrivP * pairs * kPairAccStride; }
BT_HD inline size_t priv_doubles(size_t D, size_t pairs) { return priv_copy_doubles(D, pairs) + kPrivArrive / 2; }
BT_HD inline size_t sp_tile_doubles(int max_rows16, int max_tile_pairs) {
    const size_t nt = (size_t)max_rows16 / 16;
    return nt * (nt + 1) / 2 * 256 + (size_t)max_rows16 + (size_t)(max_tile_pairs > 0 ? max_tile_pairs : 1) * 32;
}

// LDS bytes of k_etile's pose+structure instantiation (ba_etile.hip) for a plan whose largest tile has max_rows16 rows of E
// and max_tile_pairs camera pairs, with per-edge numbers of rsz bytes: the planner lays a plan out for that kernel only if
// this fits (kEtileLdsBudget)
constexpr size_t kEtileLdsBudget = 160 * 1024 - 512;
BT_HD inline size_t etile_full_lds_bytes(int max_rows16, int max_tile_pairs, size_t rsz) {
    const size_t mtp = (size_t)(max_tile_pairs > 0 ? max_tile_pairs : 1), rows = (size_t)max_rows16;
    size_t smax = 1;
    while (smax < mtp) smax <<= 1;
    return (size_t)4 * 26 * smax * sizeof(double) + (rows * kLdsRowStride + 128 + mtp * kPairGeomFloats) * rsz + rows * sizeof(int) + 64;
}

// Device-side view: raw pointers into one device allocation + sizes.
struct PlanDev {
    int E, n_buf, p_tot, fixedp, n_all, n, D, m, P, T, slots, erows, nnzb, nupd, max_rows16;
    const int32_t *kx;
    const uint32_t *act_bits;                                // bit p: patch p has a track; [ceil(p_tot / 32)]
    const int32_t *act_rank;                                 // tracks before the word's first patch: track(p) = act_rank[p>>5] + popc(bits below p)
    const int32_t *pair_i, *pair_j;
    const int32_t *tile_trk0, *tile_ntrk, *tile_ncam, *tile_cam0, *tile_slot0, *tile_nslot, *tile_erow0;
    const int32_t *tile_cams;
    const int32_t *slot_edge, *slot_pair;
    const uint16_t *slot_lab;
    const int32_t *tile_pair0, *tile_npair, *tile_pairs;     // distinct camera pairs of a tile (global pair ids)
    const uint16_t *tile_cut8, *tile_cut16;                  // per tile: slot ranges of k_tile's 8 / 16 waves [9] / [17] (ba_plan.cpp)
    const int32_t *tile_ij, *tile_kx;                        // per tile: cameras of its pairs [max_tile_pairs], patch of its tracks [64]
    const int32_t *tile_flags;                               // bit 0: same cameras as the previous tile, bit 1: same pair list
    const uint8_t *slot_lp;                                  // local pair index of a (slot, lane) within its tile
    int max_tile_pairs, max_tile_slots, max_cams;
    long long e_all;                                         // length of the caller's edge list (slot_edge indexes it)
    const uint16_t *slot_code;                               // (slot, lane): local target camera | local pair << 8
    const uint8_t *tile_la;                                  // (tile, lane): local source camera of the lane's track (0xff: fixed)
    const int32_t *tile_rec;                                 // 8 ints per tile (ba_plan.cpp)
    const int32_t *it_edge;                                  // edge-major layout (ba_plan.cpp): [iterations][64]
    const uint32_t *tile_sinfo;                              // [tiles][64]
    const int32_t *pm_edge, *pm_rec;                         // pair-major layout (ba_plan.cpp): [rounds][64], [tiles][4]
    const uint8_t *pm_lb, *pm_la;                            // [tiles][64]: target camera of local pair s / source camera of track l
    const int32_t *pp_ptr, *pp_idx;                          // sp_ok: the (tile << 6 | local pair) entries of every camera pair, CSR over the pairs
    const int32_t *sg_ptr; int sg_n;                         // sp_ok: first tile of every group of consecutive same-camera tiles (sg_n + 1 entries)
    int et_lgts;                                             // sp_ok: log2 of the track stride of StepArgs::esave (>= the largest tile's tracks)
    int sp_ok;                                               // k_etile leaves per-tile Schur products and pair sums (StepArgs::spart) instead of atomics
    int pm_ok;                                               // the pair-major tables exist (every tile has at most 64 camera pairs)
    // LOOSE tracks (ba_plan.cpp): tracks seen by more than kTileCamHard free cameras sit in no tile; their edges are walked by
    // k_loose_reduce / k_loose_update (ba_loose.hip), a workgroup per track, the track's E in LDS over ALL free cameras
    const int32_t *lz_trk, *lz_ptr, *lz_edge, *lz_pair;      // track of loose track l; its edges [lz_ptr[l], lz_ptr[l + 1]): edge id, camera pair
    int nlz;
    int dev_id;                                              // the device the plan's tables live on (the launchers' per-device caches: no hipGetDevice per launch)
    int wide;                                                // more than kMaxFree free poses, or a factor too large for LDS as double: the dense solver (ba_dense.hip); perm is the identity, the packed form is the lower triangle by blocks
    int trk_off;                                             // sharded plan: distinct tracks of the full edge list in front of this rank's first
    int em_self;                                             // some edge has ii == jj (its source-camera E lands on a target row)
    int em_ok, em_its, em_lgs;                               // every tile slot-uniform; total iterations; log2 S if the same for all tiles, else -1
    int st_ok;                                               // the compact tables of k_stream exist (the plan was laid out for the wave-per-tile kernels)
    int st_min, em_min;                                      // stream_min_tiles() / edge_min_tiles() when the plan was built: the launch-time choice is the plan's own
    const int32_t *col_ptr, *row_idx, *upd_ptr, *upd, *blk_col, *upd_next;
    // elimination order and level schedule of the reduced solver
    int nlev, ndp;
    const int32_t *perm, *blk_src, *lvl_ptr, *lvl_cols, *col_lvl, *dp_ptr, *dp;
    // fused schedule (k_solve_fused): pending updates per destination block, lazy triples per column
    const int32_t *fz_pend_ptr, *fz_pend, *fz_lazy_ptr, *fz_lazy, *fz_yurg, *fz_meta, *fz_pmeta, *bs_sync, *fz_rowinfo, *fz_pfirst, *fz_psecond;
    int fz_npend, fz_nlazy, fz_ok;      // fz_ok: no level has more than two columns
    int fzp_ok;                         // and every column's panel fits one wave (k_solve_pipe)
    const int32_t *lvl_meta;   // [nlev][kMaxLevelCols][8]: col, diag pos, #sub-blocks, first rest triple, #rest triples, dp first, #dp, 0  (col = -1: unused)
};

// Byte offsets of the regions inside the caller's workspace.
struct WsLayout {
    size_t sys, pairacc, zero_bytes;   // [sys, sys+zero_bytes) is cleared every reduce
    size_t packed, pairgeo, qw, lfac, linv, zvec, dx, dx0, status, spart, esave, total;
    size_t priv;                       // inside the cleared region: [kPrivY][D] then [kPrivP][pairs][kPairAccStride] doubles; 0 = none
};

}  // namespace bt

namespace bt { struct PlanOffsets { size_t ab, ar, bc, pme, pmr, pmb, pml, ppp, ppi, sgp, bs, bss, c0, cams, cl, cp, dp, dpp, e0, fl, flp, fm, fp, fpf, fpm, fpp, fps, fri, fy, ite, kx, lc, lm, lp, lzt, lzp, lze, lzq, pi, pj, pm, ri, s0, sc, se, sl, slp, sn, sp, t0, tc, tc16, tc8, tf, tij, tkx, tla, tn, tnp, tp0, tps, trec, tsi, u, un, up; }; }

struct bt_plan {
    bt_plan_info info{};
    bt::PlanOffsets off{};        // byte offsets of the arrays inside the device buffer (ba_api.cpp)
    size_t dev_bytes = 0;          // bytes of the device buffer in use
    size_t pk_off = 0;             // packed edge list (kk << 32 | ii << 16 | jj, 8 bytes per edge) inside the device buffer, 0 = not kept
    int n_act_words = 0, n_tile_ij = 0;   // sizes of act_bits / tile_ij (for shifted clones)
    int cnt_nlev = 0, cnt_ndp = 0, cnt_npend = 0, cnt_nlazy = 0;   // sizes bind_pointers needs without the host arrays (shifted clones have none)
    std::vector<int32_t> kx, trk_loc, act_rank;                    // trk_loc: host only
    std::vector<int32_t> trk_win;                                  // track of patch trk_win_lo + i (-1: none): the window of patches the edges name
    int64_t trk_win_lo = 0;
    mutable std::vector<int32_t> trk_of_patch;                     // the same over the whole patch buffer, expanded on request (bt_plan_array: tests)
    std::vector<uint32_t> act_bits;
    std::vector<char> stage;     // upload staging (kept with the object: ba_api.cpp)
    std::vector<int32_t> pair_i, pair_j;
    std::vector<int32_t> tile_trk0, tile_ntrk, tile_ncam, tile_cam0, tile_slot0, tile_nslot, tile_erow0;
    std::vector<int32_t> tile_cams;
    std::vector<int32_t> slot_edge, slot_pair;
    std::vector<uint16_t> slot_lab;
    std::vector<int32_t> tile_pair0, tile_npair, tile_pairs, tile_flags, tile_ij, tile_kx;
    std::vector<uint16_t> tile_cut8, tile_cut16;
    std::vector<uint8_t> slot_lp, tile_la;
    std::vector<uint16_t> slot_code;
    std::vector<int32_t> tile_rec, it_edge;
    std::vector<uint32_t> tile_sinfo;
    int em_ok = 0, em_lgs = -1, em_self = 0, st_ok = 0, st_min = 1 << 30, em_min = 1 << 30;
    std::vector<int32_t> pm_edge, pm_rec;
    std::vector<uint8_t> pm_lb, pm_la;
    std::vector<int32_t> pp_ptr, pp_idx, sg_ptr;
    std::vector<int32_t> lz_trk, lz_ptr, lz_edge, lz_pair;   // loose tracks (more than kTileCamHard free cameras)
    int pm_ok = 0, sp_ok = 0, sg_n = 0, et_lgts = 0, trk_off = 0;
    int wide = 0;                                             // more than kMaxFree free poses, or a factor too large for LDS as double: dense solve, no symbolic tables
    int nlz = 0;                                              // loose tracks (set at upload; clones copy it)
    int dev_id = 0;                                           // the device the tables were uploaded to
    // plans whose pm_edge is written on the device (plan_device.hip): the table's rounds, and what the kernels need of the host's analysis
    int dev_pm = 0;
    int dev_slots = 0;                                        // likewise the [slots][64] arrays and the wave cuts of a 64-track layout
    int dev_wpt = 0;                                          // ... and with them the tables of the wave-per-tile kernels (slot_code, tile_la, it_edge, tile_sinfo)
    std::vector<int32_t> dev_off;                             // [m + 1]: first position of every track's edges in the grouped order
    std::vector<int32_t> dev_pbase;                           // aligned slot layout (ba_plan.cpp): first slot of every (tile, local pair), as tile_pairs; empty: a track's s-th edge is its slot s
    mutable std::vector<int32_t> dev_readback;                // bt_plan_array(pm_edge / pm_rec) of such a plan
    std::vector<int32_t> dev_pair_of;                         // [nw * nw]: pair index of (i - f_lo, j - f_lo) or -1
    int64_t dev_f_lo = 0, dev_nw = 0;
    int64_t dev_q0 = 0;                                       // sharded: position of the rank's first edge in the device's sorted list
    long long pm_rounds = 0;
    long long em_its = 0;
    int max_tile_pairs = 0, max_tile_slots = 0;
    std::vector<int32_t> col_ptr, row_idx, upd_ptr, upd, blk_col, upd_next;
    std::vector<int32_t> perm, blk_src, lvl_ptr, lvl_cols, col_lvl, dp_ptr, dp, lvl_meta;
    std::vector<int32_t> fz_pend_ptr, fz_pend, fz_lazy_ptr, fz_lazy, fz_yurg, fz_meta, fz_pmeta, bs_sync, fz_rowinfo, fz_pfirst, fz_psecond;   // fused schedule (k_solve_fused)
    int fz_ok = 0, fzp_ok = 0;
    int max_rows16 = 16;
    long long e_all = 0;
    bt::WsLayout ws{};
    void *dev_base = nullptr;   // one device allocation holding every array above (from the pool in ba_api.cpp)
    // the stream the plan's kernels were last launched on: bt_plan_destroy records an event there, and the next plan that
    // reuses the device buffer makes its table upload wait for it (the tables must not change under queued kernels)
    mutable void *last_stream = nullptr;
    mutable bool launched = false;
    // hipEvent_t behind a clone's copies on the plan stream (ba_api.cpp: mark_launch), else null.  Atomic: two host threads may
    // launch the same fresh clone; whoever finds the event complete CLAIMS it (exchange) before handing it back to the pool —
    // handed back twice, two later clones would share one event.
    mutable std::atomic<void *> ready{nullptr};
    size_t dev_cap = 0;
    int64_t k_hi = -1;                                        // largest patch index the plan's tables hold (bt_plan_create_shifted_spec checks k_hi + dk < p_tot)
    void *spec_ev = nullptr;                                  // hipEvent_t behind the verdict of a speculative clone (bt_plan_spec_confirm), else null
    int *spec_flag = nullptr;                                 // its two pinned ints: list differs / index out of range
    int spec_epoch = 0;                                       // > 0: no event — the verdict is complete when spec_flag[2] holds this (bt_plan_spec_bind)
    void *spec_stream = nullptr;                              //   (the stream the comparison was launched on: what a poll that runs out of patience waits for)
    int spec_unbound = 0;                                     // made AHEAD of its list (bt_plan_preshift): no step before bt_plan_spec_bind
    bt::PlanDev dev{};

    // Back to the state of a new object, but with the vectors' capacity kept: destroyed plans are recycled
    // by bt_plan_create (ba_api.cpp), so that a plan per frame costs no heap traffic.
    void recycle() {
        info = bt_plan_info{};
        for (auto *v : {&kx, &trk_of_patch, &trk_win, &trk_loc, &pair_i, &pair_j, &tile_trk0, &tile_ntrk, &tile_ncam,
                        &tile_cam0, &tile_slot0, &tile_nslot, &tile_erow0, &tile_cams, &slot_edge, &slot_pair,
                        &tile_pair0, &tile_npair, &tile_pairs, &tile_flags, &tile_ij, &tile_kx, &col_ptr, &row_idx,
                        &upd_ptr, &upd, &blk_col, &upd_next, &perm, &blk_src, &lvl_ptr, &lvl_cols, &col_lvl, &dp_ptr,
                        &dp, &lvl_meta, &fz_pend_ptr, &fz_pend, &fz_lazy_ptr, &fz_lazy, &fz_yurg, &fz_meta, &fz_pmeta,
                        &bs_sync, &fz_rowinfo, &fz_pfirst, &fz_psecond})
            v->clear();
        tile_cut8.clear(); tile_cut16.clear();
        pm_edge.clear(); pm_rec.clear(); pm_lb.clear(); pm_la.clear(); pp_ptr.clear(); pp_idx.clear(); sg_ptr.clear(); pm_ok = 0; sp_ok = 0; pm_rounds = 0; wide = 0; nlz = 0;
        dev_pbase.clear(); lz_trk.clear(); lz_ptr.clear(); lz_edge.clear(); lz_pair.clear();
        slot_lab.clear(); slot_lp.clear(); tile_la.clear(); slot_code.clear(); tile_rec.clear(); it_edge.clear(); tile_sinfo.clear(); em_ok = 0; st_ok = 0; em_its = 0; em_lgs = -1; act_bits.clear(); act_rank.clear(); stage.clear();
        max_tile_pairs = max_tile_slots = 0;
        fz_ok = fzp_ok = 0;
        max_rows16 = 16;
        ws = bt::WsLayout{};
        dev_base = nullptr; dev_cap = 0;
        last_stream = nullptr; launched = false; ready = nullptr; k_hi = -1; spec_ev = nullptr; spec_flag = nullptr; spec_unbound = 0; spec_epoch = 0; spec_stream = nullptr;
        dev = bt::PlanDev{};
    }
};

namespace bt {
// BT_FORCE — the ONE environment switch of the kernel / solver / planner choices (tests and measurement; production leaves it
// unset).  Comma-separated tokens, parsed once per process:
//   kernel=k_tile | k_stream | k_edge2 | k_etile   the Jacobian kernel every plan whose layout admits it takes
//   solver=fused | lds | lds32 | global            the reduced solver (default: k_solve_pipe where the plan allows it)
//   order=natural                                  no twisted elimination order
//   prec=f32                                       float32 per-edge maths on the k_tile path (round 2's numerics)
//   wide=0 | wide=1                                k_tile's 8 / 16 waves per tile
//   plan=host                                      no device-side planner
//   wpt=0                                          no wave-per-tile kernels (= bt_config_wave_per_tile_kernels(0))
struct Force {
    int kernel = -1;        // -1 the plan's own choice, 0 k_tile, 1 k_stream, 2 k_edge2, 3 k_etile
    int solver = -1;        // -1 default, 0 k_solve_fused (barrier per level), 1 k_solve_lds<double>, 2 k_solve_lds<float>, 3 k_solve_global
    int natural_order = 0, f32_edges = 0, tile_wide = -1, host_plan = 0, wpt_off = 0;
};
const Force &force();
// BT_PLAN_PROF: time per planner phase on stderr (measurement)
bool plan_prof();
// Tile counts from which the wave-per-tile kernels take a graph (k_edge2, k_stream; the environment overrides are for
// measurement and tests).  The planner lays out their tables only for plans that will use them.
int edge_min_tiles();
int stream_min_tiles();
// the wave-per-tile kernels for graphs of >= 2048 tiles, on by default (bt_config_wave_per_tile_kernels); returns the previous setting
int config_wave_per_tile_kernels(int enable);
// Pure host analysis (no HIP).  Returns BT_OK or an error code.
// `packed` (optional): the edges as 8-byte words kk << 32 | ii << 16 | jj, already range-checked (ii / jj / kk are then not read)
// `dstats` (optional; plan_device.hip): the per-track figures of the edge list as a kernel gathered them — the analysis then
// reads NO edge: it lays out tracks, pairs, tiles and the reduced system from the tracks' (source frame, target mask), and
// leaves the one edge-sized table of a window plan (pm_edge) and the tiles' round counts to the device
// (bt_plan::dev_pm).  Returns BT_NEED_EDGES where that does not apply (the caller then runs the analysis on the edges).
struct PatchStat { int32_t cnt, src, src_min, pad; unsigned long long mask, mask2; };      // bit b of (mask | mask2 << 64): an edge into frame src - 64 + b
// the target frames a track observes MORE THAN ONCE (same bit numbering as PatchStat's masks): gathered for lists large enough
// for the wave-per-tile kernels, whose aligned slot layout needs to know where a track's extra edges can be
struct RepStat { unsigned long long rmask, rmask2; };
struct DevPlanStats {
    const PatchStat *tab;          // the patches tab_lo .. tab_lo + tab_n - 1: all the list names (kmin .. kmax), or — a rank's plan of a
    int64_t tab_lo, tab_n;         // sharded solve — only those of its own range: the rank's host share is then ~ its share of the tracks
    int64_t kmin, kmax, n_all, f_lo;
    int any_self;
    int rep_known;                 // the repeated targets were looked for (rtab null then means: there are none)
    const RepStat *rtab;           // as tab, or null
    // a rank's plan: what it needs to know of the OTHER ranks' tracks, reduced on the device from the full table
    int sliced;                    // tab holds the own range only
    int64_t trk_before, edges_before;      // distinct tracks / edges of the patches in front of the range
    const uint32_t *pattern;       // [n][pattern_words] bits: free cameras u >= v that some track of the list couples (null: n = 0 or a wide plan)
    int pattern_words;
};
enum { BT_NEED_EDGES = 2 };
int build_plan_host(const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E,
                    int64_t n_buf, int64_t p_tot, int64_t fixedp, int64_t n_all_min,
                    int64_t own_lo, int64_t own_hi, bt_plan *plan, const uint64_t *packed = nullptr, bool keep_slots = false,
                    const DevPlanStats *dstats = nullptr);
// the same packing on the device (plan_pack.hip): `out` E words and `bad` one int (set to 1 on an index out of range), device memory
int launch_shift_match(const uint64_t *nw, const uint64_t *ow, int64_t E, int *out, void *stream);
// pack the list into `out` and set host_flags[0] unless word[e] == ow[e] + delta for every edge, host_flags[1] on an index out of
// range (bt_plan_create_shifted_spec; host_flags: device-visible pinned memory, cleared by the caller)
int launch_pack_match_expect(const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E, int64_t n_buf, int64_t p_tot,
                             const uint64_t *ow, uint64_t delta, uint64_t *out, int *host_flags, void *stream);
// the list against `ow` word for word; host_flags (pinned): [0] differs, [1] index out of range, [2] = epoch once every workgroup is through
int launch_match_done(const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E, int64_t n_buf, int64_t p_tot,
                      const uint64_t *ow, int *host_flags, unsigned *ticket, int epoch, void *stream);
int launch_words_add(const uint64_t *ow, uint64_t delta, uint64_t *out, int64_t E, void *stream);
int launch_plan_shift(int32_t *kx, int m, int32_t *tile_kx, int nkx, int32_t *tile_ij, int nij, int32_t *pair_i, int32_t *pair_j, int P,
                      const uint32_t *old_bits, uint32_t *new_bits, int32_t *new_rank, int nwords, int df, int dk, void *stream);
int launch_pack_edges(const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E, int64_t n_buf, int64_t p_tot,
                      uint64_t *out, int *bad, void *stream);
int pack_edges_host(const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E, int64_t n_buf, int64_t p_tot, uint64_t *out);
// the passes over the edges on the device (plan_device.hip; bt_plan_create)
int plan_device_stats(const uint64_t *d_words, int64_t E, int64_t p_tot, int64_t fixedp, int64_t own_lo, int64_t own_hi, void *stream, DevPlanStats *st, int64_t *tracks);
int plan_device_rounds(const bt_plan *pl, int64_t E, void *stream, int64_t *rounds);
int plan_device_fill(const bt_plan *pl, int64_t E, int32_t *d_rec, int32_t *d_pm_edge, int64_t rounds, void *stream);
int plan_device_slots_stage(const bt_plan *pl, void *stream);
// the tables of the wave-per-tile kernels, written by the same passes (device pointers into the plan's buffer; tile_rec is the
// host's upload, the passes add the straddle flag; it_edge / tile_sinfo null: the layout of k_edge2 does not apply)
struct DevWptOut { uint16_t *slot_code; uint8_t *tile_la; int32_t *tile_rec; int32_t *it_edge; uint32_t *tile_sinfo; int64_t its; };
int plan_device_slots_fill(const bt_plan *pl, int64_t E, int32_t *d_slot_edge, int32_t *d_slot_pair, uint16_t *d_slot_lab, uint8_t *d_slot_lp,
                           uint16_t *d_cut8, uint16_t *d_cut16, void *stream, const DevWptOut *wpt = nullptr);
// after the stream of plan_device_slots_fill has been waited for: 1 = some tile is not slot-uniform (no k_edge2 for this plan)
int plan_device_em_verdict();
// Copies the arrays to the device and fills plan->dev (ba_api.cpp).
int upload_plan(bt_plan *plan);
}  // namespace bt
