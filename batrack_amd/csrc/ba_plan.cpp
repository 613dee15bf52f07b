// ba_plan.cpp — host analysis of one BA edge list (no HIP calls in this file).
//
// What the reference recomputes inside every BA_rgbd_droid call
//   n = max(ii, jj) + 1                      /root/reference/main/backend/ba.py:219
//   kx, kk' = torch.unique(kk, sorted=True)  /root/reference/main/backend/ba.py:276
//   dense E [n, m, 6] scatter targets        /root/reference/main/backend/ba.py:284-285
// is done here once per edge list, and laid out for the tile kernel:
//   * tracks sorted by patch slot; a TILE is up to 64 consecutive tracks whose
//     union of free cameras stays small; lane l of the tile's wave owns track l;
//   * SLOT s of a tile holds the s-th edge of each of its tracks (edges of a track
//     ordered by camera pair, so tracks with the same observation pattern put
//     the same pair in the same slot and the per-pair reductions are wave-uniform);
//   * the distinct (ii, jj) camera pairs (their relative pose is per pair, not per edge);
//   * the block-sparsity of the reduced camera system and of its Cholesky factor.
#include "ba_plan.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <unordered_map>

namespace bt {

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Tile counts from which a plan is laid out for the wave-per-tile kernels k_stream / k_edge2 (2048: profiles/r02_kernel_choice.txt).
// Those kernels evaluate an edge in MIXED precision since round 4 (reprojection and residual in float64, Jacobians and their
// products in float32: ba_edge.hpp edge_eval_mixed) and keep the update within north_star's 1e-5 of the reference's float64
// run (measured 1e-6 .. 5e-6 on the benchmark graphs; S and y 1e-7).  A caller who wants the reduced system itself to 1e-10
// at every size switches them off — bt_config_wave_per_tile_kernels(0) — and every plan then takes the float64-per-edge tile
// kernels (a third of the throughput from 2048 tiles on).  BT_FORCE (ba_plan.hpp) forces one kernel for every plan (tests,
// measurement).  A plan records what it was laid out for (st_ok,
// em_ok, st_min, em_min): the launch-time choice never depends on a later change of this setting.
const Force &force() {
    static const Force f = [] {
        Force r;
        const char *e = std::getenv("BT_FORCE");
        if (!e) return r;
        std::string s(e);
        size_t pos = 0;
        while (pos <= s.size()) {
            const size_t end = std::min(s.find(',', pos), s.size());
            const std::string tok = s.substr(pos, end - pos);
            pos = end + 1;
            if (tok == "kernel=k_tile") r.kernel = 0; else if (tok == "kernel=k_stream") r.kernel = 1;
            else if (tok == "kernel=k_edge2") r.kernel = 2; else if (tok == "kernel=k_etile") r.kernel = 3;
            else if (tok == "solver=fused") r.solver = 0; else if (tok == "solver=lds") r.solver = 1;
            else if (tok == "solver=lds32") r.solver = 2; else if (tok == "solver=global") r.solver = 3;
            else if (tok == "order=natural") r.natural_order = 1; else if (tok == "prec=f32") r.f32_edges = 1;
            else if (tok == "wide=0") r.tile_wide = 0; else if (tok == "wide=1") r.tile_wide = 1;
            else if (tok == "plan=host") r.host_plan = 1; else if (tok == "wpt=0") r.wpt_off = 1;
            else if (!tok.empty()) std::fprintf(stderr, "batrack: unknown BT_FORCE token '%s'\n", tok.c_str());
        }
        return r;
    }();
    return f;
}
bool plan_prof() { static const bool p = std::getenv("BT_PLAN_PROF") != nullptr; return p; }

static std::atomic<int> g_wpt_kernels{-1};           // -1: not set by the caller -> BT_FORCE decides (default on)
static int wpt_env() { return force().wpt_off ? 0 : 1; }
int config_wave_per_tile_kernels(int enable) {
    const int prev = g_wpt_kernels.load();
    if (enable >= 0) g_wpt_kernels.store(enable ? 1 : 0);
    return prev < 0 ? wpt_env() : prev;
}
static int wpt_kernels_on() {
    const int v = g_wpt_kernels.load();
    return v >= 0 ? v : wpt_env();
}
constexpr int kNever = 1 << 30;
int edge_min_tiles() {
    const int k = force().kernel;
    return k == 2 ? 1 : k >= 0 ? kNever : (wpt_kernels_on() ? 2048 : kNever);
}
int stream_min_tiles() {
    const int k = force().kernel;
    return (k == 1 || k == 2) ? 1 : k >= 0 ? kNever : (wpt_kernels_on() ? 2048 : kNever);
}
// the pair-major layout (k_etile): 1 = for the plans it was measured on (few tiles, deep tracks), 0 = never, 2 = every plan
static int etile_mode() { const int k = force().kernel; return k == 3 ? 2 : k >= 0 ? 0 : 1; }

static void layout_workspace(bt_plan *pl) {
    const bt_plan_info &I = pl->info;
    const size_t D = (size_t)(6 * I.n);
    WsLayout w{};
    size_t off = 0;
    w.sys = off;      off += (D * D + D) * sizeof(double);
    off = align_up(off, 256);
    w.pairacc = off;  off += (size_t)I.pairs * kPairAccStride * sizeof(double);
    off = align_up(off, 256);
    w.priv = 0;
    if (I.tiles >= pl->em_min) {       // (plans k_edge2 can take: by the tile count alone — whether the tiles are slot-uniform is, for
                                       //  a device-planned list, known only after the upload, and the layout must not depend on the planner)
        w.priv = off;
        off = align_up(off + priv_doubles(D, (size_t)I.pairs) * sizeof(double), 256);
    }
    w.zero_bytes = off - w.sys;
    w.packed = off;   off = align_up(off + ((size_t)I.nnz_blocks * 36 + D) * sizeof(double), 256);   // exchange form of [S | y]
    w.pairgeo = off;  off = align_up(off + (size_t)I.pairs * kPairGeomFloats * sizeof(double), 256);         // k_tile -> k_pair_finalize (float or double)
    w.qw = off;       off = align_up(off + (size_t)I.m * 2 * sizeof(double), 256);                           // (Q, w') per track (float2 or double2)
    // (a wide plan's dense factor: D x D doubles and the right-hand side behind it)
    w.lfac = off;     off = align_up(off + (pl->wide ? (D * D + D) * sizeof(double) : (size_t)I.nnz_blocks * 36 * sizeof(float)), 256);
    w.linv = off;     off = align_up(off + (size_t)I.n * 36 * sizeof(float), 256);
    w.zvec = off;     off = align_up(off + D * sizeof(float), 256);
    w.dx = off;       off = align_up(off + D * sizeof(float) + 64, 256);
    w.dx0 = off;      off = align_up(off + D * sizeof(float) + 64, 256);     // first solution of a refined solve (float32-factor systems)
    w.status = off;   off = align_up(off + 1024, 256);
    w.spart = off;
    if (pl->sp_ok) off = align_up(off + (size_t)I.tiles * sp_tile_doubles(pl->max_rows16, pl->max_tile_pairs) * sizeof(double), 256);
    w.esave = off;                                                 // k_etile -> k_etile_upd: the tiles' E, [tile][max_rows16][1 << et_lgts]
    if (pl->sp_ok) off = align_up(off + (((size_t)I.tiles * pl->max_rows16) << pl->et_lgts) * sizeof(double), 256);
    w.total = off;
    pl->ws = w;
    pl->info.workspace_bytes = (int64_t)w.total;
}

// Edges as the planner reads them: one 8-byte word per edge, kk << 32 | ii << 16 | jj (n_buf <= 32768, p_tot < 2^31).
// For index tensors on the device the words are packed (and range-checked) by a kernel and copied back in one piece —
// a third of the bytes of the three int64 arrays (ba_api.cpp); host arrays are packed here.
int pack_edges_host(const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E, int64_t n_buf, int64_t p_tot, uint64_t *out) {
    for (int64_t e = 0; e < E; ++e) {
        if (ii[e] < 0 || jj[e] < 0 || ii[e] >= n_buf || jj[e] >= n_buf) return BT_EINVAL;
        if (kk[e] < 0 || kk[e] >= p_tot) return BT_EINVAL;
        out[e] = ((uint64_t)kk[e] << 32) | ((uint64_t)ii[e] << 16) | (uint64_t)jj[e];
    }
    return BT_OK;
}

// Iterations of an edge-major tile (64 / S tracks each): an EVEN number (the last one empty where the tile's tracks end in an
// odd one), so that the flat iteration stream of a wave pairs up (2J, 2J + 1) inside every tile — k_edge2 (ba_edge2.hip) takes
// two iterations per step, one edge of each per lane.  Also for S = 1, where one iteration is the whole tile: its second
// iteration is a row of -1 (k_edge2 reads rows 2J and 2J + 1 whatever S is; an odd count would hand it the next tile's row).
static inline int32_t em_iterations(int32_t ntrk, int32_t G) {
    const int32_t nit = (ntrk + G - 1) / G;
    return (nit + 1) & ~1;
}

int build_plan_host(const int64_t *ii64, const int64_t *jj64, const int64_t *kk64, int64_t E,
                    int64_t n_buf, int64_t p_tot, int64_t fixedp, int64_t n_all_min,
                    int64_t own_lo, int64_t own_hi, bt_plan *pl, const uint64_t *packed, bool keep_slots, const DevPlanStats *dstats) {
    if (E < 0 || n_buf <= 0 || p_tot <= 0 || fixedp < 0 || n_all_min < 0 || n_all_min > n_buf) return BT_EINVAL;
    if (E > (int64_t)0x7fffffff / 2 || p_tot > (int64_t)0x7fffffff) return BT_EUNSUPPORTED;
    if (n_buf > 32768) return BT_EUNSUPPORTED;             // frame numbers are packed in pairs into signed 32-bit words (tile_ij)
    bt_plan_info &I = pl->info;
    I = bt_plan_info{};
    I.n_buf = n_buf; I.p_tot = p_tot; I.fixedp = fixedp;
    pl->e_all = E;
    // A plan of a SHARDED solve (the caller named a track range, whatever it covers): the ranks exchange [S | y] block by block of the
    // factor's pattern, so every rank must arrive at the same pattern — from what all of them can see, the tracks of the whole list.
    const bool sharded = own_hi > 0;
    if (own_hi <= 0) { own_lo = 0; own_hi = p_tot; }
    if (own_lo < 0 || own_hi > p_tot || own_lo > own_hi) return BT_EINVAL;
    const bool plan_prof = bt::plan_prof();                                        // measurement only: time per phase on stderr
    auto t_prev = std::chrono::steady_clock::now();
#define BT_TICK(name) do { if (plan_prof) { const auto t_now = std::chrono::steady_clock::now(); std::fprintf(stderr, "plan phase before %s: %.3f ms\n", name, std::chrono::duration<double, std::milli>(t_now - t_prev).count()); t_prev = t_now; } } while (0)

    BT_TICK("0");
    static thread_local std::vector<uint64_t> pk_scratch;
    const uint64_t *pk = packed;
    pl->dev_pm = 0; pl->dev_slots = 0; pl->dev_wpt = 0;
    if (dstats && (keep_slots || E <= 0)) return BT_NEED_EDGES;
    if (!pk && !dstats) {
        pk_scratch.resize((size_t)E + 1);
        const int rc = pack_edges_host(ii64, jj64, kk64, E, n_buf, p_tot, pk_scratch.data());
        if (rc != BT_OK) return rc;
        pk = pk_scratch.data();
    }
    auto KK = [&](int64_t e) { return (int64_t)(pk[e] >> 32); };
    auto II = [&](int64_t e) { return (int64_t)((pk[e] >> 16) & 0xffff); };
    auto JJ = [&](int64_t e) { return (int64_t)(pk[e] & 0xffff); };
    auto owned = [&](int64_t e) { return KK(e) >= own_lo && KK(e) < own_hi; };

    // ---- ONE pass over the edges: n_all (ba.py:219), the window of patches and frames they name, edges per track and per
    // target frame, the camera pairs in use, and per track its source frame and the set of its target frames (a 64-bit mask
    // around the first target seen: a track's observations span a window of frames, batrack.py:399-410)
    // (mask, mask2: target frames base + b and base + 64 + b, base = the source frame - 64: 128 bits around the source frame, in the
    //  host's own pass as in the device's table)
    // (32 bytes a patch: the passes below stream a million of these.  The targets a track observes MORE THAN ONCE — the aligned
    //  slot layout's multiplicities — are kept aside, looked up only for the tracks whose count exceeds their distinct targets:
    //  the device's table where it gathered them, a map filled by the host's own pass)
    struct PerPatch { int32_t cnt, src, base, last_j; uint64_t mask, mask2; };
    static thread_local std::unordered_map<int64_t, RepStat> rep_host;
    rep_host.clear();
#define BT_FOR_TARGETS(T_, fr_, ...)                                                                                            \
    do {                                                                                                                        \
        for (uint64_t mk_ = (T_).mask; mk_; mk_ &= mk_ - 1) { const int32_t fr_ = (T_).base + __builtin_ctzll(mk_); __VA_ARGS__ }        \
        for (uint64_t mk_ = (T_).mask2; mk_; mk_ &= mk_ - 1) { const int32_t fr_ = (T_).base + 64 + __builtin_ctzll(mk_); __VA_ARGS__ }  \
    } while (0)
    static thread_local std::vector<PerPatch> pp_tab;            // indexed by patch; only [kmin, kmax] of the previous plan is dirty
    static thread_local int64_t pp_lo = 0, pp_hi = -1;
    if ((int64_t)pp_tab.size() < p_tot) { pp_tab.assign((size_t)p_tot, PerPatch{0, 0, 0, 0, 0, 0}); pp_lo = 0; pp_hi = -1; }
    for (int64_t p = pp_lo; p <= pp_hi; ++p) pp_tab[(size_t)p].cnt = 0;
    PerPatch *pp = pp_tab.data();
    int64_t n_all = n_all_min, kmin = p_tot, kmax = -1;       // [kmin, kmax]: patches the edges name (a window of the buffer)
    int64_t f_lo = n_buf;                                     // first frame the edges name (the window's start, not the buffer's)
    bool sorted = true, masks_ok = true, mono_j = true;       // mono_j: within every track the target frames come in ascending order
    std::vector<int32_t> cj((size_t)n_buf + 2, 0);
    int64_t E_own = 0, k_prev = -1;
    bool src_ok = true, any_self = false;
    if (dstats) {
        // the same figures from the device's table (bit b of a mask: target frame src - 32 + b)
        n_all = std::max(n_all, dstats->n_all); f_lo = dstats->f_lo; kmin = dstats->kmin; kmax = dstats->kmax;
        any_self = dstats->any_self != 0; sorted = false;
        // a sharded plan lays out the tracks of [own_lo, own_hi) only: the device's sort keeps a patch's edges together in
        // patch order, so the rank's edges are ONE segment of the sorted list — it starts behind the edges of the patches in
        // front of the range (dev_q0).  The other ranks' tracks keep their (source frame, mask) for the pattern of S below.
        // (round 6: the table of such a plan holds the rank's own patches only — dstats->sliced —, the edges in front of them, the
        //  tracks in front of them and the coupling pattern of the whole list come reduced from the device)
        pl->dev_q0 = dstats->sliced ? dstats->edges_before : 0;
        for (int64_t k = dstats->tab_lo; k < dstats->tab_lo + dstats->tab_n; ++k) {
            const PatchStat &d = dstats->tab[(size_t)(k - dstats->tab_lo)];
            PerPatch &t = pp[k];
            t.src = d.src; t.base = d.src - 64; t.last_j = 0; t.mask = d.mask; t.mask2 = d.mask2;
            if (k >= own_lo && k < own_hi) { t.cnt = d.cnt; E_own += d.cnt; }
            else { t.cnt = 0; if (k < own_lo) pl->dev_q0 += d.cnt; }
        }
        if (E_own == 0) return BT_NEED_EDGES;                  // (a rank without tracks: nothing for the device passes to lay out)
    } else
    for (int64_t e = 0; e < E; ++e) {
        const int64_t k = KK(e), i = II(e), j = JJ(e);
        any_self |= i == j;
        n_all = std::max(n_all, std::max(i, j) + 1);
        f_lo = std::min(f_lo, std::min(i, j));
        kmin = std::min(kmin, k); kmax = std::max(kmax, k);
        if (k < k_prev) sorted = false;
        k_prev = k;
        if (k < own_lo || k >= own_hi) continue;
        ++E_own;
        ++cj[(size_t)j + 1];
        PerPatch &t = pp[k];
        // (128 bits around the SOURCE frame, as the device's table has them: a list is laid out the same way wherever its indices live)
        if (t.cnt++ == 0) { t.src = (int32_t)i; t.base = (int32_t)i - 64; t.mask = 0; t.mask2 = 0; }
        else { if (t.src != (int32_t)i) src_ok = false; if ((int32_t)j < t.last_j) mono_j = false; }
        t.last_j = (int32_t)j;          // one source frame per track: the caller builds ii = ix[kk] (batrack.py:199)
        const int64_t bit = j - t.base;
        if (bit < 0 || bit >= 128) masks_ok = false;
        else {
            uint64_t &mw = bit < 64 ? t.mask : t.mask2;
            if (mw & (1ull << (bit & 63))) { RepStat &r = rep_host[k]; (bit < 64 ? r.rmask : r.rmask2) |= 1ull << (bit & 63); }
            mw |= 1ull << (bit & 63);
        }
    }
    pp_lo = kmin; pp_hi = kmax;
    // sharded plan: the number of distinct tracks in front of this rank's range — a per-track lmbda tensor (ba.py:299-300) is
    // indexed by the GLOBAL track number, the kernels count from the rank's first track
    pl->trk_off = 0;
    if (dstats) {
        if (dstats->sliced) pl->trk_off = (int)dstats->trk_before;
        else for (int64_t k = kmin; k < std::min(own_lo, kmax + 1); ++k) pl->trk_off += dstats->tab[(size_t)(k - dstats->tab_lo)].cnt > 0 ? 1 : 0;
    } else if (E_own != E && own_lo > 0) {
        std::vector<uint8_t> seen((size_t)own_lo, 0);
        for (int64_t e = 0; e < E; ++e) { const int64_t k = KK(e); if (k < own_lo && !seen[(size_t)k]) { seen[(size_t)k] = 1; ++pl->trk_off; } }
    }
    pl->em_self = any_self ? 1 : 0;
    if (E == 0) f_lo = 0;
    I.n_all = n_all;
    I.sorted_input = sorted ? 1 : 0;
    const int64_t n = std::max<int64_t>(n_all - fixedp, 0);
    I.n = n;
    if (n > kMaxFreeWide) return BT_EUNSUPPORTED;          // (beyond kMaxFree: the dense solver, ba_dense.hip)
    if (!src_ok) return BT_EUNSUPPORTED;
    I.E = E_own;
    // Tile counts from which the wave-per-tile kernels take THIS plan.  A rank's plan of a sharded solve (own_lo / own_hi) holds
    // its share of the tiles: the thresholds shrink by that share, so that all ranks of a graph large enough for those kernels use
    // them (and the same per-edge precision), instead of a rank falling back to the float64 tile kernel because ITS shard is small.
    const auto shard_min = [&](int thr) {
        if (E_own >= E || thr <= 1 || thr >= (1 << 29)) return thr;
        return (int)std::max<int64_t>(1, ((int64_t)thr * E_own + E - 1) / E);
    };
    const int em_min_p = shard_min(edge_min_tiles()), st_min_p = shard_min(stream_min_tiles());

    BT_TICK("1");
    // unique tracks, ascending (ba.py:276); off = first position of a track's edges in the grouped order.
    // trk_of_patch is kept for the window only (trk_win[p - kmin]); bt_plan_array expands it on request.
    int32_t m = 0;
    pl->kx.clear();
    pl->trk_win_lo = kmin <= kmax ? kmin : 0;
    pl->k_hi = kmax;
    pl->trk_win.assign(kmin <= kmax ? (size_t)(kmax - kmin + 1) : 0, -1);
    static thread_local std::vector<int32_t> off_scratch;        // (a million tracks: no fresh pages per plan)
    std::vector<int32_t> &off = off_scratch;
    off.assign(1, 0);
    // (a rank's sliced table: only its own patches can carry a track of this plan)
    const int64_t p_first = dstats && dstats->sliced ? dstats->tab_lo : kmin, p_last = dstats && dstats->sliced ? dstats->tab_lo + dstats->tab_n - 1 : kmax;
    for (int64_t p = p_first; p <= p_last; ++p) {
        const int32_t c = pp[p].cnt;
        if (c > 0) { pl->kx.push_back((int32_t)p); off.push_back(off.back() + c); pl->trk_win[(size_t)(p - kmin)] = m++; }
    }
    I.m = m;
    auto trk_of = [&](int64_t k) { return pl->trk_win[(size_t)(k - kmin)]; };

    BT_TICK("2");
    // ---- distinct camera pairs, ascending (i, j); looked up through a table over the frames the edges name, [f_lo, n_all):
    // the window (about 20 frames), not the keyframe count of the whole sequence
    const int64_t nw = n_all - f_lo;
    std::vector<int32_t> pair_of((size_t)(nw * nw), -1);
    if (masks_ok) {
        // from the tracks' (source frame, target mask): a few thousand tracks instead of every edge
        // (neighbouring tracks mostly share their source frame and targets — the tracks of one frame: such a track adds nothing)
        // (the masks of a source frame's tracks are bits around the same base, src - 64: their union per frame, two words a track —
        //  a graph whose tracks all differ, observations missing at random, walked every target of a million tracks here)
        std::vector<uint64_t> fm((size_t)nw * 2, 0);
        for (int32_t k = 0; k < m; ++k) {
            const PerPatch &t = pp[pl->kx[(size_t)k]];
            fm[(size_t)(t.src - f_lo) * 2] |= t.mask; fm[(size_t)(t.src - f_lo) * 2 + 1] |= t.mask2;
        }
        for (int64_t f = 0; f < nw; ++f) {
            const PerPatch t{0, (int32_t)(f + f_lo), (int32_t)(f + f_lo) - 64, 0, fm[(size_t)f * 2], fm[(size_t)f * 2 + 1]};
            int32_t *row = pair_of.data() + (size_t)f * nw - f_lo;
            BT_FOR_TARGETS(t, fr, row[fr] = 0;);
        }
    } else {
        for (int64_t e = 0; e < E; ++e) if (owned(e)) pair_of[(size_t)((II(e) - f_lo) * nw + (JJ(e) - f_lo))] = 0;
    }
    pl->pair_i.clear(); pl->pair_j.clear();
    for (int64_t key = 0; key < nw * nw; ++key)
        if (pair_of[(size_t)key] == 0) {
            pair_of[(size_t)key] = (int32_t)pl->pair_i.size();
            pl->pair_i.push_back((int32_t)(key / nw + f_lo));
            pl->pair_j.push_back((int32_t)(key % nw + f_lo));
        }
    I.pairs = (int64_t)pl->pair_i.size();

    BT_TICK("3");
    // ---- edges grouped by track, ordered by (pair, original index) ---------
    // Two stable counting passes instead of a sort per track: first by target frame, then by track.  Within a
    // track the source frame is the same for all edges (checked above), so ascending target frame IS ascending
    // pair id, and stability keeps the original index as the tie-break (duplicates are normal, batrack.py:399-410).
    // (edge-sized temporaries persist per thread: the caller builds one plan per frame, and fresh pages cost more
    // than the passes over them)
    // (the packed words travel with the order — pks[q] = pk[ord[q]] — so that the per-tile walks below read the edges' frames
    //  in sequence instead of through ord into the caller's order: those random reads were most of the 0.9 ms of the walks)
    static thread_local std::vector<int32_t> ord_scratch, byj_scratch;
    static thread_local std::vector<uint64_t> pks_scratch;
    std::vector<int32_t> &ord = ord_scratch, &byj = byj_scratch;
    std::vector<uint64_t> &pks = pks_scratch;
    if (!dstats) { ord.resize((size_t)E_own + 1); byj.resize((size_t)E_own + 1); pks.resize((size_t)E_own + 1); }
    std::vector<int32_t> cur;
    if (!dstats) cur.assign(off.begin(), off.end() - 1);
    if (dstats) {
        // (no edge is read: the order of a track's edges is the device's sort)
    } else if (mono_j) {
        // every track's edges already come in (target frame, index) order: one stable scatter by track
        for (int64_t e = 0; e < E; ++e)
            if (E_own == E || owned(e)) { const int32_t q = cur[(size_t)trk_of(KK(e))]++; ord[(size_t)q] = (int32_t)e; pks[(size_t)q] = pk[e]; }
    } else {
        for (int64_t j = 0; j < n_all; ++j) cj[(size_t)j + 1] += cj[(size_t)j];
        if (E_own == E) {
            for (int64_t e = 0; e < E; ++e) byj[(size_t)cj[(size_t)JJ(e)]++] = (int32_t)e;
        } else {
            for (int64_t e = 0; e < E; ++e) if (owned(e)) byj[(size_t)cj[(size_t)JJ(e)]++] = (int32_t)e;
        }
        for (int64_t q = 0; q < E_own; ++q) {
            const int32_t e = byj[(size_t)q];
            const int32_t pos = cur[(size_t)trk_of(KK(e))]++;
            ord[(size_t)pos] = e; pks[(size_t)pos] = pk[e];
        }
    }
    auto IQ = [&](int64_t q) { return (int64_t)((pks[(size_t)q] >> 16) & 0xffff); };      // frames of the q-th edge of the grouped order
    auto JQ = [&](int64_t q) { return (int64_t)(pks[(size_t)q] & 0xffff); };
    auto pair_q = [&](int64_t q) { return pair_of[(size_t)((IQ(q) - f_lo) * nw + (JQ(q) - f_lo))]; };

    BT_TICK("4");
    // ---- tiles: greedy over sorted tracks ----------------------------------
    pl->tile_trk0.clear(); pl->tile_ntrk.clear(); pl->tile_ncam.clear(); pl->tile_cam0.clear();
    pl->tile_slot0.clear(); pl->tile_nslot.clear(); pl->tile_erow0.clear(); pl->tile_cams.clear();
    pl->trk_loc.assign((size_t)m, 0);
    std::vector<int32_t> stamp((size_t)n + 1, -1);      // last tile-epoch that contains camera c
    std::vector<int32_t> tstamp((size_t)n + 1, -1);     // last track that contains camera c
    std::vector<int32_t> tile_set, trk_set;
    int32_t epoch = 0, trk0 = 0;
    int64_t slots = 0, erows = 0;
    int max_cams = 0;
    auto close_tile = [&](int32_t trk_end) {
        if (trk_end == trk0) return;
        std::sort(tile_set.begin(), tile_set.end());
        const int32_t T = (int32_t)pl->tile_trk0.size();
        int32_t nslot = 0;
        for (int32_t k = trk0; k < trk_end; ++k) {
            nslot = std::max(nslot, off[(size_t)k + 1] - off[(size_t)k]);
            pl->trk_loc[(size_t)k] = (T << 6) | (k - trk0);
        }
        pl->tile_trk0.push_back(trk0); pl->tile_ntrk.push_back(trk_end - trk0);
        pl->tile_ncam.push_back((int32_t)tile_set.size());
        pl->tile_cam0.push_back((int32_t)pl->tile_cams.size());
        pl->tile_cams.insert(pl->tile_cams.end(), tile_set.begin(), tile_set.end());
        pl->tile_slot0.push_back((int32_t)slots); pl->tile_nslot.push_back(nslot);
        pl->tile_erow0.push_back((int32_t)erows);
        slots += nslot; erows += 6 * (int64_t)tile_set.size();
        max_cams = std::max(max_cams, (int)tile_set.size());
        tile_set.clear(); ++epoch; trk0 = trk_end;
    };
    // Tracks per tile: 64 (one per lane of k_tile), or 16 for the graphs of FEW tiles with DEEP edge lists per track — a sliding
    // window (batrack.py:399-410: ~2,500 tracks x ~54 edges) is 40 tiles of 64 tracks on a 256-CU part; the pair-major kernel
    // (k_etile) does not tie a lane to a track, so the same graph is spread over four times as many workgroups.  Plans whose
    // tiles turn out not to fit that kernel are rebuilt with 64 (tcap_retry).
    static thread_local int tcap_retry = 0;
    struct RetryScope {                                            // sets the flag for a nested layout pass; unwinds with it (bad_alloc)
        int &f; explicit RetryScope(int &x) : f(x) { f = 1; } ~RetryScope() { f = 0; }
    };
    int tcap = kLanes;
    {
        const int pm_env = etile_mode();
        const int64_t t64 = (m + kLanes - 1) / kLanes;
        if (pm_env && !tcap_retry && t64 > 0 && t64 <= 96 && E_own >= 24 * (int64_t)m && t64 < std::min(em_min_p, st_min_p) / 4) tcap = 16;
    }
    if (dstats && tcap == kLanes) {                                // the device then writes the [slots][64] arrays and the wave cuts
        if (etile_mode() == 2) return BT_NEED_EDGES;                    // (pair-major tables of 64-track tiles are made from the host's slot arrays)
    }
    // ALIGNED SLOTS (round 6) for the plans the wave-per-tile kernels take.  k_edge2 / k_edge2u need every tile SLOT-UNIFORM — all
    // tracks of a tile with the same camera pair (or no edge) in slot s.  With slot s = a track's s-th edge that holds only where
    // all tracks of a tile have the same observations: one missing observation shifts the rest of its track (a graph with a random
    // 15 % of its observations dropped fell to k_stream: three times the time per edge).  So a tile's slots are laid out per PAIR:
    // local pair p gets mult[p] consecutive slots from pbase[p], mult[p] = the largest number of edges a track of the tile has (or,
    // from the tracks' figures alone, can have) with that pair; a track's r-th edge with pair p sits in slot pbase[p] + r, the
    // slots it does not fill are null (edge -1: zero weight in every kernel).  Every tile is then slot-uniform by construction at
    // sum(mult) slots instead of the longest track's count — 1 / (1 - drop) of the uniform cost.  Tiles do not run across two
    // source frames (the pairs of one are no pairs of the other; the kernels also take a tile's source camera from its first track).
    // Where it does not fit (more than 64 slots in some tile) the plan is laid out again the old way.
    static thread_local int align_retry = 0;
    const bool aligned = !align_retry && masks_ok && tcap == kLanes && m > 0 && ((int64_t)m + kLanes - 1) / kLanes >= em_min_p &&
                         (!dstats || dstats->rep_known);
    pl->dev_pbase.clear();
    // LOOSE tracks: seen by more free cameras than a tile takes.  64 is what the kernels' tables take (kTileCamHard) — but a tile of
    // more than 32 cameras no longer holds its E in LDS as double, and ONE such tile turns the whole plan's per-edge maths to float32
    // (bt_plan_edge_precision).  A landmark seen from 40 or 60 frames next to ordinary tracks — a handful per plan — is therefore
    // loose as well (ba_loose.hip walks it in double; the plan stays float64: update errors 1e-6 instead of 1e-4 .. 1e-3 on such
    // graphs, tests/test_gpu_fuzz.py); a plan with MANY tracks that long (more than kFewHubs) keeps them in tiles — there the loose
    // path's atomics would cost more than float32 does — and is laid out again with the threshold at 64.
    static thread_local int hub_retry = 0;
    const int cam_hard = hub_retry ? kTileCamHard : kTileCamF64;
    std::vector<int32_t> loose;                                                               // tracks seen by more than cam_hard free cameras
    std::vector<uint64_t> tile_tmask;                                                         // aligned: per tile the union of its tracks' target masks (2 words)
    if (aligned) {
        // the same greedy rule on bit masks (a tile has one source frame, so its tracks' masks share their base): cameras = the
        // free frames among the targets and the source frame (bit 64), in ascending order as the bits are
        int32_t cur_src = -1;
        uint64_t free_lo = 0, free_hi = 0, tm_lo = 0, tm_hi = 0, tg_lo = 0, tg_hi = 0, pm_lo = ~0ull, pm_hi = ~0ull;
        auto flush = [&](int32_t k_end) {
            if (k_end == trk0) return;
            tile_set.clear();
            const int64_t base = (int64_t)cur_src - 64;
            for (uint64_t mk = tm_lo; mk; mk &= mk - 1) tile_set.push_back((int32_t)(base + __builtin_ctzll(mk) - fixedp));
            for (uint64_t mk = tm_hi; mk; mk &= mk - 1) tile_set.push_back((int32_t)(base + 64 + __builtin_ctzll(mk) - fixedp));
            tile_tmask.push_back(tg_lo); tile_tmask.push_back(tg_hi);
            close_tile(k_end);
            tm_lo = tm_hi = tg_lo = tg_hi = 0;
        };
        for (int32_t k = 0; k < m; ++k) {
            const PerPatch &t = pp[pl->kx[(size_t)k]];
            if (t.src != cur_src) {
                flush(k);
                cur_src = t.src; pm_lo = pm_hi = ~0ull;
                const int64_t first_free = fixedp - ((int64_t)t.src - 64);          // bits >= this are free frames
                free_lo = first_free <= 0 ? ~0ull : first_free >= 64 ? 0ull : ~0ull << first_free;
                free_hi = first_free <= 64 ? ~0ull : first_free >= 128 ? 0ull : ~0ull << (first_free - 64);
            }
            if (k > trk0 && k - trk0 < tcap && t.mask == pm_lo && t.mask2 == pm_hi) continue;      // (the same targets as the track before: nothing new)
            pm_lo = t.mask; pm_hi = t.mask2;
            const uint64_t c_lo = t.mask & free_lo, c_hi = (t.mask2 & free_hi) | (t.src >= fixedp ? 1ull : 0ull);
            const int nk = __builtin_popcountll(c_lo) + __builtin_popcountll(c_hi);
            if (nk > cam_hard) {                                   // a loose track (see the general loop below)
                if (dstats) return BT_NEED_EDGES;
                flush(k);
                loose.push_back(k);
                pl->trk_loc[(size_t)k] = -1;
                trk0 = k + 1;
                continue;
            }
            const int add = __builtin_popcountll(c_lo & ~tm_lo) + __builtin_popcountll(c_hi & ~tm_hi);
            const int have = __builtin_popcountll(tm_lo) + __builtin_popcountll(tm_hi);
            if (k - trk0 >= tcap || (k > trk0 && have + add > std::max<int>(kTileCamSoft, nk))) flush(k);
            tm_lo |= c_lo; tm_hi |= c_hi; tg_lo |= t.mask; tg_hi |= t.mask2;
        }
        flush(m);
    }
    int32_t src_p = -1, base_p = 0, set_epoch = -1; uint64_t mask_p = 0, mask2_p = 0;         // the previous track's figures: the same again = the same cameras
    // A tile's camera PAIRS (source frame -> target frame, the fixed frames too: every pair's geometry sits in the tile's LDS table)
    // are bounded as well: kMaxTilePairs.  The cameras bound them only among the FREE frames — with most of a long trajectory fixed,
    // 64 tracks of few free cameras each name hundreds of pairs.  Counted per run of tracks with one source frame (tracks come
    // sorted by patch and a frame's patches are neighbours; a source frame that comes back counts again: the bound may close a
    // tile early, never late).
    std::vector<int32_t> pair_stamp(aligned ? (size_t)0 : (size_t)n_buf, -1);
    int32_t pair_run = 0, run_src = -1;
    int tile_pairs = 0;
    auto mark_targets = [&](int32_t k) -> int {                    // the track's targets not yet seen in this run (and now seen)
        int cnt = 0;
        if (masks_ok) {
            const PerPatch &t = pp[pl->kx[(size_t)k]];
            BT_FOR_TARGETS(t, fr, if (pair_stamp[(size_t)fr] != pair_run) { pair_stamp[(size_t)fr] = pair_run; ++cnt; });
        } else {
            for (int32_t sidx = off[(size_t)k]; sidx < off[(size_t)k + 1]; ++sidx) {
                const int64_t fr = JQ(sidx);
                if (pair_stamp[(size_t)fr] != pair_run) { pair_stamp[(size_t)fr] = pair_run; ++cnt; }
            }
        }
        return cnt;
    };
    for (int32_t k = 0; k < m && !aligned; ++k) {
        if (masks_ok && k > 0 && pp[pl->kx[(size_t)k]].src == src_p && pp[pl->kx[(size_t)k]].base == base_p && pp[pl->kx[(size_t)k]].mask == mask_p &&
            pp[pl->kx[(size_t)k]].mask2 == mask2_p) {
            // (trk_set is the previous track's and still right; in the tile it went into, it adds nothing)
            if (set_epoch == epoch && k - trk0 < tcap) continue;
        } else {
        trk_set.clear();
        if (masks_ok) {
            // the track's free cameras from its source frame and target mask (no walk over its edges)
            const PerPatch &t = pp[pl->kx[(size_t)k]];
            src_p = t.src; base_p = t.base; mask_p = t.mask; mask2_p = t.mask2;
            const int64_t cs = (int64_t)t.src - fixedp;
            if (cs >= 0) { tstamp[(size_t)cs] = k; trk_set.push_back((int32_t)cs); }
            BT_FOR_TARGETS(t, fr,
                const int64_t c = (int64_t)fr - fixedp;
                if (c >= 0 && tstamp[(size_t)c] != k) { tstamp[(size_t)c] = k; trk_set.push_back((int32_t)c); });
        } else {
            for (int32_t sidx = off[(size_t)k]; sidx < off[(size_t)k + 1]; ++sidx) {
                const int64_t cams[2] = { IQ(sidx) - fixedp, JQ(sidx) - fixedp };
                for (int64_t c : cams)
                    if (c >= 0 && tstamp[(size_t)c] != k) { tstamp[(size_t)c] = k; trk_set.push_back((int32_t)c); }
            }
        }
        }
        const int32_t src_k = masks_ok ? pp[pl->kx[(size_t)k]].src : (int32_t)IQ(off[(size_t)k]);
        if (src_k != run_src) { run_src = src_k; ++pair_run; }
        int np_new = mark_targets(k);
        bool too_many_pairs = false;                               // (the track alone: possible only without masks — they hold 128 targets)
        if (!masks_ok && off[(size_t)k + 1] - off[(size_t)k] > kMaxTilePairs) { ++pair_run; too_many_pairs = mark_targets(k) > kMaxTilePairs; }
        if ((int)trk_set.size() > cam_hard || too_many_pairs) {
            // a LOOSE track: in no tile (its E does not fit a tile's camera budget, or its pairs a tile's table); ba_loose.hip walks its edges
            if (dstats) return BT_NEED_EDGES;                      // (their edge lists come from the host's grouped order)
            if (tcap < kLanes) {                                   // (k_etile's stored per-tile sums have no place for them: 64-track tiles, atomics)
                if (tcap_retry) return BT_EUNSUPPORTED;
                RetryScope guard(tcap_retry);
                return build_plan_host(ii64, jj64, kk64, E, n_buf, p_tot, fixedp, n_all_min, own_lo, own_hi, pl, packed, keep_slots, dstats);
            }
            close_tile(k);
            loose.push_back(k);
            pl->trk_loc[(size_t)k] = -1;
            trk0 = k + 1;
            src_p = -1;                                            // (the next track starts a tile: its cameras are looked at afresh)
            tile_pairs = 0; ++pair_run;
            continue;
        }
        int add = 0;
        for (int32_t c : trk_set) if (stamp[(size_t)c] != epoch) ++add;
        const int limit = std::max<int>(kTileCamSoft, (int)trk_set.size());
        if (k - trk0 >= tcap || (k > trk0 && ((int)tile_set.size() + add > limit || tile_pairs + np_new > kMaxTilePairs))) {
            close_tile(k);
            tile_pairs = 0; ++pair_run;
            np_new = mark_targets(k);                              // (all of the track's pairs are new to the tile it opens)
        }
        tile_pairs += np_new;
        for (int32_t c : trk_set) if (stamp[(size_t)c] != epoch) { stamp[(size_t)c] = epoch; tile_set.push_back(c); }
        set_epoch = epoch;
    }
    if (!aligned) close_tile(m);
    if (!hub_retry && (int)loose.size() > kFewHubs) {              // (many long tracks: tiles of up to 64 cameras, float32 per edge)
        RetryScope guard(hub_retry);
        return build_plan_host(ii64, jj64, kk64, E, n_buf, p_tot, fixedp, n_all_min, own_lo, own_hi, pl, packed, keep_slots, dstats);
    }
    const int32_t T = (int32_t)pl->tile_trk0.size();
    I.tiles = T; I.slots = slots; I.erows = erows; I.max_tile_cams = max_cams;
    pl->max_rows16 = (int)((6 * max_cams + 15) / 16 * 16);
    if (tcap < kLanes && T >= std::min(std::min(em_min_p, st_min_p), 1024)) {
        // the camera limit closed 16-track tiles early: the plan reached the tile count of the wave-per-tile kernels, whose
        // tables come from the [slots][64] arrays a small-tile plan does not have — or simply four times the CUs, where
        // spreading a graph over more workgroups has lost its point: lay it out again with 64 tracks per tile
        if (tcap_retry) return dstats ? (int)BT_NEED_EDGES : (int)BT_EUNSUPPORTED;
        RetryScope guard(tcap_retry);
        return build_plan_host(ii64, jj64, kk64, E, n_buf, p_tot, fixedp, n_all_min, own_lo, own_hi, pl, packed, keep_slots, dstats);
    }

    BT_TICK("6");
    // ---- per tile: its distinct camera pairs (their relative pose is computed once per tile), then the slot arrays
    // [slots][64] with the local pair index of every edge, in one walk over the tile's edges
    // (a plan tiled for k_etile — tcap < 64 — gets its pair-major tables straight from the tracks' edge lists here; the
    //  [slots][64] arrays of k_tile, three quarters of them padding at 16 tracks per tile, are then built only on request:
    //  host-only plans, which the tests' emulator executes)
    const bool dev_slots = dstats && tcap == kLanes;            // the slot arrays are written on the device (plan_device.hip)
    const bool want_slots = (tcap == kLanes || keep_slots) && !dev_slots;
    const bool pm_direct = tcap < kLanes;
    bool pm_fail = false;
    int64_t pm_rounds_acc = 0;
    std::vector<int32_t> pm_cnt;
    static thread_local std::vector<uint32_t> pm_code;
    if (pm_direct) {
        pl->pm_rec.assign((size_t)T * 4, 0);
        pl->pm_lb.assign((size_t)T * kLanes, 0xff);
        pl->pm_la.assign((size_t)T * kLanes, 0xff);
        pl->pm_edge.clear();
        pl->pm_edge.reserve((size_t)E_own * 2 + (size_t)T * kLanes * 4);      // (one allocation: the table grows tile by tile)
    }
    const size_t slot_elems = want_slots && !aligned ? (size_t)slots * kLanes : 0;      // (aligned: the arrays grow tile by tile below)
    pl->slot_edge.assign(slot_elems, -1);
    pl->slot_pair.assign(slot_elems, 0);
    pl->slot_lab.assign(slot_elems, 0xffff);
    pl->slot_lp.assign(slot_elems, 0);
    pl->tile_pair0.assign((size_t)T, 0); pl->tile_npair.assign((size_t)T, 0);
    pl->tile_pairs.clear();
    pl->max_tile_pairs = 0;
    // tile_cut8 / tile_cut16: where k_tile's 8 or 16 waves split the tile's slots.  Repeated observations of one (track,
    // target camera) are consecutive slots of that track; a wave sums such a run in registers and writes it once, but a run
    // that continues into the next wave's chunk forces LDS float atomics on every slot of it (~500 cycles per wave
    // instruction on this part).  The sliding-window caller repeats observations all the time and all tracks of a tile
    // share their pattern, so the cuts are moved to the nearest slot boundary that no track's run crosses.
    pl->tile_cut8.assign((size_t)T * 9, 0);
    pl->tile_cut16.assign((size_t)T * 17, 0);
    {
        std::vector<int32_t> local((size_t)n + 1, -1), lp_of(pl->pair_i.size(), -1), mine, mult, pbase;
        std::vector<uint8_t> crossed;                              // crossed[s]: some track's run spans slots s - 1 and s
        int64_t slots_al = 0;                                      // aligned layout: the slots of the tiles so far
        for (int32_t t = 0; t < T; ++t) {
            const int32_t c0 = pl->tile_cam0[(size_t)t], nc = pl->tile_ncam[(size_t)t], t0 = pl->tile_trk0[(size_t)t], nt = pl->tile_ntrk[(size_t)t];
            int32_t ns_t = pl->tile_nslot[(size_t)t];
            if (ns_t > 0xffff) return BT_EUNSUPPORTED;           // (a track with more than 65535 observations)
            if (!aligned) crossed.assign((size_t)ns_t + 1, 0);
            for (int32_t c = 0; c < nc; ++c) local[(size_t)pl->tile_cams[(size_t)(c0 + c)]] = c;
            mine.clear();
            if (aligned) {
                // (one source frame per tile: its pairs are (source, target) over the union of the tracks' target masks, ascending)
                const int32_t src_t = pp[pl->kx[(size_t)t0]].src;
                const PerPatch tu{0, src_t, src_t - 64, 0, tile_tmask[(size_t)t * 2], tile_tmask[(size_t)t * 2 + 1]};
                const int32_t *row = pair_of.data() + (size_t)(src_t - f_lo) * nw - f_lo;
                BT_FOR_TARGETS(tu, fr, mine.push_back(row[fr]););
                for (int32_t gp : mine) lp_of[(size_t)gp] = 0;
            } else if (masks_ok) {
                int32_t src_b = -1, base_b = 0; uint64_t mask_b = 0, mask2_b = 0;
                for (int32_t l = 0; l < nt; ++l) {
                    const PerPatch &tp = pp[pl->kx[(size_t)(t0 + l)]];
                    if (tp.src == src_b && tp.base == base_b && tp.mask == mask_b && tp.mask2 == mask2_b) continue;       // (the same pairs as the track before)
                    src_b = tp.src; base_b = tp.base; mask_b = tp.mask; mask2_b = tp.mask2;
                    const int32_t *row = pair_of.data() + (size_t)(tp.src - f_lo) * nw - f_lo;
                    BT_FOR_TARGETS(tp, fr,
                        const int32_t gp = row[fr];
                        if (lp_of[(size_t)gp] < 0) { lp_of[(size_t)gp] = 0; mine.push_back(gp); });
                }
            } else {
                for (int32_t q = off[(size_t)t0]; q < off[(size_t)(t0 + nt)]; ++q) {
                    const int32_t gp = pair_q(q);
                    if (lp_of[(size_t)gp] < 0) { lp_of[(size_t)gp] = 0; mine.push_back(gp); }
                }
            }
            std::sort(mine.begin(), mine.end());
            if ((int)mine.size() > kMaxTilePairs) return BT_EUNSUPPORTED;
            for (size_t q = 0; q < mine.size(); ++q) lp_of[(size_t)mine[q]] = (int32_t)q;
            if (aligned) {
                // slots per pair from the tracks' figures (no edge is read): a target outside the track's repeat mask has one
                // edge; the repeated ones share the track's surplus, cnt - (distinct targets), each holding at least one of it
                // (every pair of the tile has a slot; only the tracks that repeat a target are looked at bit by bit)
                mult.assign(mine.size(), 1);
                for (int32_t l = 0; l < nt; ++l) {
                    const PerPatch &tp = pp[pl->kx[(size_t)(t0 + l)]];
                    const int32_t distinct = __builtin_popcountll(tp.mask) + __builtin_popcountll(tp.mask2);
                    if (tp.cnt <= distinct) continue;
                    RepStat rs{0, 0};
                    const int64_t patch = pl->kx[(size_t)(t0 + l)];
                    if (dstats) { if (dstats->rtab) rs = dstats->rtab[(size_t)(patch - dstats->tab_lo)]; }
                    else { const auto it = rep_host.find(patch); if (it != rep_host.end()) rs = it->second; }
                    rs.rmask &= tp.mask; rs.rmask2 &= tp.mask2;
                    const int32_t nrep = __builtin_popcountll(rs.rmask) + __builtin_popcountll(rs.rmask2);
                    if (nrep == 0) { pl->dev_pbase.clear(); return BT_EINVAL; }      // (cannot happen: a surplus edge repeats some target)
                    const int32_t mrep = 1 + (tp.cnt - distinct) - (nrep - 1);
                    const int32_t *row = pair_of.data() + (size_t)(tp.src - f_lo) * nw - f_lo;
                    for (int half = 0; half < 2; ++half)
                        for (uint64_t mk = half ? rs.rmask2 : rs.rmask; mk; mk &= mk - 1) {
                            int32_t &mu = mult[(size_t)lp_of[(size_t)row[tp.base + 64 * half + __builtin_ctzll(mk)]]];
                            mu = std::max(mu, mrep);
                        }
                }
                pbase.assign(mine.size(), 0);
                int32_t acc = 0;
                for (size_t q = 0; q < mine.size(); ++q) { pbase[q] = acc; acc += mult[q]; }
                if (acc > kLanes) {
                    // (more (pair, repeat) slots than an iteration has lanes: the edge-major layout does not hold this tile)
                    RetryScope guard(align_retry);
                    return build_plan_host(ii64, jj64, kk64, E, n_buf, p_tot, fixedp, n_all_min, own_lo, own_hi, pl, packed, keep_slots, dstats);
                }
                ns_t = acc;
                pl->tile_slot0[(size_t)t] = (int32_t)slots_al; pl->tile_nslot[(size_t)t] = ns_t;
                slots_al += ns_t;
                pl->dev_pbase.insert(pl->dev_pbase.end(), pbase.begin(), pbase.end());
                crossed.assign((size_t)ns_t + 1, 0);
                if (want_slots) {
                    pl->slot_edge.resize((size_t)slots_al * kLanes, -1); pl->slot_pair.resize((size_t)slots_al * kLanes, 0);
                    pl->slot_lab.resize((size_t)slots_al * kLanes, 0xffff); pl->slot_lp.resize((size_t)slots_al * kLanes, 0);
                }
            }
            if (pm_direct && !pm_fail) {
                // pair-major tables of this tile (layout: the block further down that builds them from the slot arrays)
                const int32_t np = (int32_t)mine.size();
                if (np > kLanes) pm_fail = true;
                else if (dstats) {
                    // the tile's record but for its rounds, the local target camera of its pairs and the local source camera
                    // of its tracks; the table itself and the rounds come from the device
                    int lg = 0;
                    while ((1 << lg) < np) ++lg;
                    const int32_t G = kLanes >> lg;
                    pl->pm_rec[(size_t)t * 4 + 1] = lg;
                    pl->pm_rec[(size_t)t * 4 + 2] = (nt + G - 1) / G;
                    for (int32_t q = 0; q < np; ++q) {
                        const int64_t b2 = (int64_t)pl->pair_j[(size_t)mine[(size_t)q]] - fixedp;
                        pl->pm_lb[(size_t)t * kLanes + (size_t)q] = b2 >= 0 ? (uint8_t)local[(size_t)b2] : (uint8_t)0xff;
                    }
                    for (int32_t l = 0; l < nt; ++l) {
                        const int64_t a2 = (int64_t)pp[pl->kx[(size_t)(t0 + l)]].src - fixedp;
                        pl->pm_la[(size_t)t * kLanes + (size_t)l] = a2 >= 0 ? (uint8_t)local[(size_t)a2] : (uint8_t)0xff;
                    }
                }
                else {
                    int lg = 0;
                    while ((1 << lg) < np) ++lg;
                    const int32_t S = 1 << lg, G = kLanes >> lg, nit = (nt + G - 1) / G;
                    pm_cnt.assign((size_t)nt * (size_t)S, 0);
                    // one walk over the tile's edges (each a random access into the packed edge list): local pair, local
                    // cameras and the round of every edge; the table is filled from these codes
                    const int32_t q0 = off[(size_t)t0], q1 = off[(size_t)(t0 + nt)];
                    pm_code.resize((size_t)(q1 - q0));
                    int32_t D = 1;
                    for (int32_t l = 0; l < nt; ++l)
                        for (int32_t sidx = off[(size_t)(t0 + l)]; sidx < off[(size_t)(t0 + l) + 1]; ++sidx) {
                            const int64_t i = IQ(sidx), j = JQ(sidx), a2 = i - fixedp, b2 = j - fixedp;
                            const int32_t lp = lp_of[(size_t)pair_of[(size_t)((i - f_lo) * nw + (j - f_lo))]];
                            const int32_t d = pm_cnt[(size_t)l * S + lp]++;
                            D = std::max(D, d + 1);
                            const uint32_t la = a2 >= 0 ? (uint32_t)local[(size_t)a2] : 0xffu, lb = b2 >= 0 ? (uint32_t)local[(size_t)b2] : 0xffu;
                            pm_code[(size_t)(sidx - q0)] = (uint32_t)lp | (la << 8) | (lb << 16) | ((uint32_t)std::min(d, 255) << 24);
                        }
                    if (D > 255) pm_fail = true;
                    else {
                        pl->pm_rec[(size_t)t * 4] = (int32_t)pm_rounds_acc;
                        pl->pm_rec[(size_t)t * 4 + 1] = lg | (D << 8);
                        pl->pm_rec[(size_t)t * 4 + 2] = nit;
                        pl->pm_edge.resize((size_t)(pm_rounds_acc + (int64_t)nit * D) * kLanes, -1);
                        for (int32_t l = 0; l < nt; ++l)
                            for (int32_t sidx = off[(size_t)(t0 + l)]; sidx < off[(size_t)(t0 + l) + 1]; ++sidx) {
                                const uint32_t c = pm_code[(size_t)(sidx - q0)];
                                const int32_t lp = (int32_t)(c & 0xffu), d = (int32_t)(c >> 24);
                                pl->pm_edge[((size_t)pm_rounds_acc + (size_t)(l / G) * D + (size_t)d) * kLanes + (size_t)((l % G) << lg) + (size_t)lp] = ord[(size_t)sidx];
                                pl->pm_lb[(size_t)t * kLanes + (size_t)lp] = (uint8_t)((c >> 16) & 0xffu);
                                pl->pm_la[(size_t)t * kLanes + (size_t)l] = (uint8_t)((c >> 8) & 0xffu);
                            }
                        pm_rounds_acc += (int64_t)nit * D;
                    }
                }
            }
            for (int32_t l = 0; l < nt && want_slots; ++l) {
                const int32_t k = t0 + l;
                uint16_t lb_before = 0xff;
                int32_t gp_before = -1, rank = 0;                  // aligned layout: the edge's rank among the track's edges with its pair
                for (int32_t sidx = off[(size_t)k]; sidx < off[(size_t)k + 1]; ++sidx) {
                    const int32_t e = ord[(size_t)sidx];
                    const int64_t i = IQ(sidx), j = JQ(sidx);
                    const int64_t a = i - fixedp, b = j - fixedp;
                    const uint16_t la = a >= 0 ? (uint16_t)local[(size_t)a] : 0xff;
                    const uint16_t lb = b >= 0 ? (uint16_t)local[(size_t)b] : 0xff;
                    const int32_t gp = pair_of[(size_t)((i - f_lo) * nw + (j - f_lo))];
                    rank = gp == gp_before ? rank + 1 : 0;
                    gp_before = gp;
                    const int32_t sl = aligned ? pbase[(size_t)lp_of[(size_t)gp]] + rank : sidx - off[(size_t)k];
                    const size_t idx = ((size_t)pl->tile_slot0[(size_t)t] + (size_t)sl) * kLanes + (size_t)l;
                    pl->slot_edge[idx] = e;
                    pl->slot_pair[idx] = gp;
                    pl->slot_lab[idx] = (uint16_t)(la | (lb << 8));
                    pl->slot_lp[idx] = (uint8_t)lp_of[(size_t)gp];
                    if (lb != 0xff && lb == lb_before) crossed[(size_t)sl] = 1;
                    lb_before = lb;
                }
            }
            for (int W = 8; W <= 16 && want_slots; W += 8) {
                uint16_t *cut = (W == 8 ? pl->tile_cut8.data() + (size_t)t * 9 : pl->tile_cut16.data() + (size_t)t * 17);
                const int32_t chunk = (ns_t + W - 1) / W;
                int32_t prev = 0;
                cut[0] = 0;
                for (int w = 1; w < W; ++w) {
                    const int32_t ideal = std::min(ns_t, w * chunk);
                    int32_t best = ideal;
                    if (ideal < ns_t && crossed[(size_t)ideal]) {          // nearest uncrossed boundary within a chunk's reach, else keep it
                        for (int32_t d = 1; d <= chunk; ++d) {
                            if (ideal - d >= prev && !crossed[(size_t)(ideal - d)]) { best = ideal - d; break; }
                            if (ideal + d <= ns_t && (ideal + d == ns_t || !crossed[(size_t)(ideal + d)])) { best = ideal + d; break; }
                        }
                    }
                    best = std::max(best, prev);
                    cut[w] = (uint16_t)best;
                    prev = best;
                }
                cut[W] = (uint16_t)ns_t;
            }
            pl->tile_pair0[(size_t)t] = (int32_t)pl->tile_pairs.size();
            pl->tile_npair[(size_t)t] = (int32_t)mine.size();
            pl->tile_pairs.insert(pl->tile_pairs.end(), mine.begin(), mine.end());
            pl->max_tile_pairs = std::max(pl->max_tile_pairs, (int)mine.size());
            for (int32_t q : mine) lp_of[(size_t)q] = -1;
        }
    }

    if (aligned) {
        int64_t tot = 0;
        for (int32_t t = 0; t < T; ++t) tot += pl->tile_nslot[(size_t)t];
        slots = tot; I.slots = tot;
    }
    // the loose tracks' edge lists (grouped order: by pair, then by the caller's index) with their camera pairs
    pl->lz_trk.clear(); pl->lz_ptr.assign(1, 0); pl->lz_edge.clear(); pl->lz_pair.clear();
    for (int32_t k : loose) {
        pl->lz_trk.push_back(k);
        for (int32_t q = off[(size_t)k]; q < off[(size_t)k + 1]; ++q) { pl->lz_edge.push_back(ord[(size_t)q]); pl->lz_pair.push_back(pair_q(q)); }
        pl->lz_ptr.push_back((int32_t)pl->lz_edge.size());
    }
    // consecutive tiles with identical camera / pair lists: a persistent workgroup keeps its
    // accumulators across them
    pl->tile_flags.assign((size_t)T, 0);
    for (int32_t t = 1; t < T; ++t) {
        const int32_t nc = pl->tile_ncam[(size_t)t], np2 = pl->tile_npair[(size_t)t];
        if (nc == pl->tile_ncam[(size_t)t - 1] &&
            std::equal(pl->tile_cams.begin() + pl->tile_cam0[(size_t)t], pl->tile_cams.begin() + pl->tile_cam0[(size_t)t] + nc,
                       pl->tile_cams.begin() + pl->tile_cam0[(size_t)t - 1]))
            pl->tile_flags[(size_t)t] |= 1;
        if (np2 == pl->tile_npair[(size_t)t - 1] &&
            std::equal(pl->tile_pairs.begin() + pl->tile_pair0[(size_t)t], pl->tile_pairs.begin() + pl->tile_pair0[(size_t)t] + np2,
                       pl->tile_pairs.begin() + pl->tile_pair0[(size_t)t - 1]))
            pl->tile_flags[(size_t)t] |= 2;
    }

    BT_TICK("8");
    // More than kMaxFree (255) free poses: the block-sparse solvers' tables hold pose numbers in 8 bits and block numbers in 15.
    // Such a system (a global / loop-closing adjustment; the reference's dense solve has no size clause, ba.py:60-70) is solved
    // DENSE in the global workspace (ba_dense.hip): no symbolic factorisation here, the natural order, every lower block "non-zero".
    // The same solver takes the systems of FEWER poses whose block-sparse factor does not fit LDS as double (decided below, once
    // the symbolic factorisation has counted its blocks): a hundred poses tied by long-range edges fill in to a nearly dense factor,
    // which the block-sparse solvers can only hold as float32 (or in global memory) — 24 ms a solve at 179 poses and a float32
    // factor's precision, against the dense solver's 3 ms in double.
    const auto go_wide = [&]() {
        pl->wide = 1;
        pl->perm.resize((size_t)n);
        std::iota(pl->perm.begin(), pl->perm.end(), 0);
        pl->col_ptr.assign((size_t)n + 1, 0); pl->upd_ptr.assign((size_t)n + 1, 0); pl->upd_next.assign((size_t)n + 1, 0);
        pl->lvl_ptr.assign(1, 0); pl->dp_ptr.assign((size_t)n + 1, 0);
        for (auto *v : {&pl->row_idx, &pl->upd, &pl->blk_col, &pl->blk_src, &pl->lvl_cols, &pl->col_lvl, &pl->dp, &pl->lvl_meta, &pl->fz_pend_ptr,
                        &pl->fz_pend, &pl->fz_lazy_ptr, &pl->fz_lazy, &pl->fz_yurg, &pl->fz_meta, &pl->fz_pmeta, &pl->bs_sync, &pl->fz_rowinfo,
                        &pl->fz_pfirst, &pl->fz_psecond})
            v->clear();
        pl->fz_ok = 0; pl->fzp_ok = 0;
        I.nnz_blocks = n * (n + 1) / 2; I.updates = 0;
    };
    pl->wide = 0;
    if (n > kMaxFree) go_wide();
    const auto symbolic = [&]() -> int {
    // ---- block structure of S (lower) and symbolic Cholesky ----------------
    // S[u][v] (u >= v) may be non-zero if u and v share a tile (Schur term,
    // ba.py:321) or form a camera pair with both ends free (B, ba.py:279-282).
    std::vector<std::vector<uint8_t>> nz((size_t)n, std::vector<uint8_t>((size_t)n, 0));
    for (int64_t c = 0; c < n; ++c) nz[(size_t)c][(size_t)c] = 1;
    // One plan for the whole list: a TILE's cameras couple pairwise — more than its tracks' own couplings where the tracks of a tile
    // see different cameras, and free to have (the tile's product is formed over all of them).  A sharded plan must not: the tiles
    // are this rank's, the pattern is everybody's — there the tracks' own couplings only, of this rank's tracks as of the others'
    // (what a tile adds outside them is a sum of exact zeros: an E row of a camera a track does not see is zero).
    for (int32_t t = 0; t < T && !sharded; ++t) {
        const int32_t c0 = pl->tile_cam0[(size_t)t], nc = pl->tile_ncam[(size_t)t];
        for (int32_t u = 0; u < nc; ++u)
            for (int32_t v = 0; v <= u; ++v)
                nz[(size_t)pl->tile_cams[(size_t)(c0 + u)]][(size_t)pl->tile_cams[(size_t)(c0 + v)]] = 1;
    }
    for (size_t p = 0; p < pl->pair_i.size(); ++p) {
        const int64_t a = pl->pair_i[p] - fixedp, b = pl->pair_j[p] - fixedp;
        if (a >= 0 && b >= 0) nz[(size_t)std::max(a, b)][(size_t)std::min(a, b)] = 1;
    }
    {   // a loose track couples all its free cameras (its Schur term, ba.py:321)
        std::vector<int32_t> cset;
        for (size_t l = 0; l < pl->lz_trk.size(); ++l) {
            cset.clear();
            for (int32_t q = pl->lz_ptr[l]; q < pl->lz_ptr[l + 1]; ++q) {
                const int64_t a = pl->pair_i[(size_t)pl->lz_pair[(size_t)q]] - fixedp, b = pl->pair_j[(size_t)pl->lz_pair[(size_t)q]] - fixedp;
                if (a >= 0) cset.push_back((int32_t)a);
                if (b >= 0) cset.push_back((int32_t)b);
            }
            std::sort(cset.begin(), cset.end());
            cset.erase(std::unique(cset.begin(), cset.end()), cset.end());
            for (size_t u = 0; u < cset.size(); ++u)
                for (size_t v = 0; v <= u; ++v) nz[(size_t)cset[u]][(size_t)cset[v]] = 1;
        }
    }
    if (sharded && dstats && dstats->sliced) {
        // (the pattern of the whole list, reduced on the device — k_plan_pattern)
        if (!dstats->pattern) return BT_NEED_EDGES;
        for (int64_t u = 0; u < n; ++u)
            for (int64_t v = 0; v <= u; ++v)
                if ((dstats->pattern[(size_t)u * dstats->pattern_words + (size_t)(v >> 5)] >> (v & 31)) & 1u) nz[(size_t)u][(size_t)v] = 1;
    } else if (sharded && dstats) {
        // (the same from the device's table: a track's cameras are its source frame and the frames of its mask)
        std::vector<int32_t> cset;
        int32_t src_b = -1; uint64_t mask_b = 0, mask2_b = 0;
        for (int64_t p = kmin; p <= kmax; ++p) {
            const PatchStat &d = dstats->tab[(size_t)(p - dstats->tab_lo)];
            if (d.cnt <= 0) continue;
            if (d.src == src_b && d.mask == mask_b && d.mask2 == mask2_b) continue;          // (the same cameras as the track before: nothing new)
            src_b = d.src; mask_b = d.mask; mask2_b = d.mask2;
            cset.clear();
            if (d.src >= fixedp) cset.push_back((int32_t)(d.src - fixedp));
            const PerPatch dt{d.cnt, d.src, d.src - 64, 0, d.mask, d.mask2};
            BT_FOR_TARGETS(dt, fr,
                const int64_t c = (int64_t)fr - fixedp;
                if (c >= 0) cset.push_back((int32_t)c););
            std::sort(cset.begin(), cset.end());
            cset.erase(std::unique(cset.begin(), cset.end()), cset.end());
            for (size_t u = 0; u < cset.size(); ++u)
                for (size_t v = 0; v <= u; ++v) nz[(size_t)cset[u]][(size_t)cset[v]] = 1;
        }
    } else if (sharded) {
        // from the edges: all pairs among the free cameras of a track (the camera pair of an edge is among them), every track
        // of the list
        std::vector<int32_t> toff((size_t)p_tot + 1, 0);
        for (int64_t e = 0; e < E; ++e) toff[(size_t)KK(e) + 1]++;
        for (int64_t p = 0; p < p_tot; ++p) toff[(size_t)p + 1] += toff[(size_t)p];
        std::vector<int32_t> tord((size_t)E + 1), tcur(toff.begin(), toff.end() - 1);
        for (int64_t e = 0; e < E; ++e) tord[(size_t)tcur[(size_t)KK(e)]++] = (int32_t)e;
        std::vector<int32_t> cset;
        for (int64_t p = 0; p < p_tot; ++p) {
            if (toff[(size_t)p] == toff[(size_t)p + 1]) continue;
            cset.clear();
            for (int32_t q = toff[(size_t)p]; q < toff[(size_t)p + 1]; ++q) {
                const int32_t e = tord[(size_t)q];
                if (II(e) >= fixedp) cset.push_back((int32_t)(II(e) - fixedp));
                if (JJ(e) >= fixedp) cset.push_back((int32_t)(JJ(e) - fixedp));
            }
            std::sort(cset.begin(), cset.end());
            cset.erase(std::unique(cset.begin(), cset.end()), cset.end());
            for (size_t u = 0; u < cset.size(); ++u)
                for (size_t v = 0; v <= u; ++v) nz[(size_t)cset[u]][(size_t)cset[v]] = 1;
        }
    }
    BT_TICK("9");
    // ---- elimination order: two-ended ("twisted") when an index cut gives a small separator
    // Cameras are time-ordered and co-visibility is local in time, so a vertex separator is
    // looked for as S_k = { j >= k : some i < k is coupled to j }.  Order = [A = {<k} ascending |
    // B = {>= k} \ S_k DESCENDING | S_k ascending]: both parts are eliminated from their far end
    // towards the separator (no fill beyond their band), their columns are pairwise independent
    // and are factored two per level.  Without a useful cut the natural order is kept.
    std::vector<int32_t> perm((size_t)n);
    std::iota(perm.begin(), perm.end(), 0);
    {
        const int nd = force().natural_order ? 0 : 1;
        int best_k = -1, best_score = (int)n;        // natural chain length = n
        std::vector<uint8_t> in_s((size_t)n);
        for (int64_t k = 1; k < n && nd; ++k) {
            int ns = 0;
            for (int64_t j = k; j < n; ++j) {
                bool coupled = false;
                for (int64_t i = 0; i < k && !coupled; ++i) coupled = nz[(size_t)j][(size_t)i] != 0;
                ns += coupled ? 1 : 0;
            }
            const int na = (int)k, nb = (int)(n - k) - ns;
            if (nb < 1 || ns * 3 > (int)n) continue;
            const int score = std::max(na, nb) + ns;
            if (score < best_score) { best_score = score; best_k = (int)k; }
        }
        if (best_k > 0 && best_score + 2 < (int)n) {
            std::fill(in_s.begin(), in_s.end(), 0);
            for (int64_t j = best_k; j < n; ++j)
                for (int64_t i = 0; i < best_k; ++i) if (nz[(size_t)j][(size_t)i]) { in_s[(size_t)j] = 1; break; }
            size_t o = 0;
            for (int64_t i = 0; i < best_k; ++i) perm[o++] = (int32_t)i;
            for (int64_t j = n - 1; j >= best_k; --j) if (!in_s[(size_t)j]) perm[o++] = (int32_t)j;
            for (int64_t j = best_k; j < n; ++j) if (in_s[(size_t)j]) perm[o++] = (int32_t)j;
        }
    }
    pl->perm = perm;
    auto nzp = [&](int64_t a, int64_t b) {            // pattern in the permuted numbering
        const int64_t x = perm[(size_t)a], y = perm[(size_t)b];
        return nz[(size_t)std::max(x, y)][(size_t)std::min(x, y)] != 0;
    };

    // column structures (rows > j), then fill: struct(parent(j)) |= struct(j) \ {parent(j)}
    std::vector<std::vector<int32_t>> cs((size_t)n);
    for (int64_t j = 0; j < n; ++j) {
        for (int64_t r = j + 1; r < n; ++r) if (nzp(r, j)) cs[(size_t)j].push_back((int32_t)r);
    }
    std::vector<int32_t> parent((size_t)n, -1);
    for (int64_t j = 0; j < n; ++j) {
        auto &sj = cs[(size_t)j];
        std::sort(sj.begin(), sj.end());
        sj.erase(std::unique(sj.begin(), sj.end()), sj.end());
        if (sj.empty()) continue;
        parent[(size_t)j] = sj[0];
        auto &sp = cs[(size_t)sj[0]];
        sp.insert(sp.end(), sj.begin() + 1, sj.end());
    }
    pl->col_ptr.assign((size_t)n + 1, 0);
    pl->row_idx.clear();
    for (int64_t j = 0; j < n; ++j) {
        pl->col_ptr[(size_t)j] = (int32_t)pl->row_idx.size();
        pl->row_idx.push_back((int32_t)j);
        pl->row_idx.insert(pl->row_idx.end(), cs[(size_t)j].begin(), cs[(size_t)j].end());
    }
    pl->col_ptr[(size_t)n] = (int32_t)pl->row_idx.size();
    I.nnz_blocks = (int64_t)pl->row_idx.size();
    if (I.nnz_blocks >= 32768) return BT_EUNSUPPORTED;
    auto find_pos = [&](int32_t col, int32_t row) {
        const auto b = pl->row_idx.begin() + pl->col_ptr[(size_t)col], e = pl->row_idx.begin() + pl->col_ptr[(size_t)col + 1];
        return (int32_t)(std::lower_bound(b, e, row) - pl->row_idx.begin());
    };

    // block -> column map and where each block comes from in the caller-order lower triangle of S:
    // blk_src = (row_nat << 9) | (col_nat << 1) | transposed
    pl->blk_col.assign(pl->row_idx.size(), 0);
    pl->blk_src.assign(pl->row_idx.size(), 0);
    for (int64_t j = 0; j < n; ++j)
        for (int32_t b = pl->col_ptr[(size_t)j]; b < pl->col_ptr[(size_t)j + 1]; ++b) {
            pl->blk_col[(size_t)b] = (int32_t)j;
            const int32_t x = perm[(size_t)pl->row_idx[(size_t)b]], y = perm[(size_t)j];
            pl->blk_src[(size_t)b] = x >= y ? ((x << 9) | (y << 1)) : ((y << 9) | (x << 1) | 1);
        }

    BT_TICK("10");
    // ---- level schedule: level(j) = 1 + max level of its children in the elimination tree;
    // columns of one level are independent.  At most kMaxLevelCols per level (extra ones move up).
    std::vector<int32_t> lvl((size_t)n, 0);
    for (int64_t j = 0; j < n; ++j)
        if (parent[(size_t)j] >= 0) lvl[(size_t)parent[(size_t)j]] = std::max(lvl[(size_t)parent[(size_t)j]], lvl[(size_t)j] + 1);
    {
        // enforce the per-level cap, keeping level(parent) > level(child)
        std::vector<int32_t> cnt;
        for (int64_t j = 0; j < n; ++j) {
            int32_t l = lvl[(size_t)j];
            for (;;) {
                if ((size_t)l >= cnt.size()) cnt.resize((size_t)l + 1, 0);
                if (cnt[(size_t)l] < kMaxLevelCols) break;
                ++l;
            }
            cnt[(size_t)l]++;
            lvl[(size_t)j] = l;
            if (parent[(size_t)j] >= 0) lvl[(size_t)parent[(size_t)j]] = std::max(lvl[(size_t)parent[(size_t)j]], l + 1);
        }
    }
    // every column a column updates must sit at a strictly higher level
    for (int64_t j = 0; j < n; ++j)
        for (int32_t b = pl->col_ptr[(size_t)j] + 1; b < pl->col_ptr[(size_t)j + 1]; ++b)
            if (lvl[(size_t)pl->row_idx[(size_t)b]] <= lvl[(size_t)j]) return BT_EINVAL;   // cannot happen: rows are ancestors
    int32_t nlev = 0;
    for (int64_t j = 0; j < n; ++j) nlev = std::max(nlev, lvl[(size_t)j] + 1);
    pl->col_lvl = lvl;
    pl->lvl_ptr.assign((size_t)nlev + 1, 0);
    for (int64_t j = 0; j < n; ++j) pl->lvl_ptr[(size_t)lvl[(size_t)j] + 1]++;
    for (int32_t l = 0; l < nlev; ++l) pl->lvl_ptr[(size_t)l + 1] += pl->lvl_ptr[(size_t)l];
    pl->lvl_cols.assign((size_t)n, 0);
    {
        std::vector<int32_t> cur(pl->lvl_ptr.begin(), pl->lvl_ptr.end() - 1);
        for (int64_t j = 0; j < n; ++j) pl->lvl_cols[(size_t)cur[(size_t)lvl[(size_t)j]]++] = (int32_t)j;
    }

    BT_TICK("11");
    // ---- update triples of every column (src1, src2, dst | atomic << 15).  First those whose
    // destination is the DIAGONAL block of a column of the next level (that column's critical wave
    // applies them itself before factoring: listed per target column in dp), then the rest.
    // `atomic` marks a destination that another column of the same level also updates.
    pl->upd_ptr.assign((size_t)n + 1, 0);
    pl->upd_next.assign((size_t)n + 1, 0);
    pl->upd.clear();
    std::vector<std::vector<int32_t>> dplist((size_t)n);
    for (int64_t j = 0; j < n; ++j) {
        pl->upd_ptr[(size_t)j] = (int32_t)(pl->upd.size() / 3);
        const int32_t b = pl->col_ptr[(size_t)j] + 1, e = pl->col_ptr[(size_t)j + 1];
        for (int sub = 0; sub < 2; ++sub)
            for (int32_t s = b; s < e; ++s)
                for (int32_t t2 = b; t2 <= s; ++t2) {
                    const int32_t dcol = pl->row_idx[(size_t)t2];
                    const bool diag_next = s == t2 && lvl[(size_t)dcol] == lvl[(size_t)j] + 1;
                    if (diag_next != (sub == 0)) continue;
                    if (diag_next) { dplist[(size_t)dcol].push_back((int32_t)(pl->upd.size() / 3)); pl->upd_next[(size_t)j]++; }
                    pl->upd.push_back(s); pl->upd.push_back(t2);
                    pl->upd.push_back(find_pos(dcol, pl->row_idx[(size_t)s]));
                }
    }
    pl->upd_ptr[(size_t)n] = (int32_t)(pl->upd.size() / 3);
    I.updates = (int64_t)(pl->upd.size() / 3);
    {
        // shared destinations: scan each level's non-diagonal-next triples
        std::vector<int32_t> seen(pl->row_idx.size(), -1), seen_col(pl->row_idx.size(), -1);
        std::vector<uint8_t> shared(pl->row_idx.size(), 0);
        for (int32_t l = 0; l < nlev; ++l) {
            for (int32_t q = pl->lvl_ptr[(size_t)l]; q < pl->lvl_ptr[(size_t)l + 1]; ++q) {
                const int32_t j = pl->lvl_cols[(size_t)q];
                for (int32_t t = pl->upd_ptr[(size_t)j]; t < pl->upd_ptr[(size_t)j + 1]; ++t) {
                    const int32_t dst = pl->upd[(size_t)t * 3 + 2];
                    if (seen[(size_t)dst] == l && seen_col[(size_t)dst] != j) shared[(size_t)dst] = 1;
                    seen[(size_t)dst] = l; seen_col[(size_t)dst] = j;
                }
            }
            for (int32_t q = pl->lvl_ptr[(size_t)l]; q < pl->lvl_ptr[(size_t)l + 1]; ++q) {
                const int32_t j = pl->lvl_cols[(size_t)q];
                for (int32_t t = pl->upd_ptr[(size_t)j]; t < pl->upd_ptr[(size_t)j + 1]; ++t) {
                    int32_t &dst = pl->upd[(size_t)t * 3 + 2];
                    if (shared[(size_t)(dst & 0x7fff)]) dst |= 0x8000;
                }
            }
            for (int32_t q = pl->lvl_ptr[(size_t)l]; q < pl->lvl_ptr[(size_t)l + 1]; ++q) {
                const int32_t j = pl->lvl_cols[(size_t)q];
                for (int32_t t = pl->upd_ptr[(size_t)j]; t < pl->upd_ptr[(size_t)j + 1]; ++t) shared[(size_t)(pl->upd[(size_t)t * 3 + 2] & 0x7fff)] = 0;
            }
        }
    }
    // y rows updated by more than one column of a level: bit 16 of blk_col marks those blocks
    {
        std::vector<int32_t> seen((size_t)n, -1), cnt((size_t)n, 0);
        for (int32_t l = 0; l < nlev; ++l) {
            for (int pass = 0; pass < 2; ++pass)
                for (int32_t q = pl->lvl_ptr[(size_t)l]; q < pl->lvl_ptr[(size_t)l + 1]; ++q) {
                    const int32_t j = pl->lvl_cols[(size_t)q];
                    for (int32_t b = pl->col_ptr[(size_t)j] + 1; b < pl->col_ptr[(size_t)j + 1]; ++b) {
                        const int32_t r = pl->row_idx[(size_t)b];
                        if (pass == 0) { if (seen[(size_t)r] != l) { seen[(size_t)r] = l; cnt[(size_t)r] = 0; } cnt[(size_t)r]++; }
                        else if (cnt[(size_t)r] > 1) pl->blk_col[(size_t)b] |= 1 << 16;
                    }
                }
        }
    }
    pl->dp_ptr.assign((size_t)n + 1, 0);
    pl->dp.clear();
    for (int64_t j = 0; j < n; ++j) {
        pl->dp_ptr[(size_t)j] = (int32_t)pl->dp.size();
        pl->dp.insert(pl->dp.end(), dplist[(size_t)j].begin(), dplist[(size_t)j].end());
    }
    pl->dp_ptr[(size_t)n] = (int32_t)pl->dp.size();

    pl->lvl_meta.assign((size_t)nlev * kMaxLevelCols * 8, 0);
    for (int32_t l = 0; l < nlev; ++l)
        for (int q = 0; q < kMaxLevelCols; ++q) {
            int32_t *mrow = pl->lvl_meta.data() + ((size_t)l * kMaxLevelCols + q) * 8;
            if (pl->lvl_ptr[(size_t)l] + q >= pl->lvl_ptr[(size_t)l + 1]) { mrow[0] = -1; continue; }
            const int32_t j = pl->lvl_cols[(size_t)pl->lvl_ptr[(size_t)l] + q];
            mrow[0] = j; mrow[1] = pl->col_ptr[(size_t)j]; mrow[2] = pl->col_ptr[(size_t)j + 1] - pl->col_ptr[(size_t)j] - 1;
            mrow[3] = pl->upd_ptr[(size_t)j] + pl->upd_next[(size_t)j];
            mrow[4] = pl->upd_ptr[(size_t)j + 1] - mrow[3];
            mrow[5] = pl->dp_ptr[(size_t)j]; mrow[6] = pl->dp_ptr[(size_t)j + 1] - pl->dp_ptr[(size_t)j];
            mrow[7] = pl->lvl_ptr[(size_t)l + 1] - pl->lvl_ptr[(size_t)l];     // columns in this level
        }

    BT_TICK("12");
    // ---- fused schedule (k_solve_fused): ONE phase and one barrier per level.  A column's wave
    // applies the updates coming from the level right below to its own diagonal block and panel
    // rows itself ("pending": listed per destination block), factors and substitutes; every other
    // update ("lazy": destination two or more levels up) runs on the helper waves one level later,
    // concurrently with the next level's columns.
    //   fz_pend_ptr[nnzb+1], fz_pend[2 k]      (src1, src2) of the pending triples of a block
    //   fz_lazy_ptr[n+1],  fz_lazy[3 k]        lazy triples of a column (src1, src2, dst | shared << 15)
    //   fz_yurg[nnzb]                          1: the y contribution of this block is pending, not lazy
    //   fz_meta[nlev][kMaxLevelCols][8]        col, diag pos, #sub-blocks, first lazy triple, #lazy,
    //                                          first wave of the column, #waves (64 panel rows each), #cols
    {
        const size_t nb = pl->row_idx.size();
        std::vector<std::vector<int32_t>> pend(nb);
        pl->fz_lazy_ptr.assign((size_t)n + 1, 0);
        pl->fz_lazy.clear();
        pl->fz_yurg.assign(nb, 0);
        for (int64_t j = 0; j < n; ++j) {
            pl->fz_lazy_ptr[(size_t)j] = (int32_t)(pl->fz_lazy.size() / 3);
            for (int32_t t = pl->upd_ptr[(size_t)j]; t < pl->upd_ptr[(size_t)j + 1]; ++t) {
                const int32_t s1 = pl->upd[(size_t)t * 3], s2 = pl->upd[(size_t)t * 3 + 1], d = pl->upd[(size_t)t * 3 + 2];
                const int32_t dcol = pl->row_idx[(size_t)s2];
                if (lvl[(size_t)dcol] == lvl[(size_t)j] + 1) { pend[(size_t)(d & 0x7fff)].push_back(s1); pend[(size_t)(d & 0x7fff)].push_back(s2); }
                else { pl->fz_lazy.push_back(s1); pl->fz_lazy.push_back(s2); pl->fz_lazy.push_back(d); }
            }
            for (int32_t b = pl->col_ptr[(size_t)j] + 1; b < pl->col_ptr[(size_t)j + 1]; ++b)
                if (lvl[(size_t)pl->row_idx[(size_t)b]] == lvl[(size_t)j] + 1) pl->fz_yurg[(size_t)b] = 1;
        }
        pl->fz_lazy_ptr[(size_t)n] = (int32_t)(pl->fz_lazy.size() / 3);
        pl->fz_pend_ptr.assign(nb + 1, 0);
        pl->fz_pend.clear();
        for (size_t b = 0; b < nb; ++b) {
            pl->fz_pend_ptr[b] = (int32_t)(pl->fz_pend.size() / 2);
            pl->fz_pend.insert(pl->fz_pend.end(), pend[b].begin(), pend[b].end());
        }
        pl->fz_pend_ptr[nb] = (int32_t)(pl->fz_pend.size() / 2);
        // packed per-level record of the (at most two) columns k_solve_fused handles per level, 4 ints each:
        //   [4q]   col | #sub-blocks << 8 | source column of the diagonal block's first pending pair << 16
        //   [4q+1] diag pos | #row waves << 16 | (q = 0: #cols << 24)
        //   [4q+2] first lazy triple | #lazy << 16
        //   [4q+3] source block of the diagonal block's first pending pair | min(#pending, 3) << 15
        pl->fz_pmeta.assign((size_t)nlev * 8, 0);
        for (int32_t l = 0; l < nlev; ++l) {
            const int32_t nc = pl->lvl_ptr[(size_t)l + 1] - pl->lvl_ptr[(size_t)l];
            int32_t *m = pl->fz_pmeta.data() + (size_t)l * 8;
            for (int32_t q = 0; q < nc && q < 2; ++q) {
                const int32_t j = pl->lvl_cols[(size_t)pl->lvl_ptr[(size_t)l] + q];
                const int32_t dpos = pl->col_ptr[(size_t)j], cnt = pl->col_ptr[(size_t)j + 1] - dpos - 1;
                const int32_t k0 = pl->fz_pend_ptr[(size_t)dpos], np = pl->fz_pend_ptr[(size_t)dpos + 1] - k0;
                const int32_t sd = np > 0 ? pl->fz_pend[(size_t)k0 * 2] : 0;
                const int32_t ysrc = np > 0 ? (pl->blk_col[(size_t)sd] & 255) : 0;
                const int32_t lz0 = pl->fz_lazy_ptr[(size_t)j], nlz = pl->fz_lazy_ptr[(size_t)j + 1] - lz0;
                m[4 * q] = j | (cnt << 8) | (ysrc << 16);
                m[4 * q + 1] = dpos | (((6 * cnt + 1 + 63) / 64) << 16);
                m[4 * q + 2] = lz0 | (nlz << 16);
                // which columns (slots) of the level below hold pending sources of this column's blocks: only those
                // have to be waited for by the barrier-free solver (bits 20, 21)
                int32_t dep = 0;
                for (int32_t b = dpos; b < pl->col_ptr[(size_t)j + 1]; ++b)
                    for (int32_t k = pl->fz_pend_ptr[(size_t)b]; k < pl->fz_pend_ptr[(size_t)b + 1]; ++k) {
                        const int32_t sc = pl->blk_col[(size_t)pl->fz_pend[(size_t)k * 2]] & 255;
                        for (int32_t q2 = 0; l > 0 && q2 < pl->lvl_ptr[(size_t)l] - pl->lvl_ptr[(size_t)l - 1]; ++q2)
                            if (pl->lvl_cols[(size_t)pl->lvl_ptr[(size_t)l - 1] + q2] == sc) dep |= 1 << q2;
                    }
                m[4 * q + 3] = sd | ((np < 3 ? np : 3) << 15) | (dep << 20);
            }
            m[1] |= nc << 24;
        }
        // what the kernel keeps per block, ready-made: row | col << 8 | shared-y << 24 | pending-y << 25, and
        // the first pending pair src1 | src2 << 15 | min(#pending, 3) << 30
        pl->fz_rowinfo.assign(nb, 0);
        pl->fz_pfirst.assign(nb, 0);
        pl->fz_psecond.assign(nb, 0);                       // second pending pair src1 | src2 << 15 (a block gets at most
        bool pend_ok = true;                                //  one from each of the two columns of the level below)
        for (size_t b = 0; b < nb; ++b) {
            pl->fz_rowinfo[b] = pl->row_idx[b] | ((pl->blk_col[b] & 255) << 8) | ((pl->blk_col[b] >> 16) << 24) | (pl->fz_yurg[b] << 25);
            const int32_t k0 = pl->fz_pend_ptr[b], c = pl->fz_pend_ptr[b + 1] - k0;
            if (c > 2) pend_ok = false;
            if (c > 0 && nb < 32768)
                pl->fz_pfirst[b] = (int32_t)((uint32_t)pl->fz_pend[(size_t)k0 * 2] | ((uint32_t)pl->fz_pend[(size_t)k0 * 2 + 1] << 15) |
                                             ((uint32_t)(c < 3 ? c : 3) << 30));
            if (c > 1 && nb < 32768)
                pl->fz_psecond[b] = (int32_t)((uint32_t)pl->fz_pend[(size_t)k0 * 2 + 2] | ((uint32_t)pl->fz_pend[(size_t)k0 * 2 + 3] << 15));
        }
        // back substitution, levels descending, one wave per column slot: a barrier is needed before a level
        // only if one of its columns reads an x_i written by another slot's wave since the last barrier
        pl->bs_sync.assign((size_t)nlev, 0);
        {
            std::vector<int32_t> slot((size_t)n, 0);
            for (int32_t l = 0; l < nlev; ++l)
                for (int32_t qi = pl->lvl_ptr[(size_t)l]; qi < pl->lvl_ptr[(size_t)l + 1]; ++qi)
                    slot[(size_t)pl->lvl_cols[(size_t)qi]] = qi - pl->lvl_ptr[(size_t)l];
            int32_t last_barrier = nlev;          // no barrier yet
            for (int32_t l = nlev - 1; l >= 0; --l) {
                bool need = false;
                for (int32_t qi = pl->lvl_ptr[(size_t)l]; qi < pl->lvl_ptr[(size_t)l + 1]; ++qi) {
                    const int32_t j = pl->lvl_cols[(size_t)qi];
                    for (int32_t b = pl->col_ptr[(size_t)j] + 1; b < pl->col_ptr[(size_t)j + 1]; ++b) {
                        const int32_t i = pl->row_idx[(size_t)b];
                        if (slot[(size_t)i] != slot[(size_t)j] && last_barrier > lvl[(size_t)i] - 1) need = true;
                    }
                }
                if (need) { pl->bs_sync[(size_t)l] = 1; last_barrier = l; }
            }
        }
        pl->fz_meta.assign((size_t)nlev * kMaxLevelCols * 8, 0);
        pl->fz_ok = (pl->fz_lazy.size() / 3 < 65536 && pl->row_idx.size() < 32768 && pend_ok) ? 1 : 0;
        pl->fzp_ok = pl->fz_ok;                          // additionally: every column's panel rows (+ y) fit one wave
        for (int64_t j = 0; j < n; ++j) if (6 * (pl->col_ptr[(size_t)j + 1] - pl->col_ptr[(size_t)j] - 1) + 1 > 64) pl->fzp_ok = 0;
        for (int32_t l = 0; l < nlev; ++l) {
            if (pl->lvl_ptr[(size_t)l + 1] - pl->lvl_ptr[(size_t)l] > 2) { pl->fz_ok = 0; pl->fzp_ok = 0; }
            int32_t w0 = 0;
            for (int q = 0; q < kMaxLevelCols; ++q) {
                int32_t *mrow = pl->fz_meta.data() + ((size_t)l * kMaxLevelCols + q) * 8;
                if (pl->lvl_ptr[(size_t)l] + q >= pl->lvl_ptr[(size_t)l + 1]) { mrow[0] = -1; mrow[5] = w0; continue; }
                const int32_t j = pl->lvl_cols[(size_t)pl->lvl_ptr[(size_t)l] + q];
                mrow[0] = j; mrow[1] = pl->col_ptr[(size_t)j]; mrow[2] = pl->col_ptr[(size_t)j + 1] - pl->col_ptr[(size_t)j] - 1;
                mrow[3] = pl->fz_lazy_ptr[(size_t)j]; mrow[4] = pl->fz_lazy_ptr[(size_t)j + 1] - mrow[3];
                mrow[5] = w0; mrow[6] = (6 * mrow[2] + 1 + 63) / 64;
                mrow[7] = pl->lvl_ptr[(size_t)l + 1] - pl->lvl_ptr[(size_t)l];
                w0 += mrow[6];
            }
        }
    }

    return BT_OK;
    };
    if (!pl->wide) {
        const int src_rc = symbolic();
        if (src_rc != BT_OK) return src_rc;
        // The factor does not fit LDS as double: float32 with refinement, or the dense solver — whichever is priced lower.  Measured
        // (tools/gpu_solver_choice.py, profiles/r06_solver_choice.txt): a block-sparse solve costs ~2.5 us a level where the float32
        // factor fits LDS, ~8 us from global memory, + 26 ns a block update, and a step takes two of them (three where the first
        // refinement step has not converged) — a 255-pose band of half-width 7: 131 levels, 7k updates, 2.4 ms a step; 255 poses
        // tied by long-range edges: 2.8M updates, 140 ms; the dense solver ~12 us a pose whatever the pattern (255 poses 3.1 ms).
        // Long bands stay block-sparse, filled-in systems go dense (95 poses: 12.4 -> 1.0 ms a step, 255: 211 -> 3.5 ms) and get
        // a float64 factor with it.
        // (a forced solver — tests, measurement — keeps the block-sparse tables whatever their size)
        const size_t nlev = pl->lvl_ptr.size() - 1;
        const auto lds_at = [&](size_t elem) { return solve_lds_bytes_raw((size_t)I.nnz_blocks, (size_t)(6 * n), (size_t)I.updates, (size_t)n, nlev, pl->dp.size(), elem); };
        const double level_us = lds_at(sizeof(float)) <= kLdsBudget ? 2.5 : 8.0;
        if (force().solver < 0 && lds_at(sizeof(double)) > kLdsBudget && 2.2 * (level_us * (double)nlev + 0.026 * (double)I.updates) > 12.0 * (double)n)
            go_wide();
    }

    BT_TICK("13");
    // ---- k_tile's first loads, indexed by the tile alone (no dependent index chain in its prologue):
    //   tile_ij[t][max_tile_pairs]   cameras (i | j << 16) of the tile's pairs, in local pair order
    //   tile_kx[t][64]               patch of every track of the tile (-1: no track in this lane)
    {
        const size_t mtp = (size_t)std::max(pl->max_tile_pairs, 1);
        pl->tile_ij.assign((size_t)I.tiles * mtp, 0);
        pl->tile_kx.assign((size_t)I.tiles * kLanes, -1);
        for (int64_t t = 0; t < I.tiles; ++t) {
            for (int32_t q = 0; q < pl->tile_npair[(size_t)t]; ++q) {
                const int32_t gp = pl->tile_pairs[(size_t)(pl->tile_pair0[(size_t)t] + q)];
                pl->tile_ij[(size_t)t * mtp + (size_t)q] = pl->pair_i[(size_t)gp] | (pl->pair_j[(size_t)gp] << 16);
            }
            for (int32_t ln = 0; ln < pl->tile_ntrk[(size_t)t]; ++ln)
                pl->tile_kx[(size_t)t * kLanes + (size_t)ln] = pl->kx[(size_t)(pl->tile_trk0[(size_t)t] + ln)];
        }
    }

    // ---- compact tables of the wave-per-tile kernels (k_stream): per (slot, lane) a 16-bit code
    // (local target camera | local pair << 8; the global pair id is tile_pairs[pair0 + local pair]); per (tile,
    // lane) the local source camera (one per track: ii = ix[kk], checked above); per tile one 32-byte record
    //   [0] ntrk | ncam << 8 | npair << 16 | flags << 24   [1] slot0  [2] nslot  [3] cam0  [4] pair0  [5] trk0
    const bool want_stream_tables = want_slots && I.tiles >= std::min(em_min_p, st_min_p);   // (they are made from the slot arrays)
    const bool dev_wpt = dev_slots && I.tiles >= std::min(em_min_p, st_min_p);               // (... where those are: on the device)
    pl->st_ok = want_stream_tables || dev_wpt ? 1 : 0;
    pl->st_min = st_min_p; pl->em_min = em_min_p;
    pl->dev_wpt = dev_wpt ? 1 : 0;
    if (dev_wpt) {
        // the records but for the straddle flag (k_plan_slots adds it); slot_code, tile_la, it_edge and tile_sinfo are written
        // by the device's passes into the plan's buffer (plan_device.hip)
        pl->slot_code.clear(); pl->tile_la.clear(); pl->it_edge.clear(); pl->tile_sinfo.clear();
        pl->tile_rec.assign((size_t)I.tiles * 8, 0);
        for (int64_t t = 0; t < I.tiles; ++t) {
            int32_t *r = pl->tile_rec.data() + (size_t)t * 8;
            r[0] = pl->tile_ntrk[(size_t)t] | (pl->tile_ncam[(size_t)t] << 8) | (pl->tile_npair[(size_t)t] << 16) | (pl->tile_flags[(size_t)t] << 24);
            r[1] = pl->tile_slot0[(size_t)t]; r[2] = pl->tile_nslot[(size_t)t]; r[3] = pl->tile_cam0[(size_t)t];
            r[4] = pl->tile_pair0[(size_t)t]; r[5] = pl->tile_trk0[(size_t)t];
        }
    }
    if (want_stream_tables) {
        pl->slot_code.assign((size_t)slots * kLanes, 0xffff);
        pl->tile_la.assign((size_t)I.tiles * kLanes, 0xff);
        pl->tile_rec.assign((size_t)I.tiles * 8, 0);
        for (int64_t t = 0; t < I.tiles; ++t) {
            const size_t b0 = (size_t)pl->tile_slot0[(size_t)t] * kLanes;
            const int32_t ns = pl->tile_nslot[(size_t)t];
            for (int32_t sl = 0; sl < ns; ++sl)
                for (int ln = 0; ln < kLanes; ++ln) {
                    const size_t i = b0 + (size_t)sl * kLanes + (size_t)ln;
                    if (pl->slot_edge[i] < 0) continue;
                    pl->slot_code[i] = (uint16_t)((pl->slot_lab[i] >> 8) | ((uint16_t)pl->slot_lp[i] << 8));
                    pl->tile_la[(size_t)t * kLanes + (size_t)ln] = (uint8_t)(pl->slot_lab[i] & 0xff);
                }
            // flag bit 2: some track repeats a target camera across the boundary of the two half-chunks of slots the
            // streaming kernel gives its two waves (both then add to the same element of the local E)
            int32_t straddle = 0;
            const int32_t ch = (ns + 1) >> 1;
            if (ch < ns)
                for (int ln = 0; ln < kLanes && !straddle; ++ln) {
                    const size_t i0 = b0 + (size_t)(ch - 1) * kLanes + (size_t)ln, i1 = i0 + kLanes;
                    if (pl->slot_edge[i0] >= 0 && pl->slot_edge[i1] >= 0 && (pl->slot_lab[i0] >> 8) != 0xff &&
                        (pl->slot_lab[i0] >> 8) == (pl->slot_lab[i1] >> 8)) straddle = 4;
                }
            int32_t *r = pl->tile_rec.data() + (size_t)t * 8;
            r[0] = pl->tile_ntrk[(size_t)t] | (pl->tile_ncam[(size_t)t] << 8) | (pl->tile_npair[(size_t)t] << 16) | ((pl->tile_flags[(size_t)t] | straddle) << 24);
            r[1] = pl->tile_slot0[(size_t)t]; r[2] = ns; r[3] = pl->tile_cam0[(size_t)t];
            r[4] = pl->tile_pair0[(size_t)t]; r[5] = pl->tile_trk0[(size_t)t];
        }
    }

    // ---- edge-major layout of the same tiles (k_edge2 / k_edge2u, ba_edge2.hip / ba_edge2u.hip): the slots of a track padded to S = the next
    // power of two, an ITERATION = 64 lanes = 64 / S consecutive tracks x S slots (lane = track_in_iteration * S + slot),
    // so that on the caller's track-major edge lists a wave reads 64 consecutive edges, a track's sums are reductions over
    // S adjacent lanes, and a lane meets the same camera pair in every iteration of a tile (per-lane pair sums, no
    // wave-wide reduction per slot).  Usable when every tile is SLOT-UNIFORM: all tracks of the tile have the same pair
    // (or no edge) in slot s.
    //   it_edge[(it0 + i) * 64 + lane]   edge of (tile, iteration i, lane), -1 = none
    //   tile_sinfo[t * 64 + s]           local target camera | local pair << 8 | repeat << 16 | used << 17   (s < S)
    //                                    repeat: this slot or a neighbour holds the same pair again (repeated observation)
    //   tile_rec[6] = it0, tile_rec[7] = log2 S | iterations << 8
    pl->em_ok = 0; pl->em_its = 0; pl->em_lgs = -1;
    if (dev_wpt && I.tiles >= em_min_p) {
        // (the iteration counts follow from the tiles' slot and track counts; whether every tile is slot-uniform is k_plan_sinfo's
        //  verdict, read back by upload_plan: em_ok here is "as far as the host can tell")
        pl->em_ok = 1;
        int64_t its = 0;
        for (int64_t t = 0; t < I.tiles; ++t) {
            const int32_t ns = pl->tile_nslot[(size_t)t], nt = pl->tile_ntrk[(size_t)t];
            int lg = 0;
            while ((1 << lg) < ns) ++lg;
            if (lg > 6) { pl->em_ok = 0; break; }
            const int32_t G = kLanes >> lg;
            const int32_t nit_t = em_iterations(nt, G);
            pl->tile_rec[(size_t)t * 8 + 6] = (int32_t)its;
            pl->tile_rec[(size_t)t * 8 + 7] = lg | (nit_t << 8);
            its += nit_t;
            if (t == 0) pl->em_lgs = lg; else if (pl->em_lgs != lg) pl->em_lgs = -1;
        }
        if (!pl->em_ok) { for (int64_t t = 0; t < I.tiles; ++t) pl->tile_rec[(size_t)t * 8 + 6] = pl->tile_rec[(size_t)t * 8 + 7] = 0; pl->em_lgs = -1; }
        pl->em_its = pl->em_ok ? its : 0;
    }
    if (want_stream_tables && I.tiles >= em_min_p) {
        pl->em_ok = 1;
        int64_t its = 0;
        std::vector<int32_t> it0((size_t)I.tiles), lgS((size_t)I.tiles), nit((size_t)I.tiles);
        for (int64_t t = 0; t < I.tiles; ++t) {
            const int32_t ns = pl->tile_nslot[(size_t)t], nt = pl->tile_ntrk[(size_t)t];
            int lg = 0;
            while ((1 << lg) < ns) ++lg;
            if (lg > 6) { pl->em_ok = 0; break; }
            const int32_t G = kLanes >> lg;
            it0[(size_t)t] = (int32_t)its; lgS[(size_t)t] = lg; nit[(size_t)t] = em_iterations(nt, G);
            its += nit[(size_t)t];
        }
        pl->tile_sinfo.assign(pl->em_ok ? (size_t)I.tiles * kLanes : 0, 0);
        for (int64_t t = 0; t < I.tiles && pl->em_ok; ++t) {
            const size_t b0 = (size_t)pl->tile_slot0[(size_t)t] * kLanes;
            const int32_t ns = pl->tile_nslot[(size_t)t], nt = pl->tile_ntrk[(size_t)t];
            for (int32_t sl = 0; sl < ns && pl->em_ok; ++sl) {
                int32_t code = -1;
                for (int ln = 0; ln < nt; ++ln) {
                    const size_t i = b0 + (size_t)sl * kLanes + (size_t)ln;
                    if (pl->slot_edge[i] < 0) continue;
                    const int32_t c = (int32_t)pl->slot_code[i];
                    if (code < 0) code = c; else if (code != c) { pl->em_ok = 0; break; }
                }
                if (code >= 0) pl->tile_sinfo[(size_t)t * kLanes + (size_t)sl] = (uint32_t)code | (1u << 17);
            }
            for (int32_t sl = 0; sl + 1 < ns && pl->em_ok; ++sl) {
                uint32_t &x = pl->tile_sinfo[(size_t)t * kLanes + (size_t)sl], &y = pl->tile_sinfo[(size_t)t * kLanes + (size_t)sl + 1];
                if ((x >> 17 & 1u) && (y >> 17 & 1u) && ((x >> 8) & 0xffu) == ((y >> 8) & 0xffu)) { x |= 1u << 16; y |= 1u << 16; }
            }
        }
        if (pl->em_ok) {
            pl->it_edge.assign((size_t)its * kLanes, -1);
            for (int64_t t = 0; t < I.tiles; ++t) {
                const size_t b0 = (size_t)pl->tile_slot0[(size_t)t] * kLanes;
                const int32_t ns = pl->tile_nslot[(size_t)t], nt = pl->tile_ntrk[(size_t)t], lg = lgS[(size_t)t], G = kLanes >> lg;
                for (int32_t k = 0; k < nt; ++k)
                    for (int32_t sl = 0; sl < ns; ++sl)
                        pl->it_edge[((size_t)it0[(size_t)t] + (size_t)(k / G)) * kLanes + (size_t)((k % G) << lg) + (size_t)sl] =
                            pl->slot_edge[b0 + (size_t)sl * kLanes + (size_t)k];
                pl->tile_rec[(size_t)t * 8 + 6] = it0[(size_t)t];
                pl->tile_rec[(size_t)t * 8 + 7] = lg | (nit[(size_t)t] << 8);
            }
        } else {
            pl->it_edge.clear(); pl->tile_sinfo.clear();
        }
        pl->em_its = pl->em_ok ? its : 0;
        pl->em_lgs = -1;
        if (pl->em_ok && I.tiles > 0) {
            pl->em_lgs = lgS[0];
            for (int64_t t = 1; t < I.tiles; ++t) if (lgS[(size_t)t] != pl->em_lgs) { pl->em_lgs = -1; break; }
        }
    }

    // ---- pair-major layout of the same tiles (k_etile, ba_etile.hip): lane = track_in_iteration * S + s, where s is the LOCAL
    // PAIR of the tile (S = its pair count padded to a power of two, an iteration = 64 / S consecutive tracks) and the edges a
    // track has with that pair — repeated observations (batrack.py:399-410 appends a window's factors again every keyframe
    // step) — are the lane's D ROUNDS.  A lane then keeps one camera pair for the whole tile (the 27 per-pair products are
    // summed per lane in registers), owns its (track, target camera) element of the local E exclusively (plain store of the
    // sum over its rounds), and a track's own sums (C, w, source-camera E) are reductions over S adjacent lanes.  Any tile of
    // at most 64 pairs can be laid out this way (a track that lacks a pair has no edge in that lane).
    //   pm_edge[(round0 + it * D + d) * 64 + lane]   edge of (tile, iteration it, round d, lane), -1 = none
    //   pm_rec[t * 4]   = round0, log2 S | D << 8, iterations, 0
    //   pm_lb[t * 64 + s]   local target camera of local pair s (0xff: fixed, or s >= npair)
    //   pm_la[t * 64 + l]   local source camera of track l of the tile (0xff: fixed / no track)
    // pm_ok: 2 = the plan was tiled FOR the pair-major kernel (small tiles) or BT_ETILE=2 forces it, 1 = the tables exist but
    // k_tile keeps the plan (at one tile per CU, e.g. the 64-keyframe benchmark, k_tile measured 12.7 us against 15.0)
    pl->pm_ok = I.tiles > 0 ? 1 : 0; pl->pm_rounds = 0;
    if (dstats) {
        if (I.tiles <= 0) return BT_NEED_EDGES;
        if (pm_direct) {
            if (pm_fail || etile_full_lds_bytes(pl->max_rows16, pl->max_tile_pairs, sizeof(double)) > kEtileLdsBudget) {
                if (tcap_retry) return BT_NEED_EDGES;
                RetryScope guard(tcap_retry);                      // small tiles are for k_etile only: lay the plan out again for k_tile
                return build_plan_host(ii64, jj64, kk64, E, n_buf, p_tot, fixedp, n_all_min, own_lo, own_hi, pl, packed, keep_slots, dstats);
            }
            pl->pm_ok = 2; pl->dev_pm = 1;
        } else {
            pl->pm_ok = 0; pl->dev_slots = 1;
            pl->pm_edge.clear(); pl->pm_rec.clear(); pl->pm_lb.clear(); pl->pm_la.clear();
            pl->dev_off = off;
        }
        pl->dev_pair_of = pair_of; pl->dev_f_lo = f_lo; pl->dev_nw = nw;
    } else
    if (pm_direct) {
        // (built tile by tile above) — the plan stays with k_etile only if that kernel's LDS need fits in float64
        if (pm_fail || I.tiles <= 0 || etile_full_lds_bytes(pl->max_rows16, pl->max_tile_pairs, sizeof(double)) > kEtileLdsBudget) {
            pl->pm_ok = 0;
            pl->pm_edge.clear(); pl->pm_rec.clear(); pl->pm_lb.clear(); pl->pm_la.clear();
        } else {
            pl->pm_ok = 2;
            pl->pm_rounds = pm_rounds_acc;
        }
    } else {
        pl->pm_edge.clear(); pl->pm_rec.clear(); pl->pm_lb.clear(); pl->pm_la.clear();
    }
    if (!pm_direct) {
        const int pm_env = etile_mode();                            // 0: k_tile forced
        if (!pm_env || (pm_env != 2 && tcap == kLanes)) pl->pm_ok = 0;      // (tables only for the plans that will use them)
        if (!loose.empty()) pl->pm_ok = 0;                                   // (k_etile's stored per-tile sums bypass the accumulators the loose tracks add to)
        for (int64_t t = 0; t < I.tiles && pl->pm_ok; ++t) if (pl->tile_npair[(size_t)t] > kLanes) pl->pm_ok = 0;
        if (pl->pm_ok) {
            pl->pm_rec.assign((size_t)I.tiles * 4, 0);
            pl->pm_lb.assign((size_t)I.tiles * kLanes, 0xff);
            pl->pm_la.assign((size_t)I.tiles * kLanes, 0xff);
            std::vector<int32_t> cnt;                              // edges of (track, local pair) placed so far
            int64_t rounds = 0;
            for (int64_t t = 0; t < I.tiles && pl->pm_ok; ++t) {
                const size_t b0 = (size_t)pl->tile_slot0[(size_t)t] * kLanes;
                const int32_t ns = pl->tile_nslot[(size_t)t], nt = pl->tile_ntrk[(size_t)t], np = pl->tile_npair[(size_t)t];
                int lg = 0;
                while ((1 << lg) < np) ++lg;
                const int32_t S = 1 << lg, G = kLanes >> lg, nit = (nt + G - 1) / G;
                // multiplicities
                cnt.assign((size_t)nt * (size_t)S, 0);
                int32_t D = 1;
                for (int32_t k = 0; k < nt; ++k)
                    for (int32_t sl = 0; sl < ns; ++sl) {
                        const size_t i = b0 + (size_t)sl * kLanes + (size_t)k;
                        if (pl->slot_edge[i] < 0) continue;
                        D = std::max(D, ++cnt[(size_t)k * S + pl->slot_lp[i]]);
                    }
                if (D > 255) { pl->pm_ok = 0; break; }
                pl->pm_rec[(size_t)t * 4] = (int32_t)rounds;
                pl->pm_rec[(size_t)t * 4 + 1] = lg | (D << 8);
                pl->pm_rec[(size_t)t * 4 + 2] = nit;
                pl->pm_edge.resize((size_t)(rounds + (int64_t)nit * D) * kLanes, -1);
                std::fill(cnt.begin(), cnt.end(), 0);
                for (int32_t k = 0; k < nt; ++k)
                    for (int32_t sl = 0; sl < ns; ++sl) {
                        const size_t i = b0 + (size_t)sl * kLanes + (size_t)k;
                        if (pl->slot_edge[i] < 0) continue;
                        const int32_t lp = pl->slot_lp[i], d = cnt[(size_t)k * S + lp]++;
                        pl->pm_edge[((size_t)rounds + (size_t)(k / G) * D + (size_t)d) * kLanes + (size_t)((k % G) << lg) + (size_t)lp] = pl->slot_edge[i];
                        pl->pm_lb[(size_t)t * kLanes + (size_t)lp] = (uint8_t)(pl->slot_lab[i] >> 8);
                        pl->pm_la[(size_t)t * kLanes + (size_t)k] = (uint8_t)(pl->slot_lab[i] & 0xff);
                    }
                rounds += (int64_t)nit * D;
            }
            pl->pm_rounds = rounds;
            if (pl->pm_ok && (tcap < kLanes || pm_env == 2)) pl->pm_ok = 2;
            if (!pl->pm_ok) { pl->pm_edge.clear(); pl->pm_rec.clear(); pl->pm_lb.clear(); pl->pm_la.clear(); }
        }
    }
    {
        if (!pl->pm_ok && tcap < kLanes && !tcap_retry) {          // small tiles are for k_etile only: lay the plan out again for k_tile
            RetryScope guard(tcap_retry);
            return build_plan_host(ii64, jj64, kk64, E, n_buf, p_tot, fixedp, n_all_min, own_lo, own_hi, pl, packed, keep_slots);
        }
    }
    // ---- partial sums instead of atomics for the pair-major kernel: hundreds of workgroups' float64 atomics on the same few
    // thousand elements of [S | y] and of the per-pair sums queue up in the L2 (measured on the 160-tile window: half of the
    // kernel).  Every tile stores its Schur product, E Q w' and pair sums (StepArgs::spart); k_pair_finalize adds them up —
    // the products by position over each GROUP of consecutive tiles with the same cameras (a sliding window: the tiles of one
    // source frame; at most kSpGroupMax tiles, so that the sum is one round of independent loads), the pair sums over the
    // plan's list of every pair's (tile, local pair) entries.
    pl->sp_ok = 0; pl->et_lgts = 0;
    {
        int mt = 1;
        for (int64_t t = 0; t < I.tiles; ++t) mt = std::max(mt, (int)pl->tile_ntrk[(size_t)t]);
        while ((1 << pl->et_lgts) < mt) ++pl->et_lgts;
    }
    if (pl->pm_ok == 2 && ((int64_t)sp_tile_doubles(pl->max_rows16, pl->max_tile_pairs) + ((int64_t)pl->max_rows16 << pl->et_lgts)) * I.tiles * 8 <= ((int64_t)96 << 20)) pl->sp_ok = 1;
    pl->sg_ptr.clear();
    if (pl->sp_ok) {
        for (int64_t t = 0; t < I.tiles; ++t)
            if (t == 0 || !(pl->tile_flags[(size_t)t] & 1) || t - pl->sg_ptr.back() >= kSpGroupMax) pl->sg_ptr.push_back((int32_t)t);
        pl->sg_ptr.push_back((int32_t)I.tiles);
    }
    pl->pp_ptr.clear(); pl->pp_idx.clear();
    if (pl->sp_ok) {                                               // which tiles hold sums of which camera pair
        pl->pp_ptr.assign((size_t)I.pairs + 1, 0);
        for (size_t q = 0; q < pl->tile_pairs.size(); ++q) pl->pp_ptr[(size_t)pl->tile_pairs[q] + 1]++;
        for (int64_t p = 0; p < I.pairs; ++p) pl->pp_ptr[(size_t)p + 1] += pl->pp_ptr[(size_t)p];
        pl->pp_idx.assign(pl->tile_pairs.size(), 0);
        std::vector<int32_t> cur(pl->pp_ptr.begin(), pl->pp_ptr.end() - 1);
        for (int64_t t = 0; t < I.tiles; ++t)
            for (int32_t q = 0; q < pl->tile_npair[(size_t)t]; ++q)
                pl->pp_idx[(size_t)cur[(size_t)pl->tile_pairs[(size_t)(pl->tile_pair0[(size_t)t] + q)]]++] = (int32_t)(t << 6) | q;
    }

    BT_TICK("14");
    pl->max_tile_slots = 0;
    for (int64_t t = 0; t < I.tiles; ++t) pl->max_tile_slots = std::max(pl->max_tile_slots, (int)pl->tile_nslot[(size_t)t]);

    // ---- k_update: which patches carry a track (bitmap + rank per 32 patches: the patch buffer is BUFFER_SIZE x M
    // = 262,144 slots in the reference's configuration, the window's tracks a few thousand)
    const size_t nwords = (size_t)((p_tot + 31) / 32);
    pl->act_bits.assign(nwords, 0u);
    pl->act_rank.assign(nwords, 0);
    for (int32_t k = 0; k < I.m; ++k) pl->act_bits[(size_t)(pl->kx[(size_t)k] >> 5)] |= 1u << (pl->kx[(size_t)k] & 31);
    {
        int32_t run = 0;
        for (size_t w = 0; w < nwords; ++w) { pl->act_rank[w] = run; run += __builtin_popcount(pl->act_bits[w]); }
    }

    BT_TICK("end");
    layout_workspace(pl);
    return BT_OK;
}

}  // namespace bt
