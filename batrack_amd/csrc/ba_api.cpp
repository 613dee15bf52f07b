// ba_api.cpp — the C ABI declared in include/batrack_ba.h.
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <utility>
#include <vector>

#include "ba_kernels.hpp"
#include "dev_cache.hpp"

#define BT_VERSION 204

namespace bt {

template <class T>
static size_t put(std::vector<char> &buf, const std::vector<T> &v) {
    size_t off = (buf.size() + 255) / 256 * 256;
    buf.resize(off + v.size() * sizeof(T) + 1);
    if (!v.empty()) std::memcpy(buf.data() + off, v.data(), v.size() * sizeof(T));
    return off;
}

// Device buffers of destroyed plans are kept for the next plan: the caller replaces its edge list every frame
// (batrack.py:189-212), so a plan lives for 2*ITER steps, and hipMalloc / hipFree per frame is what a frame
// would otherwise wait for (hipFree synchronises the device and unmaps).  Sizes are rounded up so that the
// slowly varying edge count of a sliding window maps onto the same bucket.
struct DevPool {
    struct Entry { void *p; size_t cap; hipEvent_t ev; bool pending; };   // ev: recorded behind the last kernels that read the buffer
    std::mutex mu;
    std::vector<Entry> free_list;
    static size_t bucket(size_t bytes) {           // 12.5 % head room, whole MiB, at least 2 MiB
        const size_t g = (size_t)1 << 20;
        return std::max<size_t>(2 * g, (bytes + bytes / 8 + g - 1) / g * g);
    }
    // *wait: an event the caller's first write into the buffer has to be ordered behind (nullptr: none); the caller hands
    // it back with give_event() once the wait is enqueued
    void *acquire(size_t bytes, size_t *cap, hipEvent_t *wait) {
        const size_t want = bucket(bytes);
        *wait = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            size_t best = free_list.size();
            for (size_t i = 0; i < free_list.size(); ++i)
                if (free_list[i].cap >= bytes && free_list[i].cap <= 2 * want &&
                    (best == free_list.size() || free_list[i].cap < free_list[best].cap)) best = i;
            if (best != free_list.size()) {
                const Entry e = free_list[best];
                free_list.erase(free_list.begin() + (long)best);
                *cap = e.cap;
                if (e.pending) *wait = e.ev; else if (e.ev) spare.push_back(e.ev);
                return e.p;
            }
        }
        void *d = nullptr;
        if (hipMalloc(&d, want) != hipSuccess) return nullptr;
        *cap = want;
        return d;
    }
    std::vector<hipEvent_t> spare;                 // events not attached to a buffer (guarded by mu)
    void give_event(hipEvent_t ev) { if (ev) { std::lock_guard<std::mutex> lk(mu); spare.push_back(ev); } }
    hipEvent_t take_event() {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!spare.empty()) { hipEvent_t ev = spare.back(); spare.pop_back(); return ev; }
        }
        hipEvent_t ev = nullptr;
        return hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess ? ev : nullptr;
    }
    // last_stream != nullptr / launched: kernels reading the buffer may still be queued there
    void release(void *p, size_t cap, bool launched, hipStream_t last_stream) {
        Entry e{p, cap, nullptr, false};
        if (launched) {
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!spare.empty()) { e.ev = spare.back(); spare.pop_back(); }
            }
            if (!e.ev && hipEventCreateWithFlags(&e.ev, hipEventDisableTiming) != hipSuccess) e.ev = nullptr;
            if (e.ev && hipEventRecord(e.ev, last_stream) == hipSuccess) e.pending = true;
            else (void)hipStreamSynchronize(last_stream);            // no event: the plain hipFree this pool replaces synchronised too
        }
        void *drop = nullptr;
        hipEvent_t drop_ev = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            free_list.push_back(e);
            if (free_list.size() > 8) {            // keep the eight largest
                size_t sm = 0;
                for (size_t i = 1; i < free_list.size(); ++i) if (free_list[i].cap < free_list[sm].cap) sm = i;
                drop = free_list[sm].p; drop_ev = free_list[sm].ev;
                free_list.erase(free_list.begin() + (long)sm);
            }
        }
        if (drop) (void)hipFree(drop);             // (hipFree waits for the device: nothing can still read the buffer)
        if (drop_ev) (void)hipEventDestroy(drop_ev);
    }
    void trim() {
        std::vector<Entry> all;
        std::vector<hipEvent_t> evs;
        { std::lock_guard<std::mutex> lk(mu); all.swap(free_list); evs.swap(spare); }
        for (auto &e : all) { (void)hipFree(e.p); if (e.ev) (void)hipEventDestroy(e.ev); }
        for (auto ev : evs) (void)hipEventDestroy(ev);
    }
};
static DevPool &dev_pool() { static DevPool *p = new DevPool(); return *p; }   // never destroyed: no HIP call at exit

// Destroyed plan objects, vectors emptied but not released (bt_plan::recycle).
struct PlanPool {
    std::mutex mu;
    std::vector<bt_plan *> idle;
    bt_plan *take() {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!idle.empty()) { bt_plan *p = idle.back(); idle.pop_back(); return p; }
        }
        return new (std::nothrow) bt_plan();
    }
    void give(bt_plan *p) {
        p->recycle();
        {
            std::lock_guard<std::mutex> lk(mu);
            if (idle.size() < 8) { idle.push_back(p); return; }
        }
        delete p;
    }
    void trim() {
        std::vector<bt_plan *> all;
        { std::lock_guard<std::mutex> lk(mu); all.swap(idle); }
        for (bt_plan *p : all) delete p;
    }
};
static PlanPool &plan_pool() { static PlanPool *p = new PlanPool(); return *p; }

// Plan transfers (index download, table upload) run on a stream of their own, created non-blocking: a plan can
// then be built from a second host thread while the caller's streams are busy — BATRACK knows the edge list of an
// update() a whole tracker pass before it calls the BA (batrack.py:983-993) — without the implicit waits a
// synchronous hipMemcpy on the null stream would bring.
static hipStream_t copy_stream() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    static std::mutex mu;
    static std::vector<std::pair<int, hipStream_t>> *streams = new std::vector<std::pair<int, hipStream_t>>();
    std::lock_guard<std::mutex> lk(mu);
    for (auto &e : *streams) if (e.first == dev) return e.second;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;   // null stream: still correct
    streams->emplace_back(dev, s);
    return s;
}

// Per host thread: device words + flag and their pinned host mirror for the packed edge list (grown, never shrunk)
struct PackBuffers {
    uint64_t *d_words = nullptr, *h_words = nullptr;
    int *d_bad = nullptr, *h_bad = nullptr;
    int *d_cmp = nullptr, *h_cmp = nullptr;                  // result of the shift comparison (bt_plan_create_shifted)
    size_t cap = 0;
    bool ensure(size_t n) {
        if (n <= cap && d_words) return true;
        if (d_words) { (void)hipFree(d_words); (void)hipHostFree(h_words); d_words = nullptr; h_words = nullptr; cap = 0; }
        const size_t want = n + n / 4 + 4096;
        if (hipMalloc(reinterpret_cast<void **>(&d_words), want * sizeof(uint64_t)) != hipSuccess) return false;
        if (hipHostMalloc(reinterpret_cast<void **>(&h_words), want * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) return false;
        if (!d_bad && hipMalloc(reinterpret_cast<void **>(&d_bad), sizeof(int)) != hipSuccess) return false;
        if (!h_bad && hipHostMalloc(reinterpret_cast<void **>(&h_bad), sizeof(int), hipHostMallocDefault) != hipSuccess) return false;
        cap = want;
        return true;
    }
};

static PackBuffers &pack_buffers() { static thread_local PackBuffers pb; return pb; }

static bool api_prof() { return plan_prof(); }
struct ApiTick {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void operator()(const char *what) {
        if (!api_prof()) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "plan api %s: %.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// PlanDev of a plan whose arrays sit at `d` in the layout recorded in pl->off (upload_plan, clone_plan_shifted)
static void bind_pointers(bt_plan *pl, const void *d) {
    const PlanOffsets &O = pl->off;
    const char *b = static_cast<const char *>(d);
    const bt_plan_info &I = pl->info;
    PlanDev &P = pl->dev;
    P.E = (int)I.E; P.n_buf = (int)I.n_buf; P.p_tot = (int)I.p_tot; P.fixedp = (int)I.fixedp;
    P.n_all = (int)I.n_all; P.n = (int)I.n; P.D = (int)(6 * I.n); P.m = (int)I.m; P.P = (int)I.pairs;
    P.T = (int)I.tiles; P.slots = (int)I.slots; P.erows = (int)I.erows; P.nnzb = (int)I.nnz_blocks;
    P.nupd = (int)I.updates; P.max_rows16 = pl->max_rows16; P.dev_id = pl->dev_id;
#define BT_I32(off) reinterpret_cast<const int32_t *>(b + (off))
    P.kx = BT_I32(O.kx);
    P.act_bits = reinterpret_cast<const uint32_t *>(b + O.ab); P.act_rank = BT_I32(O.ar);
    P.pair_i = BT_I32(O.pi); P.pair_j = BT_I32(O.pj);
    P.tile_trk0 = BT_I32(O.t0); P.tile_ntrk = BT_I32(O.tn); P.tile_ncam = BT_I32(O.tc); P.tile_cam0 = BT_I32(O.c0);
    P.tile_slot0 = BT_I32(O.s0); P.tile_nslot = BT_I32(O.sn); P.tile_erow0 = BT_I32(O.e0); P.tile_cams = BT_I32(O.cams);
    P.slot_edge = BT_I32(O.se); P.slot_pair = BT_I32(O.sp);
    P.slot_lab = reinterpret_cast<const uint16_t *>(b + O.sl);
    P.col_ptr = BT_I32(O.cp); P.row_idx = BT_I32(O.ri); P.upd_ptr = BT_I32(O.up); P.upd = BT_I32(O.u); P.blk_col = BT_I32(O.bc); P.upd_next = BT_I32(O.un);
    P.perm = BT_I32(O.pm); P.blk_src = BT_I32(O.bs); P.lvl_ptr = BT_I32(O.lp); P.lvl_cols = BT_I32(O.lc);
    P.col_lvl = BT_I32(O.cl); P.dp_ptr = BT_I32(O.dpp); P.dp = BT_I32(O.dp);
    P.nlev = pl->cnt_nlev; P.ndp = pl->cnt_ndp;
    P.lvl_meta = BT_I32(O.lm); P.tile_flags = BT_I32(O.tf);
    P.fz_pend_ptr = BT_I32(O.fpp); P.fz_pend = BT_I32(O.fp); P.fz_lazy_ptr = BT_I32(O.flp); P.fz_lazy = BT_I32(O.fl);
    P.fz_yurg = BT_I32(O.fy); P.fz_meta = BT_I32(O.fm); P.fz_pmeta = BT_I32(O.fpm); P.bs_sync = BT_I32(O.bss); P.fz_rowinfo = BT_I32(O.fri); P.fz_pfirst = BT_I32(O.fpf); P.fz_psecond = BT_I32(O.fps); P.tile_ij = BT_I32(O.tij); P.tile_kx = BT_I32(O.tkx);
    P.tile_cut8 = reinterpret_cast<const uint16_t *>(b + O.tc8); P.tile_cut16 = reinterpret_cast<const uint16_t *>(b + O.tc16);
    P.fz_npend = pl->cnt_npend; P.fz_nlazy = pl->cnt_nlazy; P.fz_ok = pl->fz_ok; P.fzp_ok = pl->fzp_ok;
    P.tile_pair0 = BT_I32(O.tp0); P.tile_npair = BT_I32(O.tnp); P.tile_pairs = BT_I32(O.tps);
    P.slot_lp = reinterpret_cast<const uint8_t *>(b + O.slp); P.max_tile_pairs = pl->max_tile_pairs; P.max_tile_slots = pl->max_tile_slots; P.max_cams = (int)I.max_tile_cams; P.e_all = pl->e_all;
    P.slot_code = reinterpret_cast<const uint16_t *>(b + O.sc); P.tile_la = reinterpret_cast<const uint8_t *>(b + O.tla); P.tile_rec = BT_I32(O.trec); P.it_edge = BT_I32(O.ite); P.tile_sinfo = reinterpret_cast<const uint32_t *>(b + O.tsi); P.em_ok = pl->em_ok; P.st_ok = pl->st_ok; P.st_min = pl->st_min; P.em_min = pl->em_min; P.em_its = (int)pl->em_its; P.em_lgs = pl->em_lgs; P.em_self = pl->em_self;
    P.pm_edge = BT_I32(O.pme); P.pm_rec = BT_I32(O.pmr); P.pm_lb = reinterpret_cast<const uint8_t *>(b + O.pmb); P.pm_la = reinterpret_cast<const uint8_t *>(b + O.pml); P.pm_ok = pl->pm_ok; P.sp_ok = pl->sp_ok; P.wide = pl->wide; P.trk_off = pl->trk_off; P.pp_ptr = BT_I32(O.ppp); P.pp_idx = BT_I32(O.ppi); P.sg_ptr = BT_I32(O.sgp); P.sg_n = pl->sg_n; P.et_lgts = pl->et_lgts;
    P.lz_trk = BT_I32(O.lzt); P.lz_ptr = BT_I32(O.lzp); P.lz_edge = BT_I32(O.lze); P.lz_pair = BT_I32(O.lzq); P.nlz = pl->nlz;
#undef BT_I32
}

// Plans of up to this many edges keep their packed edge list on the device (8 bytes per edge) so that the next edge
// list can be recognised as a shifted copy (bt_plan_create_shifted)
constexpr int64_t kKeepPackedMaxEdges = 4 << 20;

int upload_plan(bt_plan *pl, const uint64_t *d_packed = nullptr) {
    ApiTick tick;
    std::vector<char> &buf = pl->stage;           // capacity survives with the recycled plan object
    PlanOffsets &O = pl->off;
    buf.clear();
    O.kx = put(buf, pl->kx), O.ab = put(buf, pl->act_bits), O.ar = put(buf, pl->act_rank);
    O.pi = put(buf, pl->pair_i), O.pj = put(buf, pl->pair_j);
    O.t0 = put(buf, pl->tile_trk0), O.tn = put(buf, pl->tile_ntrk), O.tc = put(buf, pl->tile_ncam);
    O.c0 = put(buf, pl->tile_cam0), O.s0 = put(buf, pl->tile_slot0), O.sn = put(buf, pl->tile_nslot);
    O.e0 = put(buf, pl->tile_erow0), O.cams = put(buf, pl->tile_cams);
    O.se = put(buf, pl->slot_edge), O.sp = put(buf, pl->slot_pair), O.sl = put(buf, pl->slot_lab);
    O.cp = put(buf, pl->col_ptr), O.ri = put(buf, pl->row_idx), O.up = put(buf, pl->upd_ptr), O.u = put(buf, pl->upd);
    O.bc = put(buf, pl->blk_col), O.un = put(buf, pl->upd_next);
    O.lm = put(buf, pl->lvl_meta), O.tf = put(buf, pl->tile_flags);
    O.tp0 = put(buf, pl->tile_pair0), O.tnp = put(buf, pl->tile_npair), O.tps = put(buf, pl->tile_pairs), O.slp = put(buf, pl->slot_lp);
    O.pm = put(buf, pl->perm), O.bs = put(buf, pl->blk_src), O.lp = put(buf, pl->lvl_ptr), O.lc = put(buf, pl->lvl_cols);
    O.cl = put(buf, pl->col_lvl), O.dpp = put(buf, pl->dp_ptr), O.dp = put(buf, pl->dp);
    O.fpp = put(buf, pl->fz_pend_ptr), O.fp = put(buf, pl->fz_pend), O.flp = put(buf, pl->fz_lazy_ptr), O.fl = put(buf, pl->fz_lazy);
    O.fy = put(buf, pl->fz_yurg), O.fm = put(buf, pl->fz_meta), O.fpm = put(buf, pl->fz_pmeta), O.bss = put(buf, pl->bs_sync), O.fri = put(buf, pl->fz_rowinfo), O.fpf = put(buf, pl->fz_pfirst), O.fps = put(buf, pl->fz_psecond), O.tij = put(buf, pl->tile_ij), O.tkx = put(buf, pl->tile_kx);
    O.tc8 = put(buf, pl->tile_cut8), O.tc16 = put(buf, pl->tile_cut16);
    O.sc = put(buf, pl->slot_code), O.tla = put(buf, pl->tile_la), O.trec = put(buf, pl->tile_rec), O.ite = put(buf, pl->it_edge), O.tsi = put(buf, pl->tile_sinfo);
    O.pme = put(buf, pl->pm_edge), O.pmr = put(buf, pl->pm_rec), O.pmb = put(buf, pl->pm_lb), O.pml = put(buf, pl->pm_la);
    O.ppp = put(buf, pl->pp_ptr), O.ppi = put(buf, pl->pp_idx), O.sgp = put(buf, pl->sg_ptr);
    O.lzt = put(buf, pl->lz_trk), O.lzp = put(buf, pl->lz_ptr), O.lze = put(buf, pl->lz_edge), O.lzq = put(buf, pl->lz_pair);
    pl->nlz = (int)pl->lz_trk.size();
    pl->sg_n = pl->sg_ptr.empty() ? 0 : (int)pl->sg_ptr.size() - 1;
    tick("pack arrays");
    size_t cap = 0;
    hipEvent_t reuse_after = nullptr;
    const bool keep_pk = d_packed && pl->e_all > 0 && pl->e_all <= kKeepPackedMaxEdges && pl->info.E == pl->e_all;
    // (a plan whose pm_edge is written on the device: the table lies behind the staged bytes, nothing of it crosses PCIe)
    size_t tables_end = buf.size();
    if (pl->dev_pm) { O.pme = (buf.size() + 255) / 256 * 256; tables_end = O.pme + (size_t)pl->pm_rounds * kLanes * sizeof(int32_t); }
    if (pl->dev_slots) {                                   // likewise the [slots][64] arrays of a 64-track layout
        const size_t ns = (size_t)pl->info.slots * kLanes;
        auto take = [&](size_t bytes) { const size_t o = (tables_end + 255) / 256 * 256; tables_end = o + bytes; return o; };
        O.se = take(ns * sizeof(int32_t)); O.sp = take(ns * sizeof(int32_t)); O.sl = take(ns * sizeof(uint16_t)); O.slp = take(ns);
        if (pl->dev_wpt) {                                 // ... and the tables of the wave-per-tile kernels
            O.sc = take(ns * sizeof(uint16_t)); O.tla = take((size_t)pl->info.tiles * kLanes);
            if (pl->em_ok) { O.ite = take((size_t)pl->em_its * kLanes * sizeof(int32_t)); O.tsi = take((size_t)pl->info.tiles * kLanes * sizeof(uint32_t)); }
        }
    }
    const size_t pk_off = (tables_end + 255) / 256 * 256, total = keep_pk ? pk_off + (size_t)pl->e_all * sizeof(uint64_t) : tables_end;
    void *d = dev_pool().acquire(total + 256, &cap, &reuse_after);
    if (!d) return BT_ENOMEM;
    if (hipGetDevice(&pl->dev_id) != hipSuccess) pl->dev_id = 0;      // (once per plan: its launches take their per-device figures from it)
    tick("device buffer");
    hipStream_t cs = copy_stream();
    // a recycled buffer may still be read by kernels of the destroyed plan queued on the caller's stream: the upload
    // is ordered behind them on the device (no host wait)
    const bool waited = !reuse_after || hipStreamWaitEvent(cs, reuse_after, 0) == hipSuccess;
    if (!waited) (void)hipEventSynchronize(reuse_after);
    dev_pool().give_event(reuse_after);
    bool ok = hipMemcpyAsync(d, buf.data(), buf.size(), hipMemcpyHostToDevice, cs) == hipSuccess &&
              (!keep_pk || hipMemcpyAsync(static_cast<char *>(d) + pk_off, d_packed, (size_t)pl->e_all * sizeof(uint64_t), hipMemcpyDeviceToDevice, cs) == hipSuccess);
    if (ok && (pl->dev_pm || pl->dev_slots)) {
        bind_pointers(pl, d);                          // (the fill kernels read the plan's tile tables in the buffer)
        char *db = static_cast<char *>(d);
        if (pl->dev_pm)
            ok = plan_device_fill(pl, pl->info.E, reinterpret_cast<int32_t *>(db + O.pmr), reinterpret_cast<int32_t *>(db + O.pme), pl->pm_rounds, cs) == BT_OK;
        else {
            DevWptOut w{};
            if (pl->dev_wpt) {
                w.slot_code = reinterpret_cast<uint16_t *>(db + O.sc); w.tile_la = reinterpret_cast<uint8_t *>(db + O.tla); w.tile_rec = reinterpret_cast<int32_t *>(db + O.trec);
                if (pl->em_ok) { w.it_edge = reinterpret_cast<int32_t *>(db + O.ite); w.tile_sinfo = reinterpret_cast<uint32_t *>(db + O.tsi); w.its = pl->em_its; }
            }
            ok = plan_device_slots_fill(pl, pl->info.E, reinterpret_cast<int32_t *>(db + O.se), reinterpret_cast<int32_t *>(db + O.sp),
                                        reinterpret_cast<uint16_t *>(db + O.sl), reinterpret_cast<uint8_t *>(db + O.slp),
                                        reinterpret_cast<uint16_t *>(db + O.tc8), reinterpret_cast<uint16_t *>(db + O.tc16), cs, pl->dev_wpt ? &w : nullptr) == BT_OK;
        }
    }
    if (!ok || hipStreamSynchronize(cs) != hipSuccess) { dev_pool().release(d, cap, false, nullptr); return BT_EHIP; }
    // (k_plan_sinfo's verdict: a tile whose tracks do not share their slots' pairs leaves the plan with k_stream)
    if (pl->dev_slots && pl->dev_wpt && pl->em_ok && plan_device_em_verdict()) { pl->em_ok = 0; pl->em_its = 0; pl->em_lgs = -1; }
    tick("H2D copy");
    pl->dev_base = d;
    pl->dev_cap = cap;
    pl->dev_bytes = total;
    pl->pk_off = keep_pk ? pk_off : 0;
    pl->n_act_words = (int)pl->act_bits.size(); pl->n_tile_ij = (int)pl->tile_ij.size();
    pl->cnt_nlev = (int)pl->lvl_ptr.size() - 1; pl->cnt_ndp = (int)pl->dp.size();
    pl->cnt_npend = (int)(pl->fz_pend.size() / 2); pl->cnt_nlazy = (int)(pl->fz_lazy.size() / 3);
    bind_pointers(pl, d);
    PlanDev &P = pl->dev;
    const int rc = configure_kernels(P);
    tick("configure kernels");
    return rc;
}

static StepArgs make_args(const bt_plan *pl, const bt_ba_args *a, void *ws) {
    char *w = static_cast<char *>(ws);
    const WsLayout &L = pl->ws;
    StepArgs s{};
    s.poses = a->poses; s.patches = a->patches; s.mono = a->mono_disp; s.intr = a->intrinsics;
    s.targets = a->targets; s.weights = a->weights; s.tstride = (int)a->target_stride; s.mstride = a->mono_stride > 1 ? (int)a->mono_stride : 1;
    s.poses_out = a->poses_out; s.patches_out = a->patches_out;
    s.b0 = a->bounds[0]; s.b1 = a->bounds[1]; s.b2 = a->bounds[2]; s.b3 = a->bounds[3];
    s.lmbda = a->lmbda; s.ep = a->ep; s.alpha = a->alpha; s.loss = a->loss;
    s.lmbda_trk = a->lmbda_per_track;
    const size_t D = (size_t)(6 * pl->info.n);
    s.S = reinterpret_cast<double *>(w + L.sys); s.y = s.S + D * D;
    s.pairacc = reinterpret_cast<double *>(w + L.pairacc);
    // (k_edge2's private copies: only where k_edge2 is the plan's Jacobian kernel — k_pair_finalize adds up what it is given)
    s.priv = L.priv && edge_applies(pl->dev) ? reinterpret_cast<double *>(w + L.priv) : nullptr;
    s.pairgeo = reinterpret_cast<float *>(w + L.pairgeo);
    s.packed = reinterpret_cast<double *>(w + L.packed); s.qw = reinterpret_cast<float2 *>(w + L.qw);
    s.lfac = reinterpret_cast<float *>(w + L.lfac);
    s.linv = reinterpret_cast<float *>(w + L.linv); s.zvec = reinterpret_cast<float *>(w + L.zvec);
    s.dx = reinterpret_cast<float *>(w + L.dx); s.dx0 = reinterpret_cast<float *>(w + L.dx0); s.status = reinterpret_cast<int *>(w + L.status);
    s.spart = reinterpret_cast<double *>(w + L.spart);
    s.esave = reinterpret_cast<double *>(w + L.esave);
    static const int dbg = std::getenv("BT_DEBUG_MODE") ? std::atoi(std::getenv("BT_DEBUG_MODE")) : 0;
    s.dbg = dbg;
    s.prec = edge_precision(pl->dev);
    return s;
}

// Every entry point that launches a plan's kernels passes here first: the stream is remembered for bt_plan_destroy, and a clone
// whose tables are still being copied on the plan stream (bt_plan_create_shifted_any) gets its launches ordered behind them.
static void mark_launch(const bt_plan *pl, void *stream) {
    pl->last_stream = stream; pl->launched = true;
    if (void *r = pl->ready.load(std::memory_order_acquire)) {
        hipEvent_t ev = static_cast<hipEvent_t>(r);
        if (hipEventQuery(ev) == hipSuccess) {
            if (pl->ready.exchange(nullptr, std::memory_order_acq_rel) == r) dev_pool().give_event(ev);      // (one claimant)
        }
        else if (hipStreamWaitEvent(static_cast<hipStream_t>(stream), ev, 0) != hipSuccess) (void)hipEventSynchronize(ev);
    }
}

static int check(const bt_plan *pl, const bt_ba_args *a, const void *ws) {
    if (!pl || !a || !ws || !pl->dev_base) return BT_EINVAL;
    if (pl->spec_unbound) return BT_EINVAL;                    // (a clone made ahead of its list, never given it: bt_plan_spec_bind)
    if (!a->poses || !a->patches || !a->mono_disp || !a->intrinsics || !a->patches_out) return BT_EINVAL;
    if (pl->info.E > 0 && (!a->targets || !a->weights || a->target_stride < 2)) return BT_EINVAL;
    if (a->loss < BT_LOSS_TRIVIAL || a->loss > BT_LOSS_CAUCHY) return BT_EINVAL;
    if (a->mono_stride < 0) return BT_EINVAL;                 // 0 and 1 both mean contiguous; a prior of one repeated element must be materialised
    return BT_OK;
}

// ba.py:316: "structure_only or n == 0" take the same branch
static bool is_so(const bt_plan *pl, const bt_ba_args *a) { return a->structure_only != 0 || pl->info.n == 0; }

}  // namespace bt

using namespace bt;

extern "C" {

int bt_version(void) { return BT_VERSION; }
int bt_plan_jacobian_kernel(const bt_plan *pl) {
    if (!pl || !pl->dev_base) return -1;
    return edge_applies(pl->dev) ? 2 : stream_applies(pl->dev) ? 1 : etile_precision_bytes(pl->dev) ? 3 : 0;
}
int bt_plan_built_on_device(const bt_plan *pl) { return pl && (pl->dev_pm || pl->dev_slots) ? 1 : 0; }

int bt_plan_edge_precision(const bt_plan *pl) {
    if (!pl || !pl->dev_base) return -1;
    if (edge_precision(pl->dev)) return 8;
    return (pl->dev.T > 0 && (edge_applies(pl->dev) || stream_applies(pl->dev))) ? 6 : 4;     // 6: mixed (ba_edge.hpp: edge_eval_mixed)
}
const char *bt_target_arch(void) { return "gfx950"; }

int bt_plan_create(const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E, int64_t n_buf,
                   int64_t p_tot, int64_t fixedp, int64_t n_all_min, int64_t own_lo, int64_t own_hi,
                   int on_device, int upload, bt_plan **out) {
    if (!out || E < 0 || (E > 0 && (!ii || !jj || !kk))) return BT_EINVAL;
    *out = nullptr;
    const uint64_t *packed = nullptr;
    ApiTick tick;
    if (on_device && E > 0) {
        // packed and range-checked on the device (8 of the 24 bytes per edge cross PCIe), into a pinned host buffer
        if (n_buf > 32768 || p_tot > (int64_t)0x7fffffff) return BT_EUNSUPPORTED;      // tile_ij packs two frame numbers into a signed 32-bit word
        PackBuffers &pb = pack_buffers();
        if (!pb.ensure((size_t)E)) return BT_ENOMEM;
        hipStream_t cs = copy_stream();
        if (hipMemsetAsync(pb.d_bad, 0, sizeof(int), cs) != hipSuccess ||
            launch_pack_edges(ii, jj, kk, E, n_buf, p_tot, pb.d_words, pb.d_bad, cs) != BT_OK)
            return BT_EHIP;
        // window plans: the passes over the edges stay on the device (plan_device.hip), the host lays out what is small
        const bool dev_planner = !force().host_plan;
        if (dev_planner && upload && E >= 4096) {
            if (hipMemcpyAsync(pb.h_bad, pb.d_bad, sizeof(int), hipMemcpyDeviceToHost, cs) != hipSuccess) return BT_EHIP;
            DevPlanStats st{};
            int64_t tracks = 0;
            int rc = plan_device_stats(pb.d_words, E, p_tot, fixedp, own_lo, own_hi, cs, &st, &tracks);       // (rc == BT_OK: it synchronised, h_bad is in as well)
            if (rc != BT_OK && rc != BT_NEED_EDGES) return rc;
            // (BT_NEED_EDGES can come back before anything was waited for — p_tot or E beyond the device path — and h_bad may
            //  then still hold an earlier list's verdict: it is read only behind a synchronisation; the host path below does its own)
            if (rc == BT_OK && *pb.h_bad) return BT_EINVAL;
            tick("device: per-track figures");
            if (rc == BT_OK) {
                bt_plan *pl = plan_pool().take();
                if (!pl) return BT_ENOMEM;
                try {
                    rc = build_plan_host(nullptr, nullptr, nullptr, E, n_buf, p_tot, fixedp, n_all_min, own_lo, own_hi, pl, nullptr, false, &st);
                    tick("host analysis (no edges)");
                    int64_t rounds = 0;
                    if (rc == BT_OK) rc = pl->dev_pm ? plan_device_rounds(pl, pl->info.E, cs, &rounds) : plan_device_slots_stage(pl, cs);
                    tick("device: rounds");
                    if (rc == BT_OK) { pl->pm_rounds = rounds; rc = upload_plan(pl, pb.d_words); }
                } catch (const std::bad_alloc &) {
                    rc = BT_ENOMEM;
                }
                if (rc == BT_OK) { *out = pl; return BT_OK; }
                bt_plan_destroy(pl);
                if (rc != BT_NEED_EDGES) return rc;
            }
        }
        if (hipMemcpyAsync(pb.h_words, pb.d_words, (size_t)E * sizeof(uint64_t), hipMemcpyDeviceToHost, cs) != hipSuccess ||
            hipMemcpyAsync(pb.h_bad, pb.d_bad, sizeof(int), hipMemcpyDeviceToHost, cs) != hipSuccess ||
            hipStreamSynchronize(cs) != hipSuccess)
            return BT_EHIP;
        if (*pb.h_bad) return BT_EINVAL;
        packed = pb.h_words;
        tick("D2H indices");
    }
    bt_plan *pl = plan_pool().take();
    if (!pl) return BT_ENOMEM;
    int rc = BT_OK;
    try {
        rc = build_plan_host(packed ? nullptr : ii, packed ? nullptr : jj, packed ? nullptr : kk, E, n_buf, p_tot, fixedp, n_all_min, own_lo, own_hi, pl, packed, /*keep_slots=*/!upload);
        tick("host analysis");
        if (rc == BT_OK && upload) rc = upload_plan(pl, packed ? pack_buffers().d_words : nullptr);
    } catch (const std::bad_alloc &) {
        rc = BT_ENOMEM;
    }
    if (rc != BT_OK) { bt_plan_destroy(pl); return rc; }
    *out = pl;
    return BT_OK;
}

// The clone of `src` for a list shifted by di frames / dk patches: tables copied and shifted on the plan stream `cs` (the new list's
// packed words, d_words, behind them), no host wait — the plan carries the event its first launches are ordered behind.
static int clone_shifted(const bt_plan *src, const uint64_t *d_words, int64_t E, int64_t fixedp, int64_t di, int64_t dk, hipStream_t cs, bt_plan **out) {
    ApiTick tick;
    bt_plan *pl = plan_pool().take();
    if (!pl) return BT_ENOMEM;
    pl->info = src->info; pl->info.fixedp = fixedp; pl->info.n_all = src->info.n_all + di;
    pl->ws = src->ws; pl->off = src->off; pl->dev_bytes = src->dev_bytes; pl->pk_off = src->pk_off;
    pl->n_act_words = src->n_act_words; pl->n_tile_ij = src->n_tile_ij;
    pl->cnt_nlev = src->cnt_nlev; pl->cnt_ndp = src->cnt_ndp; pl->cnt_npend = src->cnt_npend; pl->cnt_nlazy = src->cnt_nlazy;
    pl->max_rows16 = src->max_rows16; pl->max_tile_pairs = src->max_tile_pairs; pl->max_tile_slots = src->max_tile_slots;
    pl->fz_ok = src->fz_ok; pl->fzp_ok = src->fzp_ok; pl->em_ok = src->em_ok; pl->st_ok = src->st_ok; pl->st_min = src->st_min; pl->em_min = src->em_min; pl->em_its = src->em_its; pl->em_lgs = src->em_lgs;
    pl->em_self = src->em_self; pl->e_all = src->e_all; pl->pm_ok = src->pm_ok; pl->sp_ok = src->sp_ok; pl->wide = src->wide; pl->nlz = src->nlz; pl->dev_id = src->dev_id; pl->sg_n = src->sg_n; pl->et_lgts = src->et_lgts; pl->dev_pm = 0; pl->dev_slots = 0; pl->dev_wpt = 0; pl->trk_off = src->trk_off; pl->pm_rounds = src->pm_rounds; pl->k_hi = src->k_hi >= 0 ? src->k_hi + dk : -1;
    size_t cap = 0;
    hipEvent_t reuse_after = nullptr;
    void *d = dev_pool().acquire(pl->dev_bytes + 256, &cap, &reuse_after);
    if (!d) { plan_pool().give(pl); return BT_ENOMEM; }
    const bool waited = !reuse_after || hipStreamWaitEvent(cs, reuse_after, 0) == hipSuccess;
    if (!waited) (void)hipEventSynchronize(reuse_after);
    dev_pool().give_event(reuse_after);
    char *nb = static_cast<char *>(d);
    const char *ob = static_cast<const char *>(src->dev_base);
    const PlanOffsets &O = pl->off;
    auto I32 = [&](size_t off) { return reinterpret_cast<int32_t *>(nb + off); };
    int rc = BT_OK;
    // the tables as they are, the new packed edge list behind them, then the ones that hold absolute frame / patch numbers
    // (d_words null — a clone made ahead of its list, bt_plan_preshift: the list it expects, the source's words moved by the shift)
    if (hipMemcpyAsync(nb, ob, src->pk_off, hipMemcpyDeviceToDevice, cs) != hipSuccess) rc = BT_EHIP;
    else if (d_words) { if (hipMemcpyAsync(nb + pl->pk_off, d_words, (size_t)E * sizeof(uint64_t), hipMemcpyDeviceToDevice, cs) != hipSuccess) rc = BT_EHIP; }
    else rc = launch_words_add(reinterpret_cast<const uint64_t *>(ob + src->pk_off), ((uint64_t)dk << 32) | ((uint64_t)di << 16) | (uint64_t)di,
                               reinterpret_cast<uint64_t *>(nb + pl->pk_off), E, cs);
    if (rc == BT_OK)
        rc = launch_plan_shift(I32(O.kx), (int)pl->info.m, I32(O.tkx), (int)pl->info.tiles * kLanes, I32(O.tij), pl->n_tile_ij, I32(O.pi), I32(O.pj),
                               (int)pl->info.pairs, reinterpret_cast<const uint32_t *>(ob + O.ab), reinterpret_cast<uint32_t *>(nb + O.ab), I32(O.ar),
                               pl->n_act_words, (int)di, (int)dk, cs);
    // no host wait for the copies: the plan's first launches are ordered behind this event on whatever stream they use
    hipEvent_t ready = rc == BT_OK ? dev_pool().take_event() : nullptr;
    if (rc == BT_OK && (!ready || hipEventRecord(ready, cs) != hipSuccess)) {
        dev_pool().give_event(ready); ready = nullptr;
        if (hipStreamSynchronize(cs) != hipSuccess) rc = BT_EHIP;
    }
    tick("shifted: copies enqueued");
    if (rc != BT_OK) { (void)hipStreamSynchronize(cs); dev_pool().release(d, cap, false, nullptr); plan_pool().give(pl); return rc; }
    pl->ready = ready;
    pl->dev_base = d;
    pl->dev_cap = cap;
    bind_pointers(pl, d);
    rc = configure_kernels(pl->dev);
    tick("shifted: configure kernels");
    if (rc != BT_OK) { bt_plan_destroy(pl); return rc; }
    *out = pl;
    return BT_OK;
}

// Verdicts of speculative clones: pinned / device int pairs handed out round robin (a plan's verdict is read by
// bt_plan_spec_confirm right after its first step; sixteen unconfirmed speculative plans at a time are sixteen more than the
// caller has)
struct SpecSlots {
    int *h = nullptr;                                                        // pinned, mapped: the comparison kernel writes its verdict there (4 ints a slot)
    unsigned *tickets[kMaxDevices] = {};                                     // device memory, per device: one counter a slot (k_match_done's last-workgroup ticket)
    hipEvent_t ev_in = nullptr;                                              // orders the plan stream behind the caller's stream
    unsigned next = 0;
    bool ensure() {
        if (h && ev_in) return true;
        if (!h) { if (hipHostMalloc(reinterpret_cast<void **>(&h), 16 * 4 * sizeof(int), hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) return false; std::memset(h, 0, 16 * 4 * sizeof(int)); }
        return ev_in || hipEventCreateWithFlags(&ev_in, hipEventDisableTiming) == hipSuccess;
    }
    // the tickets of the device a plan lives on (allocated on first use, with that device current — as it is for every launch of the plan)
    unsigned *tickets_of(int dev) {
        if (dev < 0 || dev >= kMaxDevices) return nullptr;
        if (!tickets[dev] && (hipMalloc(reinterpret_cast<void **>(&tickets[dev]), 16 * sizeof(unsigned)) != hipSuccess ||
                              hipMemset(tickets[dev], 0, 16 * sizeof(unsigned)) != hipSuccess)) { tickets[dev] = nullptr; return nullptr; }
        return tickets[dev];
    }
};
static SpecSlots &spec_slots() { static SpecSlots s; return s; }
static std::mutex &spec_mutex() { static std::mutex m; return m; }

int bt_plan_create_shifted_spec(const bt_plan *src, const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E,
                                int64_t n_buf, int64_t p_tot, int64_t fixedp, void *in_stream, bt_plan **out) {
    if (!out) return BT_EINVAL;
    *out = nullptr;
    if (!src || !ii || !jj || !kk || E <= 0) return BT_EINVAL;
    if (n_buf > 32768 || p_tot > (int64_t)0x7fffffff || n_buf <= 0 || p_tot % n_buf != 0) return BT_NO_MATCH;
    if (!src->dev_base || !src->pk_off || src->e_all != E || src->info.E != E || src->info.n_buf != n_buf || src->info.p_tot != p_tot) return BT_NO_MATCH;
    if (src->spec_ev || src->spec_epoch || src->spec_unbound) return BT_NO_MATCH;   // (an unconfirmed speculation is no source)
    const int64_t di = fixedp - src->info.fixedp, dk = di * (p_tot / n_buf);
    // every number the clone's tables will hold stays inside the caller's buffers, whatever the new list turns out to be
    if (di <= 0 || di >= 32768 || src->info.n_all + di > n_buf || src->k_hi < 0 || src->k_hi + dk >= p_tot) return BT_NO_MATCH;
    ApiTick tick;
    PackBuffers &pb = pack_buffers();
    if (!pb.ensure((size_t)E)) return BT_ENOMEM;
    int *h_flag;
    hipStream_t cs = copy_stream();
    {
        std::lock_guard<std::mutex> g(spec_mutex());
        SpecSlots &ss = spec_slots();
        if (!ss.ensure()) return BT_ENOMEM;
        h_flag = ss.h + 4 * (ss.next++ % 16);
        // the index tensors are complete when `in_stream` gets here: the plan stream waits for that, the host does not
        if (hipEventRecord(ss.ev_in, static_cast<hipStream_t>(in_stream)) != hipSuccess || hipStreamWaitEvent(cs, ss.ev_in, 0) != hipSuccess) {
            if (hipStreamSynchronize(static_cast<hipStream_t>(in_stream)) != hipSuccess) return BT_EHIP;
        }
    }
    const uint64_t *old_words = reinterpret_cast<const uint64_t *>(static_cast<const char *>(src->dev_base) + src->pk_off);
    const uint64_t delta = ((uint64_t)dk << 32) | ((uint64_t)di << 16) | (uint64_t)di;
    h_flag[0] = 0; h_flag[1] = 0;
    hipEvent_t spec_ev = dev_pool().take_event();
    if (!spec_ev) return BT_ENOMEM;
    const bool ok = launch_pack_match_expect(ii, jj, kk, E, n_buf, p_tot, old_words, delta, pb.d_words, h_flag, cs) == BT_OK &&
                    hipEventRecord(spec_ev, cs) == hipSuccess;
    if (!ok) { (void)hipStreamSynchronize(cs); dev_pool().give_event(spec_ev); return BT_EHIP; }
    tick("speculative shift: pack + match enqueued");
    bt_plan *pl = nullptr;
    const int rc = clone_shifted(src, pb.d_words, E, fixedp, di, dk, cs, &pl);
    if (rc != BT_OK) { (void)hipEventSynchronize(spec_ev); dev_pool().give_event(spec_ev); return rc; }
    pl->spec_ev = spec_ev; pl->spec_flag = h_flag;
    *out = pl;
    return BT_OK;
}

// The two halves of bt_plan_create_shifted_spec apart.  The caller's window moves by the same number of frames update() after
// update(): the clone for the NEXT list can be made while the steps of the current one run (its tables depend on the source and
// the shift only), and what is left for the call that brings the list — the first of an update(), with the GPU idle behind it — is
// the comparison kernel.
int bt_plan_preshift(const bt_plan *src, int64_t df, bt_plan **out) {
    if (!out) return BT_EINVAL;
    *out = nullptr;
    if (!src) return BT_EINVAL;
    const int64_t n_buf = src->info.n_buf, p_tot = src->info.p_tot, E = src->e_all;
    if (!src->dev_base || !src->pk_off || E <= 0 || src->info.E != E || src->spec_ev || src->spec_epoch || src->spec_unbound) return BT_NO_MATCH;
    if (n_buf <= 0 || n_buf > 32768 || p_tot > (int64_t)0x7fffffff || p_tot % n_buf != 0) return BT_NO_MATCH;
    const int64_t dk = df * (p_tot / n_buf);
    if (df <= 0 || df >= 32768 || src->info.n_all + df > n_buf || src->k_hi < 0 || src->k_hi + dk >= p_tot) return BT_NO_MATCH;
    bt_plan *pl = nullptr;
    const int rc = clone_shifted(src, nullptr, E, src->info.fixedp + df, df, dk, copy_stream(), &pl);
    if (rc != BT_OK) return rc;
    pl->spec_unbound = 1;
    *out = pl;
    return BT_OK;
}

int bt_plan_spec_bind(bt_plan *pl, const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E, int64_t n_buf, int64_t p_tot,
                      int64_t fixedp, void *in_stream) {
    if (!pl || !ii || !jj || !kk) return BT_EINVAL;
    if (!pl->spec_unbound || !pl->dev_base) return BT_EINVAL;
    if (pl->e_all != E || pl->info.n_buf != n_buf || pl->info.p_tot != p_tot || pl->info.fixedp != fixedp) return BT_NO_MATCH;
    ApiTick tick;
    int *h_flag;
    unsigned *ticket;
    int epoch;
    {
        std::lock_guard<std::mutex> g(spec_mutex());
        SpecSlots &ss = spec_slots();
        if (!ss.ensure()) return BT_ENOMEM;
        unsigned *tk = ss.tickets_of(pl->dev_id);
        if (!tk) return BT_ENOMEM;
        const unsigned slot = ss.next++ % 16;
        h_flag = ss.h + 4 * slot; ticket = tk + slot;
        epoch = (int)((ss.next & 0x3fffffffu) | 0x40000000u);          // (never 0, never the slot's previous one)
    }
    h_flag[0] = 0; h_flag[1] = 0; h_flag[2] = 0;
    // the list against the one the clone was made for (its own packed words): ONE launch, on the stream that made the index tensors
    // and will run the step — no cross-stream order to set up, no event to record (the verdict is polled: bt_plan_spec_confirm)
    const uint64_t *own = reinterpret_cast<const uint64_t *>(static_cast<const char *>(pl->dev_base) + pl->pk_off);
    if (launch_match_done(ii, jj, kk, E, n_buf, p_tot, own, h_flag, ticket, epoch, in_stream) != BT_OK) return BT_EHIP;
    tick("bind: comparison launched");
    pl->spec_flag = h_flag; pl->spec_epoch = epoch; pl->spec_stream = in_stream; pl->spec_unbound = 0;
    return BT_OK;
}

int bt_plan_spec_confirm(bt_plan *pl) {
    if (!pl) return BT_EINVAL;
    if (pl->spec_unbound) return BT_NO_MATCH;                    // (never given its list)
    if (pl->spec_epoch) {
        // (bt_plan_spec_bind: the comparison ran in front of the step that was just enqueued; its last workgroup wrote the epoch)
        volatile int *f = pl->spec_flag;
        const int want = pl->spec_epoch;
        bool done = false;
        for (int spins = 0; spins < (1 << 17) && !done; ++spins) done = f[2] == want;      // (~ 2 ms)
        if (!done) {
            // a GPU far behind — the caller's earlier work sits in front of the comparison: keep looking, but let the core go in
            // between (returning when the VERDICT is there, not when the step behind it is done, keeps the next call's launches
            // under that step); after seconds of that, the stream itself
            const auto t0 = std::chrono::steady_clock::now();
            while (!(done = f[2] == want) && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(5)) std::this_thread::yield();
            if (!done) {
                if (hipStreamSynchronize(static_cast<hipStream_t>(pl->spec_stream)) != hipSuccess) return BT_EHIP;
                done = f[2] == want;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        const int differs = f[0], bad = f[1];
        pl->spec_epoch = 0; pl->spec_flag = nullptr; pl->spec_stream = nullptr;
        if (!done) return BT_EHIP;
        if (bad) return BT_EINVAL;
        return differs ? BT_NO_MATCH : BT_OK;
    }
    if (!pl->spec_ev) return BT_OK;
    hipEvent_t ev = static_cast<hipEvent_t>(pl->spec_ev);
    const bool waited = hipEventSynchronize(ev) == hipSuccess;
    dev_pool().give_event(ev);
    pl->spec_ev = nullptr;
    if (!waited) return BT_EHIP;
    const int differs = pl->spec_flag[0], bad = pl->spec_flag[1];
    pl->spec_flag = nullptr;
    if (bad) return BT_EINVAL;
    return differs ? BT_NO_MATCH : BT_OK;
}

// The plan of a list that is a shifted copy of ONE of up to four earlier plans' lists (the caller's cache, most likely first):
// packed once, compared with every candidate in the same queue, one synchronisation for all of them (a candidate that does not
// match used to cost a round trip of its own: with a keyframe stride of 2 every other update of the replay paid one).  The clone's
// tables are copied and shifted on the plan stream WITHOUT a host wait: the plan carries an event (`ready`) that its first
// launches are ordered behind (mark_launch).
constexpr int kMaxShiftSources = 4;
int bt_plan_create_shifted_any(const bt_plan *const *srcs, int nsrc, const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E,
                               int64_t n_buf, int64_t p_tot, int64_t fixedp, int *which, bt_plan **out) {
    if (!out) return BT_EINVAL;
    *out = nullptr;
    if (which) *which = -1;
    if (!srcs || nsrc <= 0 || !ii || !jj || !kk || E <= 0) return BT_EINVAL;
    if (n_buf > 32768 || p_tot > (int64_t)0x7fffffff) return BT_NO_MATCH;
    const bt_plan *cand[kMaxShiftSources];
    int idx[kMaxShiftSources], nc = 0;
    for (int q = 0; q < nsrc && nc < kMaxShiftSources; ++q) {
        const bt_plan *src = srcs[q];
        if (!src) return BT_EINVAL;
        if (!src->dev_base || !src->pk_off || src->e_all != E || src->info.E != E || src->info.n_buf != n_buf || src->info.p_tot != p_tot) continue;
        if (src->spec_ev || src->spec_epoch || src->spec_unbound) continue;
        cand[nc] = src; idx[nc++] = q;
    }
    if (nc == 0) return BT_NO_MATCH;
    ApiTick tick;
    PackBuffers &pb = pack_buffers();
    if (!pb.ensure((size_t)E)) return BT_ENOMEM;
    constexpr int kCmpInts = 4 * kMaxShiftSources + 4;            // per candidate: mismatch, -, shift lo, shift hi; then the range check's verdict
    if (!pb.d_cmp && (hipMalloc(reinterpret_cast<void **>(&pb.d_cmp), kCmpInts * sizeof(int)) != hipSuccess ||
                      hipHostMalloc(reinterpret_cast<void **>(&pb.h_cmp), kCmpInts * sizeof(int), hipHostMallocDefault) != hipSuccess)) return BT_ENOMEM;
    hipStream_t cs = copy_stream();
    int *d_bad = pb.d_cmp + 4 * kMaxShiftSources;
    bool ok = hipMemsetAsync(pb.d_cmp, 0, kCmpInts * sizeof(int), cs) == hipSuccess &&
              launch_pack_edges(ii, jj, kk, E, n_buf, p_tot, pb.d_words, d_bad, cs) == BT_OK;
    for (int q = 0; ok && q < nc; ++q) {
        const uint64_t *old_words = reinterpret_cast<const uint64_t *>(static_cast<const char *>(cand[q]->dev_base) + cand[q]->pk_off);
        ok = launch_shift_match(pb.d_words, old_words, E, pb.d_cmp + 4 * q, cs) == BT_OK;
    }
    if (!ok || hipMemcpyAsync(pb.h_cmp, pb.d_cmp, kCmpInts * sizeof(int), hipMemcpyDeviceToHost, cs) != hipSuccess ||
        hipStreamSynchronize(cs) != hipSuccess)
        return BT_EHIP;
    tick("shifted: pack + match (synchronised)");
    if (pb.h_cmp[4 * kMaxShiftSources]) return BT_EINVAL;
    const bt_plan *src = nullptr;
    int64_t di = 0, dk = 0;
    for (int q = 0; q < nc && !src; ++q) {
        const int *c = pb.h_cmp + 4 * q;
        if (c[0]) continue;
        const uint64_t d0 = (uint64_t)(uint32_t)c[2] | ((uint64_t)(uint32_t)c[3] << 32);
        const int64_t dj = (int64_t)(d0 & 0xffff);
        di = (int64_t)((d0 >> 16) & 0xffff); dk = (int64_t)(d0 >> 32);
        // (shifts forward in time only; both frame fields by the same amount; the fixed prefix moves along)
        if (di != dj || dk >= ((int64_t)1 << 31) || fixedp != cand[q]->info.fixedp + di || cand[q]->info.n_all + di > n_buf) continue;
        if (di == 0 && dk == 0) continue;                           // the same list: the caller's cache has that plan already
        src = cand[q];
        if (which) *which = idx[q];
    }
    if (!src) return BT_NO_MATCH;
    return clone_shifted(src, pb.d_words, E, fixedp, di, dk, cs, out);
}

int bt_plan_create_shifted(const bt_plan *src, const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E,
                           int64_t n_buf, int64_t p_tot, int64_t fixedp, bt_plan **out) {
    if (!out) return BT_EINVAL;
    *out = nullptr;
    if (!src) return BT_EINVAL;
    return bt_plan_create_shifted_any(&src, 1, ii, jj, kk, E, n_buf, p_tot, fixedp, nullptr, out);
}

void bt_plan_destroy(bt_plan *pl) {
    if (!pl) return;
    if (pl->spec_ev) { (void)hipEventSynchronize(static_cast<hipEvent_t>(pl->spec_ev)); dev_pool().give_event(static_cast<hipEvent_t>(pl->spec_ev)); pl->spec_ev = nullptr; pl->spec_flag = nullptr; }
    if (pl->spec_epoch) { (void)hipStreamSynchronize(static_cast<hipStream_t>(pl->spec_stream)); pl->spec_epoch = 0; pl->spec_flag = nullptr; pl->spec_stream = nullptr; }   // (a comparison that reads this plan's words may still be queued)
    if (void *r = pl->ready.exchange(nullptr, std::memory_order_acq_rel)) {
        // (copies of a clone that was never launched may still be queued on the plan stream: the buffer's next owner writes it
        //  on that same stream, behind them)
        dev_pool().give_event(static_cast<hipEvent_t>(r));
    }
    if (pl->dev_base) dev_pool().release(pl->dev_base, pl->dev_cap, pl->launched, static_cast<hipStream_t>(pl->last_stream));
    plan_pool().give(pl);
}

void bt_plan_pool_trim(void) { dev_pool().trim(); plan_pool().trim(); }

int bt_plan_get_info(const bt_plan *pl, bt_plan_info *info) {
    if (!pl || !info) return BT_EINVAL;
    *info = pl->info;
    return BT_OK;
}

size_t bt_plan_workspace_bytes(const bt_plan *pl) { return pl ? pl->ws.total : 0; }

int64_t bt_plan_array(const bt_plan *pl, const char *name, const void **data) {
    if (!pl || !name || !data) return -1;
#define BT_ARR(n)                                                        \
    if (std::strcmp(name, #n) == 0) { *data = pl->n.data(); return (int64_t)pl->n.size(); }
    if (std::strcmp(name, "trk_of_patch") == 0) {            // kept for the window of patches only: expanded here (tests, tooling)
        pl->trk_of_patch.assign((size_t)pl->info.p_tot, -1);
        for (size_t i = 0; i < pl->trk_win.size(); ++i) pl->trk_of_patch[(size_t)pl->trk_win_lo + i] = pl->trk_win[i];
        *data = pl->trk_of_patch.data();
        return (int64_t)pl->trk_of_patch.size();
    }
    if (std::strcmp(name, "trk_off") == 0) {                  // sharded: distinct tracks in front of the rank's range (one element)
        pl->dev_readback.assign(1, (int32_t)pl->trk_off);
        *data = pl->dev_readback.data();
        return 1;
    }
    if (pl->dev_slots && pl->dev_base) {
        // a 64-track plan whose slot arrays and wave cuts were written on the device: read back on request (tests, tooling)
        const size_t ns = (size_t)pl->info.slots * kLanes, T = (size_t)pl->info.tiles;
        size_t off = 0, bytes = 0, esz = 4;
        if (std::strcmp(name, "slot_edge") == 0) { off = pl->off.se; bytes = ns * 4; }
        else if (std::strcmp(name, "slot_pair") == 0) { off = pl->off.sp; bytes = ns * 4; }
        else if (std::strcmp(name, "slot_lab") == 0) { off = pl->off.sl; bytes = ns * 2; esz = 2; }
        else if (std::strcmp(name, "slot_lp") == 0) { off = pl->off.slp; bytes = ns; esz = 1; }
        else if (std::strcmp(name, "tile_cut8") == 0) { off = pl->off.tc8; bytes = T * 9 * 2; esz = 2; }
        else if (std::strcmp(name, "tile_cut16") == 0) { off = pl->off.tc16; bytes = T * 17 * 2; esz = 2; }
        else if (pl->dev_wpt && std::strcmp(name, "slot_code") == 0) { off = pl->off.sc; bytes = ns * 2; esz = 2; }
        else if (pl->dev_wpt && std::strcmp(name, "tile_la") == 0) { off = pl->off.tla; bytes = T * kLanes; esz = 1; }
        else if (pl->dev_wpt && std::strcmp(name, "tile_rec") == 0) { off = pl->off.trec; bytes = T * 8 * 4; }
        else if (pl->dev_wpt && pl->em_ok && std::strcmp(name, "it_edge") == 0) { off = pl->off.ite; bytes = (size_t)pl->em_its * kLanes * 4; }
        else if (pl->dev_wpt && pl->em_ok && std::strcmp(name, "tile_sinfo") == 0) { off = pl->off.tsi; bytes = T * kLanes * 4; }
        if (bytes) {
            std::vector<int32_t> &v = pl->dev_readback;
            v.assign((bytes + 3) / 4, 0);
            if (hipMemcpy(v.data(), static_cast<const char *>(pl->dev_base) + off, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
            *data = v.data();
            return (int64_t)(bytes / esz);
        }
    }
    if (pl->dev_pm && pl->dev_base && (std::strcmp(name, "pm_edge") == 0 || std::strcmp(name, "pm_rec") == 0)) {
        // a plan whose pair-major table was written on the device: read back on request (tests, tooling)
        const bool edge = name[3] == 'e';
        std::vector<int32_t> &v = pl->dev_readback;
        v.assign(edge ? (size_t)pl->pm_rounds * kLanes : (size_t)pl->info.tiles * 4, 0);
        if (hipMemcpy(v.data(), static_cast<const char *>(pl->dev_base) + (edge ? pl->off.pme : pl->off.pmr), v.size() * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        *data = v.data();
        return (int64_t)v.size();
    }
    BT_ARR(kx) BT_ARR(trk_loc) BT_ARR(pair_i) BT_ARR(pair_j) BT_ARR(pp_ptr) BT_ARR(pp_idx) BT_ARR(sg_ptr)
    BT_ARR(tile_trk0) BT_ARR(tile_ntrk) BT_ARR(tile_ncam) BT_ARR(tile_cam0) BT_ARR(tile_slot0)
    BT_ARR(tile_nslot) BT_ARR(tile_erow0) BT_ARR(tile_cams) BT_ARR(slot_edge) BT_ARR(slot_pair)
    BT_ARR(slot_lab) BT_ARR(col_ptr) BT_ARR(row_idx) BT_ARR(upd_ptr) BT_ARR(upd) BT_ARR(blk_col) BT_ARR(upd_next) BT_ARR(perm) BT_ARR(blk_src) BT_ARR(lvl_ptr) BT_ARR(lvl_cols)
    BT_ARR(col_lvl) BT_ARR(dp_ptr) BT_ARR(dp) BT_ARR(tile_pair0) BT_ARR(tile_npair) BT_ARR(tile_pairs) BT_ARR(slot_lp) BT_ARR(tile_flags)
    BT_ARR(fz_pend_ptr) BT_ARR(fz_pend) BT_ARR(fz_lazy_ptr) BT_ARR(fz_lazy) BT_ARR(fz_yurg) BT_ARR(fz_meta) BT_ARR(fz_pmeta) BT_ARR(bs_sync) BT_ARR(fz_rowinfo) BT_ARR(fz_pfirst) BT_ARR(fz_psecond) BT_ARR(act_bits) BT_ARR(act_rank) BT_ARR(tile_ij) BT_ARR(tile_kx) BT_ARR(lvl_meta) BT_ARR(slot_code) BT_ARR(tile_la) BT_ARR(tile_rec) BT_ARR(it_edge) BT_ARR(tile_sinfo) BT_ARR(tile_cut8) BT_ARR(tile_cut16) BT_ARR(pm_edge) BT_ARR(pm_rec) BT_ARR(pm_lb) BT_ARR(pm_la)
#undef BT_ARR
    return -1;
}

int bt_ba_workspace_init(const bt_plan *pl, void *ws, void *stream) {
    if (!pl || !ws) return BT_EINVAL;
    return hipMemsetAsync(static_cast<char *>(ws) + pl->ws.sys, 0, pl->ws.zero_bytes, static_cast<hipStream_t>(stream)) == hipSuccess
               ? BT_OK : BT_EHIP;
}

int bt_ba_reduce(const bt_plan *pl, const bt_ba_args *a, void *ws, void *stream) {
    const int rc = check(pl, a, ws);
    if (rc != BT_OK) return rc;
    const StepArgs s = make_args(pl, a, ws);
    mark_launch(pl, stream);
    return launch_reduce(pl->dev, s, pl->ws.zero_bytes / sizeof(double), is_so(pl, a), static_cast<hipStream_t>(stream));
}

int bt_ba_solve_update(const bt_plan *pl, const bt_ba_args *a, void *ws, void *stream) {
    const int rc = check(pl, a, ws);
    if (rc != BT_OK) return rc;
    const bool so = is_so(pl, a);
    if (!so && !a->poses_out) return BT_EINVAL;
    if (!so && a->poses_out == a->poses) return BT_EINVAL;
    const StepArgs s = make_args(pl, a, ws);
    const bool copy_poses = so && a->poses_out && a->poses_out != a->poses;
    mark_launch(pl, stream);
    return launch_solve_update(pl->dev, s, so, copy_poses, static_cast<hipStream_t>(stream));
}

int bt_ba_step(const bt_plan *pl, const bt_ba_args *a, void *ws, void *stream) {
    const int rc = check(pl, a, ws);
    if (rc != BT_OK) return rc;
    const bool so = is_so(pl, a);
    if (!so && (!a->poses_out || a->poses_out == a->poses)) return BT_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    mark_launch(pl, stream);
    const StepArgs s = make_args(pl, a, ws);
    const bool copy_poses = so && a->poses_out && a->poses_out != a->poses;
    bool fused = false;              // (structure-only steps on the k_tile path are one launch)
    int r = launch_reduce(pl->dev, s, pl->ws.zero_bytes / sizeof(double), so, st, nullptr, nullptr, so ? (copy_poses ? 1 : 0) : -1, &fused);
    if (r == BT_OK && !fused) r = launch_solve_update(pl->dev, s, so, copy_poses, st);
    return r;
}

int bt_ba_step_timed(const bt_plan *pl, const bt_ba_args *a, void *ws, void *stream, float *ms) {
    const int rc = check(pl, a, ws);
    if (rc != BT_OK || !ms) return rc != BT_OK ? rc : BT_EINVAL;
    const bool so = is_so(pl, a);
    if (!so && (!a->poses_out || a->poses_out == a->poses)) return BT_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    mark_launch(pl, stream);
    hipEvent_t ev[12];
    for (auto &e : ev) if (hipEventCreate(&e) != hipSuccess) return BT_EHIP;
    const StepArgs s = make_args(pl, a, ws);
    const bool copy_poses = so && a->poses_out && a->poses_out != a->poses;
    unsigned ran = 0;
    int r = launch_reduce(pl->dev, s, pl->ws.zero_bytes / sizeof(double), so, st, ev, &ran);
    if (r == BT_OK) r = launch_solve_update(pl->dev, s, so, copy_poses, st, ev, &ran);
    if (r == BT_OK && hipStreamSynchronize(st) != hipSuccess) r = BT_EHIP;
    for (int k = 0; k < 6; ++k) {
        ms[k] = 0.0f;                                  // a kernel that was not launched (structure-only steps skip two) reads 0
        if (r == BT_OK && (ran >> k & 1u) && hipEventElapsedTime(&ms[k], ev[2 * k], ev[2 * k + 1]) != hipSuccess) ms[k] = -1.0f;
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return r;
}

int bt_ba_pack(const bt_plan *pl, const bt_ba_args *a, void *ws, void *stream) {
    const int rc = check(pl, a, ws);
    if (rc != BT_OK) return rc;
    if (is_so(pl, a)) return BT_OK;
    mark_launch(pl, stream);
    return launch_pack(pl->dev, make_args(pl, a, ws), false, static_cast<hipStream_t>(stream));
}

int bt_ba_unpack(const bt_plan *pl, const bt_ba_args *a, void *ws, void *stream) {
    const int rc = check(pl, a, ws);
    if (rc != BT_OK) return rc;
    if (is_so(pl, a)) return BT_OK;
    mark_launch(pl, stream);
    return launch_pack(pl->dev, make_args(pl, a, ws), true, static_cast<hipStream_t>(stream));
}

int bt_ba_reduce_pack(const bt_plan *pl, const bt_ba_args *a, void *ws, void *stream) {
    const int rc = bt_ba_reduce(pl, a, ws, stream);
    return rc != BT_OK ? rc : bt_ba_pack(pl, a, ws, stream);
}

int bt_ba_unpack_solve_update(const bt_plan *pl, const bt_ba_args *a, void *ws, void *stream) {
    const int rc = bt_ba_unpack(pl, a, ws, stream);
    return rc != BT_OK ? rc : bt_ba_solve_update(pl, a, ws, stream);
}

size_t bt_xchg_bytes(const bt_plan *pl, int world) {
    return (pl && pl->dev_base && world > 0 && world <= kMaxRanks) ? xchg_bytes(pl->dev, world) : 0;
}

int bt_xchg_alloc(size_t bytes, void **buf, unsigned char handle[64]) {
    if (!buf || !handle || bytes == 0) return BT_EINVAL;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    void *d = nullptr;
    // uncached: the peers' stores and this rank's loads meet in memory, not in caches that are not coherent across dies / devices
    if (hipExtMallocWithFlags(&d, bytes, hipDeviceMallocUncached) != hipSuccess) return BT_ENOMEM;
    hipIpcMemHandle_t h;
    if (hipMemset(d, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess || hipIpcGetMemHandle(&h, d) != hipSuccess) {
        (void)hipFree(d);
        return BT_EHIP;
    }
    std::memcpy(handle, &h, 64);
    *buf = d;
    return BT_OK;
}

int bt_config_wave_per_tile_kernels(int enable) { return bt::config_wave_per_tile_kernels(enable); }

int bt_xchg_open(const unsigned char handle[64], void **peer) {
    if (!handle || !peer) return BT_EINVAL;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, 64);
    return hipIpcOpenMemHandle(peer, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess ? BT_OK : BT_EHIP;
}

int bt_xchg_close(void *peer) { return (!peer || hipIpcCloseMemHandle(peer) == hipSuccess) ? BT_OK : BT_EHIP; }
int bt_xchg_free(void *buf) { return (!buf || hipFree(buf) == hipSuccess) ? BT_OK : BT_EHIP; }

int bt_ba_reduce_push(const bt_plan *pl, const bt_ba_args *a, void *ws, void *const *bufs, int world, int rank, int64_t epoch, void *stream) {
    // (validated BEFORE anything is enqueued: an EINVAL must not leave half a step on the stream)
    if (!bufs || world < 1 || world > kMaxRanks || rank < 0 || rank >= world || epoch < 1) return BT_EINVAL;
    for (int q = 0; q < world; ++q) if (!bufs[q]) return BT_EINVAL;
    const int rc = bt_ba_reduce(pl, a, ws, stream);
    if (rc != BT_OK || is_so(pl, a)) return rc;
    return launch_xchg_push(pl->dev, make_args(pl, a, ws), bufs, world, rank, epoch, static_cast<hipStream_t>(stream));
}

int bt_ba_pull_solve_update(const bt_plan *pl, const bt_ba_args *a, void *ws, void *own, int world, int64_t epoch, void *stream) {
    const int rc = check(pl, a, ws);
    if (rc != BT_OK) return rc;
    if (!is_so(pl, a)) {
        if (!own || world < 1 || world > kMaxRanks || epoch < 1) return BT_EINVAL;
        mark_launch(pl, stream);
        const int r2 = launch_xchg_pull(pl->dev, make_args(pl, a, ws), own, world, epoch, static_cast<hipStream_t>(stream));
        if (r2 != BT_OK) return r2;
    }
    return bt_ba_solve_update(pl, a, ws, stream);
}

int bt_ba_xchg_status(const bt_plan *pl, void *ws, void *stream, int32_t *status) {
    if (!pl || !ws || !status) return BT_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(status, static_cast<char *>(ws) + pl->ws.status + sizeof(int32_t), sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess)
        return BT_EHIP;
    return hipStreamSynchronize(st) == hipSuccess ? BT_OK : BT_EHIP;
}

double *bt_ba_packed(const bt_plan *pl, void *ws, int64_t *count) {
    if (!pl || !ws) return nullptr;
    if (count) *count = pl->info.nnz_blocks * 36 + 6 * pl->info.n;
    return reinterpret_cast<double *>(static_cast<char *>(ws) + pl->ws.packed);
}

double *bt_ba_system(const bt_plan *pl, void *ws, int64_t *count) {
    if (!pl || !ws) return nullptr;
    const int64_t D = 6 * pl->info.n;
    if (count) *count = D * D + D;
    return reinterpret_cast<double *>(static_cast<char *>(ws) + pl->ws.sys);
}

float *bt_ba_dx(const bt_plan *pl, void *ws) {
    return (pl && ws) ? reinterpret_cast<float *>(static_cast<char *>(ws) + pl->ws.dx) : nullptr;
}

int bt_ba_status(const bt_plan *pl, void *ws, void *stream, int32_t *status) {
    if (!pl || !ws || !status) return BT_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(status, static_cast<char *>(ws) + pl->ws.status, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess)
        return BT_EHIP;
    return hipStreamSynchronize(st) == hipSuccess ? BT_OK : BT_EHIP;
}

}  // extern "C"
