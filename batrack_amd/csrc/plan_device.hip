// plan_device.hip — the planner's passes over the EDGES on the device (the layouts of k_etile and k_tile: graphs below 2048 tiles).
//
// A plan from scratch cost ~1.4 ms of host analysis for the 138k-edge window (ba_plan.cpp), nearly all of it in three passes
// that touch every edge: the per-track figures (count, source frame, set of target frames), the grouping of the edges by
// track, and the walk that writes the one edge-sized table of such a plan (pm_edge).  The caller's sliding window changes its
// edge list every frame (batrack.py:189-212) and the list is a shifted copy of an earlier one only in steady state: while the
// window fills, every update() pays a plan from scratch.  Here the three passes run where the edge list already is:
//   k_plan_stats    one thread per edge: atomics into a per-patch table (count, source frame, 128-bit mask of target frames around
//                   the source frame) and the figures of the whole list (frames and patches named, self edges)
//   radix sort      of (patch - first patch) << bits | (target frame - first frame), some 18 bits, with the edge index as value
//                   (hipcub / rocPRIM, stable): a track's edges in (target frame, index) order — the order the host's stable
//                   counting passes produce.  Runs while the host lays out the small tables.
//   k_plan_rounds   the round of every edge = its rank among the edges of the same (track, pair), the tiles' round counts
//   k_plan_prefix   first round of every tile, the records' final words, the table's size
//   k_plan_fill     writes pm_edge into the uploaded plan
// The host keeps what is small: tracks, pairs, tiles, the reduced system's symbolic factorisation (ba_plan.cpp reads the
// per-patch table instead of the edges).  64-track layouts: k_plan_slots / k_plan_cuts instead of the last three — and, for the
// graphs of the wave-per-tile kernels (2048 tiles and more), k_plan_slots writes their compact tables as well and k_plan_sinfo
// decides whether the edge-major layout of k_edge2 applies (every tile slot-uniform).  A sharded plan runs the passes on the
// rank's segment of the sorted list.  Anything that does not fit — a track whose target frames are not within 64 of its source
// frame, two source frames for one track — falls back to the analysis on the edges.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "ba_plan.hpp"

namespace bt {

// glob: 0 n_all, 1 f_lo, 2 kmin, 3 kmax, 4 any_self, 5 two source frames for a track, 6 a target outside the mask, 7 tracks
// (no atomic here returns a value: 415k returning atomics on the per-patch records were 85 us of a 138k-edge list; and the list's
//  figures leave a workgroup as ONE record of partials, reduced by k_plan_count: four atomics per workgroup on the same four words
//  — 131k of them for 8.4M edges — are served one after the other, 25 ns each: 3.3 ms of such a plan)
constexpr int kStatBlocks = 1024;                 // workgroups of k_plan_stats (grid-stride), = records of partials
__global__ __launch_bounds__(256) void k_plan_stats(const unsigned long long *words, long long E, PatchStat *stat, int *part, int *vals) {
    __shared__ int s_max_f, s_min_f, s_kmin, s_kmax, s_flags;
    if (threadIdx.x == 0) { s_max_f = 0; s_min_f = 0x7fffffff; s_kmin = 0x7fffffff; s_kmax = -1; s_flags = 0; }
    __syncthreads();
    int max_f = 0, min_f = 0x7fffffff, kmin = 0x7fffffff, kmax = -1, fl = 0;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long long)gridDim.x * blockDim.x) {
        const unsigned long long w = words[e];
        const int k = (int)(w >> 32), i = (int)((w >> 16) & 0xffff), j = (int)(w & 0xffff);
        vals[e] = (int)e;
        if (i == j) fl |= 1;
        PatchStat *t = stat + k;
        atomicAdd(&t->cnt, 1);
        atomicMax(&t->src, i);                      // (one source frame per track is checked below: min == max)
        atomicMin(&t->src_min, i);
        const int bit = j - (i - 64);
        if (bit < 0 || bit >= 128) fl |= 4;
        else atomicOr(bit < 64 ? &t->mask : &t->mask2, 1ull << (bit & 63));
        max_f = max(max_f, max(i, j) + 1); min_f = min(min_f, min(i, j));
        kmin = min(kmin, k); kmax = max(kmax, k);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        max_f = max(max_f, __shfl_xor(max_f, m)); min_f = min(min_f, __shfl_xor(min_f, m));
        kmin = min(kmin, __shfl_xor(kmin, m)); kmax = max(kmax, __shfl_xor(kmax, m)); fl |= __shfl_xor(fl, m);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&s_max_f, max_f); atomicMin(&s_min_f, min_f); atomicMin(&s_kmin, kmin); atomicMax(&s_kmax, kmax);
        if (fl) atomicOr(&s_flags, fl);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int *o = part + 8 * blockIdx.x;
        o[0] = s_max_f; o[1] = s_min_f; o[2] = s_kmin; o[3] = s_kmax; o[4] = s_flags;
    }
}

// tracks (patches with an edge) and the one-source-frame check over the window's slice of the table.  Every workgroup first
// reduces the partials k_plan_stats left (shuffles, no atomics: a kernel of its own for that was another launch and ~2k
// same-address LDS atomics); workgroup 0 writes the list's figures to glob[0..6].
// (own_lo: a rank's plan also counts the tracks and edges of the patches in front of its range — glob[13], glob[14])
__global__ __launch_bounds__(256) void k_plan_count(const PatchStat *stat, const int *part, int nblk, int *glob, int own_lo) {
    __shared__ int red[4][5];
    int max_f = 0, min_f = 0x7fffffff, kmin = 0x7fffffff, kmax = -1, fl = 0;
    for (int r = threadIdx.x; r < nblk; r += blockDim.x) {
        const int *o = part + 8 * r;
        max_f = max(max_f, o[0]); min_f = min(min_f, o[1]); kmin = min(kmin, o[2]); kmax = max(kmax, o[3]); fl |= o[4];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        max_f = max(max_f, __shfl_xor(max_f, m)); min_f = min(min_f, __shfl_xor(min_f, m));
        kmin = min(kmin, __shfl_xor(kmin, m)); kmax = max(kmax, __shfl_xor(kmax, m)); fl |= __shfl_xor(fl, m);
    }
    if ((threadIdx.x & 63) == 0) { int *o = red[threadIdx.x >> 6]; o[0] = max_f; o[1] = min_f; o[2] = kmin; o[3] = kmax; o[4] = fl; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) { max_f = max(max_f, red[w][0]); min_f = min(min_f, red[w][1]); kmin = min(kmin, red[w][2]); kmax = max(kmax, red[w][3]); fl |= red[w][4]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        glob[0] = max_f; glob[1] = min_f; glob[2] = kmin; glob[3] = kmax;
        if (fl & 1) glob[4] = 1;
        if (fl & 4) glob[6] = 1;
    }
    int n = 0, bad = 0, nb = 0, eb = 0;
    for (int p = kmin + (int)(blockIdx.x * blockDim.x + threadIdx.x); p <= kmax; p += (int)(gridDim.x * blockDim.x)) {
        const PatchStat t = stat[p];
        if (t.cnt > 0) { ++n; bad |= t.src != t.src_min; if (p < own_lo) { ++nb; eb += t.cnt; } }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { n += __shfl_xor(n, m); bad |= __shfl_xor(bad, m); nb += __shfl_xor(nb, m); eb += __shfl_xor(eb, m); }
    if ((threadIdx.x & 63) == 0) { if (n) atomicAdd(&glob[7], n); if (bad) glob[5] = 1; if (nb) { atomicAdd(&glob[13], nb); atomicAdd(&glob[14], eb); } }
}

// The block pattern of the reduced system over ALL tracks of the list, for a rank's plan of a sharded solve: the free cameras of a
// track (its source frame and its targets) couple pairwise (the track's Schur term, ba.py:321; the pairs' B blocks are among them).
// bits [u][v], v <= u, n x W words; every workgroup ORs into a bitmap in LDS first.
__global__ __launch_bounds__(256) void k_plan_pattern(const PatchStat *stat, int kmin, int kmax, int fixedp, int n, int W, unsigned *out) {
    extern __shared__ unsigned bm[];
    for (int i = threadIdx.x; i < n * W; i += blockDim.x) bm[i] = 0u;
    __syncthreads();
    for (int p = kmin + (int)(blockIdx.x * blockDim.x + threadIdx.x); p <= kmax; p += (int)(gridDim.x * blockDim.x)) {
        const PatchStat t = stat[p];
        if (t.cnt <= 0) continue;
        if (p > kmin) { const PatchStat q = stat[p - 1]; if (q.cnt > 0 && q.src == t.src && q.mask == t.mask && q.mask2 == t.mask2) continue; }   // (as the patch before)
        // the track's cameras as bits of camera numbers: camera of mask bit b = off + b, off = src - 64 - fixedp; the source is bit 64
        const int off = t.src - 64 - fixedp;
        unsigned long long m0 = t.mask, m1 = t.mask2 | 1ull;
        unsigned cw[8] = {0, 0, 0, 0, 0, 0, 0, 0};                       // n <= 255: eight words
        for (int half = 0; half < 2; ++half)
            for (unsigned long long mk = half ? m1 : m0; mk; mk &= mk - 1) {
                const int c = off + 64 * half + __builtin_ctzll(mk);
                if (c >= 0 && c < n) cw[c >> 5] |= 1u << (c & 31);
            }
        for (int w = 0; w < W; ++w)
            for (unsigned um = cw[w]; um; um &= um - 1) {
                const int u = 32 * w + __builtin_ctz(um);
                for (int v = 0; v <= w; ++v) {
                    const unsigned val = v < w ? cw[v] : (cw[v] & (0xffffffffu >> (31 - (u & 31))));       // columns <= u
                    if (val && (bm[u * W + v] & val) != val) atomicOr(&bm[u * W + v], val);
                }
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n * W; i += blockDim.x) if (bm[i]) atomicOr(&out[i], bm[i]);
}

__global__ __launch_bounds__(256) void k_plan_stat_clear(PatchStat *stat, RepStat *rstat, long long rstat_n, long long lo, long long hi) {
    const long long p = lo + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p <= hi) {
        stat[p].cnt = 0; stat[p].src = -1; stat[p].src_min = 0x7fffffff; stat[p].mask = 0ull; stat[p].mask2 = 0ull;
        if (p < rstat_n) { rstat[p].rmask = 0ull; rstat[p].rmask2 = 0ull; }      // (the two tables are dirty over the same range)
    }
}

// The sort key of an edge: (patch - kmin) << jbits | (target frame - f_lo) — some 18 bits for a window instead of the 50 of
// the packed word (a track has ONE source frame, so the word's middle field orders nothing): three radix passes, not seven.
__global__ __launch_bounds__(256) void k_plan_keys(const unsigned long long *words, long long E, int kmin, int f_lo, int jbits, unsigned *keys) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const unsigned long long w = words[e];
    keys[e] = ((unsigned)((int)(w >> 32) - kmin) << jbits) | (unsigned)((int)(w & 0xffff) - f_lo);
}

// The targets a track observes MORE THAN ONCE (RepStat), from the sorted keys: equal keys are neighbours.  For the lists large
// enough for the wave-per-tile kernels only (their aligned slot layout, ba_plan.cpp).  (A returning atomicOr in k_plan_stats gives
// the same masks without the sort: 2.1 ms instead of 0.3 for 8.4M edges — returning atomics on the per-patch records again.)
__global__ __launch_bounds__(256) void k_plan_rep(const unsigned *keys, long long E, int jbits, int kmin, int f_lo, const PatchStat *stat, RepStat *rstat, int *flag) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q <= 0 || q >= E) return;
    const unsigned key = keys[q];
    if (keys[q - 1] != key) return;
    const int k = (int)(key >> jbits) + kmin, j = (int)(key & ((1u << jbits) - 1u)) + f_lo;
    const int bit = j - (stat[k].src - 64);
    if (bit < 0 || bit >= 128) return;
    atomicOr(bit < 64 ? &rstat[k].rmask : &rstat[k].rmask2, 1ull << (bit & 63));
    *flag = 1;
}

struct PlanFillArgs {
    const unsigned *keys;                  // the sorted keys
    const int *vals;                       // their edge indices
    const unsigned long long *words;       // the packed edge list (caller's order)
    int jbits;
    long long E;
    const int *trk_win;                    // patch - kmin -> track
    const int *trk_loc;                    // track -> tile << 6 | lane
    const int *pair_of; int f_lo, nw;      // (i - f_lo) * nw + (j - f_lo) -> pair
    const int *tile_pair0, *tile_npair, *tile_pairs;
    int *rec;                              // pm_rec [tiles][4]
    unsigned char *dcode;                  // [E]: the round of every sorted edge
    int *dmax;                             // [tiles]
    int *out;                              // 0: rounds of the whole table, 1: some (track, pair) has more than 255 edges
    int *pm_edge;
};

__global__ __launch_bounds__(256) void k_plan_rounds(PlanFillArgs a) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = q < a.E;                              // (the grid's last wave is partial: its idle lanes take part in the votes below)
    if ((long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63) >= a.E) return;
    const unsigned key = valid ? a.keys[q] : 0u;
    int d = 0;
    if (valid) {
        while (d < 256 && q - d - 1 >= 0 && a.keys[q - d - 1] == key) ++d;
        if (d > 255) { a.out[1] = 1; d = 255; }
        a.dcode[q] = (unsigned char)d;
    }
    // one atomic per wave where the wave's edges are all of one tile — the sorted order keeps a tile's ~900 edges together
    // (138k atomics on the tiles' 160 counters were 0.3 ms of this kernel)
    const int t = valid ? a.trk_loc[a.trk_win[key >> a.jbits]] >> 6 : -1;
    const int t0 = __builtin_amdgcn_readfirstlane(t);           // (lane 0 of a wave that got here is valid)
    if (__all(!valid || t == t0)) {
        int mx = valid ? d + 1 : 0;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) mx = max(mx, __shfl_xor(mx, m));
        if ((threadIdx.x & 63) == 0) atomicMax(&a.dmax[t0], mx);
    } else if (valid && (q + 1 == a.E || a.keys[q + 1] != key)) atomicMax(&a.dmax[t], d + 1);
}

// first round of every tile (exclusive prefix of iterations x rounds), the records' final words, the table's rounds
__global__ __launch_bounds__(256) void k_plan_prefix(int *rec, const int *dmax, int T, int *out) {
    __shared__ long long part[256];
    const int tid = threadIdx.x, per = (T + 255) / 256, t0 = tid * per, t1 = min(T, t0 + per);
    long long sum = 0;
    for (int t = t0; t < t1; ++t) sum += (long long)rec[4 * t + 2] * max(dmax[t], 1);
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const long long v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    long long acc = tid ? part[tid - 1] : 0;
    for (int t = t0; t < t1; ++t) {
        const int D = max(dmax[t], 1);
        rec[4 * t] = (int)acc;
        rec[4 * t + 1] = (rec[4 * t + 1] & 0xff) | (D << 8);
        acc += (long long)rec[4 * t + 2] * D;
    }
    if (tid == 255) out[0] = part[255] > 0x7fffffffll / 64 ? -1 : (int)part[255];
}

__global__ __launch_bounds__(256) void k_plan_fill(PlanFillArgs a) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.E) return;
    const unsigned key = a.keys[q];
    const int e = a.vals[q];
    const unsigned long long w = a.words[e];
    const int i = (int)((w >> 16) & 0xffff), j = (int)(w & 0xffff);
    const int loc = a.trk_loc[a.trk_win[key >> a.jbits]], t = loc >> 6, l = loc & 63;
    const int gp = a.pair_of[(i - a.f_lo) * a.nw + (j - a.f_lo)];
    const int *tp = a.tile_pairs + a.tile_pair0[t];
    int lo = 0, hi = a.tile_npair[t] - 1;                       // (the tile's pairs are ascending)
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (tp[mid] < gp) lo = mid + 1; else hi = mid; }
    const int round0 = a.rec[4 * t], lg = a.rec[4 * t + 1] & 0xff, D = a.rec[4 * t + 1] >> 8, G = 64 >> lg;
    a.pm_edge[((size_t)round0 + (size_t)(l / G) * D + a.dcode[q]) * 64 + (size_t)((l % G) << lg) + (size_t)lo] = e;
}

// ---- 64-track layouts (k_tile): the [slots][64] arrays and the waves' slot cuts (ba_plan.cpp describes them)
struct PlanSlotArgs {
    const unsigned *keys; const int *vals; const unsigned long long *words; int jbits; long long E;
    const int *trk_win, *trk_loc, *pair_of, *off; int f_lo, nw, fixedp;
    const int *pbase;                       // aligned slot layout (ba_plan.cpp): first slot of every (tile, local pair), as tile_pairs; null: slot = the edge's position in its track
    const int *tile_slot0, *tile_cam0, *tile_ncam, *tile_cams, *tile_pair0, *tile_npair, *tile_pairs, *tile_nslot;
    int *slot_edge, *slot_pair; unsigned short *slot_lab; unsigned char *slot_lp;
    unsigned char *crossed;                 // [slots + 1], by global slot: some track's run of one target camera spans slots s - 1 and s
    unsigned short *cut8, *cut16; int T;
    // tables of the wave-per-tile kernels (ba_plan.cpp: slot_code, tile_la, tile_rec, it_edge, tile_sinfo), null: not wanted
    unsigned short *slot_code; unsigned char *tile_la; int *tile_rec; int *it_edge; unsigned *tile_sinfo; int *em_bad;
    const int *tile_ntrk;
};

__device__ __forceinline__ int local_cam(const int *cams, int nc, int c) {      // position of free camera c in the tile's ascending list, 0xff: fixed
    if (c < 0) return 0xff;
    int lo = 0, hi = nc - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cams[mid] < c) lo = mid + 1; else hi = mid; }
    return lo;
}

__global__ __launch_bounds__(256) void k_plan_slots(PlanSlotArgs a) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.E) return;
    const unsigned key = a.keys[q];
    const int e = a.vals[q];
    const unsigned long long w = a.words[e];
    const int i = (int)((w >> 16) & 0xffff), j = (int)(w & 0xffff);
    const int trk = a.trk_win[key >> a.jbits], loc = a.trk_loc[trk], t = loc >> 6, l = loc & 63;
    const bool first = q == a.off[trk];                         // the track's first edge
    const int *cams = a.tile_cams + a.tile_cam0[t];
    const int nc = a.tile_ncam[t];
    const int la = local_cam(cams, nc, i - a.fixedp), lb = local_cam(cams, nc, j - a.fixedp);
    const int gp = a.pair_of[(i - a.f_lo) * a.nw + (j - a.f_lo)];
    const int *tp = a.tile_pairs + a.tile_pair0[t];
    int lo = 0, hi = a.tile_npair[t] - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (tp[mid] < gp) lo = mid + 1; else hi = mid; }
    int s = (int)(q - a.off[trk]);
    if (a.pbase) {
        // aligned: the pair's first slot + the edge's rank among the track's edges with that pair (equal keys are neighbours)
        int r = 0;
        while (r < 64 && q - r - 1 >= 0 && a.keys[q - r - 1] == key) ++r;
        s = a.pbase[a.tile_pair0[t] + lo] + r;
    }
    const size_t idx = ((size_t)a.tile_slot0[t] + (size_t)s) * 64 + (size_t)l;
    a.slot_edge[idx] = e; a.slot_pair[idx] = gp; a.slot_lab[idx] = (unsigned short)(la | (lb << 8)); a.slot_lp[idx] = (unsigned char)lo;
    // the previous edge of the track has the same target frame (its key is the same): a run that continues into this slot
    const bool run = !first && lb != 0xff && a.keys[q - 1] == key;
    if (run) a.crossed[a.tile_slot0[t] + s] = 1;
    if (a.slot_code) {
        a.slot_code[idx] = (unsigned short)(lb | (lo << 8));
        if (first) a.tile_la[(size_t)t * 64 + (size_t)l] = (unsigned char)la;
        // (flag bit 2 of the tile's record: a run spans the boundary of the two half-chunks of slots of k_stream's two waves)
        const int ns = a.tile_nslot[t], ch = (ns + 1) >> 1;
        if (run && s == ch && ch < ns) atomicOr(&a.tile_rec[(size_t)t * 8], 4 << 24);
        if (a.it_edge) {
            const int it0 = a.tile_rec[(size_t)t * 8 + 6], lg = a.tile_rec[(size_t)t * 8 + 7] & 0xff, G = 64 >> lg;
            a.it_edge[((size_t)it0 + (size_t)(l / G)) * 64 + (size_t)((l % G) << lg) + (size_t)s] = e;
        }
    }
}

// one wave per tile: is every slot of the tile the same (target camera, pair) for all of its tracks?  tile_sinfo as ba_plan.cpp
// describes it (lane = slot: at most 64 slots per tile where this layout is considered at all)
__global__ __launch_bounds__(256) void k_plan_sinfo(PlanSlotArgs a) {
    const int t = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (t >= a.T) return;
    const int ns = a.tile_nslot[t], nt = a.tile_ntrk[t];
    const size_t b0 = (size_t)a.tile_slot0[t] * 64;
    unsigned mine = 0;
    bool bad = false;
    for (int sl = 0; sl < ns; ++sl) {
        const size_t i = b0 + (size_t)sl * 64 + (size_t)lane;
        const int c = (lane < nt && a.slot_edge[i] >= 0) ? (int)a.slot_code[i] : -1;
        const unsigned long long valid = __ballot(c >= 0);
        if (!valid) continue;
        const int first = __shfl(c, __ffsll((long long)valid) - 1);
        if (__any(c >= 0 && c != first)) bad = true;
        if (lane == sl) mine = (unsigned)first | (1u << 17);
    }
    const unsigned up = __shfl_down(mine, 1), dn = __shfl_up(mine, 1);
    const bool used = (mine >> 17) & 1u;
    const bool rep = used && ((lane + 1 < ns && ((up >> 17) & 1u) && ((up >> 8) & 0xffu) == ((mine >> 8) & 0xffu)) ||
                              (lane > 0 && ((dn >> 17) & 1u) && ((dn >> 8) & 0xffu) == ((mine >> 8) & 0xffu)));
    a.tile_sinfo[(size_t)t * 64 + (size_t)lane] = lane < ns ? (mine | (rep ? 1u << 16 : 0u)) : 0u;
    if (bad && lane == 0) *a.em_bad = 1;
}

// one thread per (tile, 8 or 16 waves): the nearest slot boundary to the even split that no run crosses (ba_plan.cpp)
__global__ __launch_bounds__(256) void k_plan_cuts(PlanSlotArgs a) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x, t = id >> 1, W = (id & 1) ? 16 : 8;
    if (t >= a.T) return;
    const int ns = a.tile_nslot[t];
    const unsigned char *crossed = a.crossed + a.tile_slot0[t];
    unsigned short *cut = W == 8 ? a.cut8 + (size_t)t * 9 : a.cut16 + (size_t)t * 17;
    const int chunk = (ns + W - 1) / W;
    int prev = 0;
    cut[0] = 0;
    for (int w = 1; w < W; ++w) {
        const int ideal = min(ns, w * chunk);
        int best = ideal;
        if (ideal < ns && crossed[ideal]) {
            for (int d = 1; d <= chunk; ++d) {
                if (ideal - d >= prev && !crossed[ideal - d]) { best = ideal - d; break; }
                if (ideal + d <= ns && (ideal + d == ns || !crossed[ideal + d])) { best = ideal + d; break; }
            }
        }
        best = max(best, prev);
        cut[w] = (unsigned short)best;
        prev = best;
    }
    cut[W] = (unsigned short)ns;
}

// ------------------------------------------------------------------ host side of the passes
namespace {
struct DevPlanBuffers {
    PatchStat *stat = nullptr; size_t stat_cap = 0; long long dirty_lo = 0, dirty_hi = -1;
    RepStat *rstat = nullptr; size_t rstat_cap = 0; RepStat *h_rtab = nullptr; size_t rtab_cap = 0;   // the tracks' repeated targets (large lists only)
    unsigned *pat = nullptr, *h_pat = nullptr;                    // the coupling pattern of a sharded solve: 255 x 8 words (device, pinned host)
    int *glob = nullptr, *h_glob = nullptr;                       // 8 + 2 ints (device, pinned host)
    int *part = nullptr;                                          // [kStatBlocks][8]: the workgroups' partials of k_plan_stats
    unsigned *keys_in = nullptr, *keys = nullptr; int *vals_in = nullptr, *vals = nullptr; unsigned char *dcode = nullptr; size_t e_cap = 0;
    const unsigned long long *words = nullptr; int jbits = 0;
    void *temp = nullptr; size_t temp_cap = 0;
    PatchStat *h_tab = nullptr; size_t tab_cap = 0;              // pinned: the window's slice of the table
    unsigned char *crossed = nullptr; size_t crossed_cap = 0;
    int *small = nullptr, *h_small = nullptr; size_t small_cap = 0;   // trk_win | trk_loc | pair_of | rec | dmax (device, pinned staging)
    bool grow_edges(size_t E) {
        if (E <= e_cap) return true;
        (void)hipFree(keys_in); (void)hipFree(keys); (void)hipFree(vals_in); (void)hipFree(vals); (void)hipFree(dcode);
        const size_t want = E + E / 4 + 4096;
        if (hipMalloc(reinterpret_cast<void **>(&keys_in), want * 4) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&keys), want * 4) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&vals_in), want * 4) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&vals), want * 4) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&dcode), want) != hipSuccess) { e_cap = 0; return false; }
        e_cap = want;
        return true;
    }
};
DevPlanBuffers &bufs() { static thread_local DevPlanBuffers b; return b; }
}  // namespace

// Pass 1: the per-patch table and the list's figures.  On BT_OK *st describes the table slice (pinned host memory, valid until
// the thread's next call) and the sort of the words has been queued behind it on `stream`.  BT_NEED_EDGES: not a list this
// path takes.
int plan_device_stats(const uint64_t *d_words, int64_t E, int64_t p_tot, int64_t fixedp, int64_t own_lo, int64_t own_hi, void *stream, DevPlanStats *st, int64_t *tracks) {
    hipStream_t cs = static_cast<hipStream_t>(stream);
    DevPlanBuffers &b = bufs();
    if (!b.glob) {
        if (hipMalloc(reinterpret_cast<void **>(&b.glob), 16 * sizeof(int)) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&b.part), kStatBlocks * 8 * sizeof(int)) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void **>(&b.h_glob), 16 * sizeof(int), hipHostMallocDefault) != hipSuccess) return BT_ENOMEM;
    }
    if (p_tot > ((int64_t)8 << 20) || E > (int64_t)0x7fffffff / 2) return BT_NEED_EDGES;      // (a per-patch table of 32 B per slot of the buffer: 8 MB for the reference's 1024 x 256 slots)
    if ((size_t)p_tot > b.stat_cap) {
        (void)hipFree(b.stat);
        if (hipMalloc(reinterpret_cast<void **>(&b.stat), (size_t)p_tot * sizeof(PatchStat)) != hipSuccess) { b.stat_cap = 0; return BT_ENOMEM; }
        b.stat_cap = (size_t)p_tot; b.dirty_lo = 0; b.dirty_hi = p_tot - 1;
    }
    if (!b.grow_edges((size_t)E)) return BT_ENOMEM;
    // lists that can reach the tile count of the wave-per-tile kernels (64 tracks a tile): the repeated targets as well
    const bool rep = E >= (int64_t)edge_min_tiles() * kLanes && edge_min_tiles() < (1 << 29);
    if (rep && (size_t)p_tot > b.rstat_cap) {
        (void)hipFree(b.rstat);
        if (hipMalloc(reinterpret_cast<void **>(&b.rstat), (size_t)p_tot * sizeof(RepStat)) != hipSuccess) { b.rstat_cap = 0; return BT_ENOMEM; }
        b.rstat_cap = (size_t)p_tot; b.dirty_lo = 0; b.dirty_hi = p_tot - 1;
    }
    if (b.dirty_hi >= b.dirty_lo)
        hipLaunchKernelGGL(k_plan_stat_clear, dim3((unsigned)((b.dirty_hi - b.dirty_lo + 256) / 256)), dim3(256), 0, cs, b.stat,
                           b.rstat, (long long)b.rstat_cap, b.dirty_lo, b.dirty_hi);
    b.h_glob[0] = 0; b.h_glob[1] = 0x7fffffff; b.h_glob[2] = 0x7fffffff; b.h_glob[3] = -1;
    for (int c = 4; c < 16; ++c) b.h_glob[c] = 0;
    if (hipMemcpyAsync(b.glob, b.h_glob, 16 * sizeof(int), hipMemcpyHostToDevice, cs) != hipSuccess) return BT_EHIP;
    const int nblk = (int)std::min<int64_t>(kStatBlocks, (E + 255) / 256);
    hipLaunchKernelGGL(k_plan_stats, dim3((unsigned)nblk), dim3(256), 0, cs, reinterpret_cast<const unsigned long long *>(d_words),
                       (long long)E, b.stat, b.part, b.vals_in);
    const bool ranked = own_hi > 0 && !(own_lo <= 0 && own_hi >= p_tot);      // a rank's plan of a sharded solve
    hipLaunchKernelGGL(k_plan_count, dim3(64), dim3(256), 0, cs, b.stat, b.part, nblk, b.glob, ranked ? (int)std::min<int64_t>(own_lo, 0x7fffffff) : 0);
    if (hipMemcpyAsync(b.h_glob, b.glob, 8 * sizeof(int), hipMemcpyDeviceToHost, cs) != hipSuccess ||
        hipMemcpyAsync(b.h_glob + 13, b.glob + 13, 2 * sizeof(int), hipMemcpyDeviceToHost, cs) != hipSuccess || hipStreamSynchronize(cs) != hipSuccess) return BT_EHIP;
    const int *g = b.h_glob;
    b.dirty_lo = g[2]; b.dirty_hi = g[3];
    *tracks = g[7];
    if (g[5] || g[6] || g[3] < g[2]) return BT_NEED_EDGES;
    // (not a patch range mostly empty: the table's slice would be a larger copy than the edges)
    const int64_t t64 = ((int64_t)g[7] + kLanes - 1) / kLanes;
    if (t64 <= 0 || (size_t)(g[3] - g[2] + 1) > 4 * (size_t)g[7] + 65536) return BT_NEED_EDGES;
    // the slice of the table the host reads: all the patches the list names, or the rank's own
    const int64_t t_lo = ranked ? std::max<int64_t>(g[2], own_lo) : g[2], t_hi = ranked ? std::min<int64_t>(g[3], own_hi - 1) : g[3];
    const size_t nt = t_hi >= t_lo ? (size_t)(t_hi - t_lo + 1) : 0;
    if (nt == 0) return BT_NEED_EDGES;                             // (a rank without tracks: the host's analysis, as before)
    if (nt > b.tab_cap) {
        (void)hipHostFree(b.h_tab);
        if (hipHostMalloc(reinterpret_cast<void **>(&b.h_tab), (nt + nt / 4 + 1024) * sizeof(PatchStat), hipHostMallocDefault) != hipSuccess) { b.tab_cap = 0; return BT_ENOMEM; }
        b.tab_cap = nt + nt / 4 + 1024;
    }
    // a rank's plan: the coupling pattern of the whole list, reduced here instead of read record by record on the host
    const int n_free = (int)std::max<int64_t>(0, (int64_t)g[0] - fixedp), pat_w = (n_free + 31) / 32;
    const bool want_pat = ranked && n_free > 0 && n_free <= kMaxFree;
    if (want_pat) {
        if (!b.pat && (hipMalloc(reinterpret_cast<void **>(&b.pat), 256 * 8 * sizeof(unsigned)) != hipSuccess ||
                       hipHostMalloc(reinterpret_cast<void **>(&b.h_pat), 256 * 8 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess)) return BT_ENOMEM;
        if (hipMemsetAsync(b.pat, 0, (size_t)n_free * pat_w * sizeof(unsigned), cs) != hipSuccess) return BT_EHIP;
        hipLaunchKernelGGL(k_plan_pattern, dim3(64), dim3(256), (size_t)n_free * pat_w * sizeof(unsigned), cs, b.stat, g[2], g[3], (int)fixedp, n_free, pat_w, b.pat);
        if (hipMemcpyAsync(b.h_pat, b.pat, (size_t)n_free * pat_w * sizeof(unsigned), hipMemcpyDeviceToHost, cs) != hipSuccess) return BT_EHIP;
    }
    hipEvent_t ev = nullptr;
    if (hipMemcpyAsync(b.h_tab, b.stat + t_lo, nt * sizeof(PatchStat), hipMemcpyDeviceToHost, cs) != hipSuccess ||
        hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return BT_EHIP;
    if (hipEventRecord(ev, cs) != hipSuccess) { (void)hipEventDestroy(ev); return BT_EHIP; }
    // the sort runs while the host lays out tracks, pairs and tiles
    int jbits = 1, kbits = 1;
    while ((1 << jbits) < g[0] - g[1]) ++jbits;
    while (kbits < 31 && ((int64_t)1 << kbits) < (int64_t)g[3] - g[2] + 1) ++kbits;
    if (jbits + kbits > 32) { (void)hipEventDestroy(ev); return BT_NEED_EDGES; }
    b.words = reinterpret_cast<const unsigned long long *>(d_words); b.jbits = jbits;
    hipLaunchKernelGGL(k_plan_keys, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, cs, b.words, (long long)E, g[2], g[1], jbits, b.keys_in);
    size_t need = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, need, b.keys_in, b.keys, b.vals_in, b.vals, (int)E, 0, jbits + kbits, cs);
    if (need > b.temp_cap) {
        (void)hipFree(b.temp);
        if (hipMalloc(&b.temp, need + need / 4 + 4096) != hipSuccess) { b.temp_cap = 0; (void)hipEventDestroy(ev); return BT_ENOMEM; }
        b.temp_cap = need + need / 4 + 4096;
    }
    size_t tb = b.temp_cap;
    const bool sorted_ok = hipcub::DeviceRadixSort::SortPairs(b.temp, tb, b.keys_in, b.keys, b.vals_in, b.vals, (int)E, 0, jbits + kbits, cs) == hipSuccess;
    bool any_rep = false;
    if (rep && sorted_ok) {
        // the repeated targets: behind the sort (the host's analysis needs them from its first pass: this waits for the sort)
        hipLaunchKernelGGL(k_plan_rep, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, cs, b.keys, (long long)E, jbits, g[2], g[1], b.stat, b.rstat, b.glob + 8);
        if (hipMemcpyAsync(b.h_glob + 8, b.glob + 8, sizeof(int), hipMemcpyDeviceToHost, cs) != hipSuccess || hipStreamSynchronize(cs) != hipSuccess) { (void)hipEventDestroy(ev); return BT_EHIP; }
        any_rep = b.h_glob[8] != 0;
        if (any_rep) {
            if (nt > b.rtab_cap) {
                (void)hipHostFree(b.h_rtab);
                if (hipHostMalloc(reinterpret_cast<void **>(&b.h_rtab), (nt + nt / 4 + 1024) * sizeof(RepStat), hipHostMallocDefault) != hipSuccess) { b.rtab_cap = 0; (void)hipEventDestroy(ev); return BT_ENOMEM; }
                b.rtab_cap = nt + nt / 4 + 1024;
            }
            if (hipMemcpyAsync(b.h_rtab, b.rstat + t_lo, nt * sizeof(RepStat), hipMemcpyDeviceToHost, cs) != hipSuccess || hipStreamSynchronize(cs) != hipSuccess) { (void)hipEventDestroy(ev); return BT_EHIP; }
        }
    }
    const bool waited = hipEventSynchronize(ev) == hipSuccess;
    (void)hipEventDestroy(ev);
    if (!sorted_ok || !waited) return BT_EHIP;
    st->tab = b.h_tab; st->tab_lo = t_lo; st->tab_n = (int64_t)nt; st->kmin = g[2]; st->kmax = g[3]; st->n_all = g[0]; st->f_lo = g[1]; st->any_self = g[4];
    st->sliced = ranked ? 1 : 0; st->trk_before = g[13]; st->edges_before = g[14];
    st->pattern = want_pat ? b.h_pat : nullptr; st->pattern_words = pat_w;
    st->rep_known = rep ? 1 : 0; st->rtab = any_rep ? b.h_rtab : nullptr;
    return BT_OK;
}

// Pass 2 (after the host analysis): the rounds.  Leaves the final tile records and the staged tables on the device for
// plan_device_fill; *rounds = the table's rounds.  BT_NEED_EDGES if some (track, pair) has more than 255 edges.
int plan_device_rounds(const bt_plan *pl, int64_t E, void *stream, int64_t *rounds) {
    hipStream_t cs = static_cast<hipStream_t>(stream);
    DevPlanBuffers &b = bufs();
    const size_t T = (size_t)pl->info.tiles, nwin = pl->trk_win.size(), m = pl->trk_loc.size(), npo = pl->dev_pair_of.size();
    const size_t n_small = nwin + m + npo + 4 * T + T + 2;
    if (n_small > b.small_cap) {
        (void)hipFree(b.small); (void)hipHostFree(b.h_small);
        const size_t want = n_small + n_small / 4 + 4096;
        if (hipMalloc(reinterpret_cast<void **>(&b.small), want * sizeof(int)) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void **>(&b.h_small), want * sizeof(int), hipHostMallocDefault) != hipSuccess) { b.small_cap = 0; return BT_ENOMEM; }
        b.small_cap = want;
    }
    int *h = b.h_small;
    std::copy(pl->trk_win.begin(), pl->trk_win.end(), h);
    std::copy(pl->trk_loc.begin(), pl->trk_loc.end(), h + nwin);
    std::copy(pl->dev_pair_of.begin(), pl->dev_pair_of.end(), h + nwin + m);
    std::copy(pl->pm_rec.begin(), pl->pm_rec.end(), h + nwin + m + npo);
    std::fill(h + nwin + m + npo + 4 * T, h + n_small, 0);
    if (hipMemcpyAsync(b.small, h, n_small * sizeof(int), hipMemcpyHostToDevice, cs) != hipSuccess) return BT_EHIP;
    PlanFillArgs a{};
    a.keys = b.keys + pl->dev_q0; a.vals = b.vals + pl->dev_q0; a.words = b.words; a.jbits = b.jbits; a.E = E;
    a.trk_win = b.small; a.trk_loc = b.small + nwin;
    a.pair_of = b.small + nwin + m; a.f_lo = (int)pl->dev_f_lo; a.nw = (int)pl->dev_nw;
    a.rec = b.small + nwin + m + npo; a.dmax = a.rec + 4 * T; a.out = a.dmax + T; a.dcode = b.dcode;
    hipLaunchKernelGGL(k_plan_rounds, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, cs, a);
    hipLaunchKernelGGL(k_plan_prefix, dim3(1), dim3(256), 0, cs, a.rec, a.dmax, (int)T, a.out);
    if (hipMemcpyAsync(b.h_glob + 10, a.out, 2 * sizeof(int), hipMemcpyDeviceToHost, cs) != hipSuccess || hipStreamSynchronize(cs) != hipSuccess) return BT_EHIP;
    if (b.h_glob[11] || b.h_glob[10] < 0) return BT_NEED_EDGES;
    *rounds = b.h_glob[10];
    return BT_OK;
}

// Pass 3 (queued behind the upload of the plan's tables on `stream`): the tile records and pm_edge in the plan's buffer.
int plan_device_fill(const bt_plan *pl, int64_t E, int32_t *d_rec, int32_t *d_pm_edge, int64_t rounds, void *stream) {
    hipStream_t cs = static_cast<hipStream_t>(stream);
    DevPlanBuffers &b = bufs();
    const size_t T = (size_t)pl->info.tiles, nwin = pl->trk_win.size(), m = pl->trk_loc.size(), npo = pl->dev_pair_of.size();
    PlanFillArgs a{};
    a.keys = b.keys + pl->dev_q0; a.vals = b.vals + pl->dev_q0; a.words = b.words; a.jbits = b.jbits; a.E = E;
    a.trk_win = b.small; a.trk_loc = b.small + nwin;
    a.pair_of = b.small + nwin + m; a.f_lo = (int)pl->dev_f_lo; a.nw = (int)pl->dev_nw;
    a.tile_pair0 = pl->dev.tile_pair0; a.tile_npair = pl->dev.tile_npair; a.tile_pairs = pl->dev.tile_pairs;
    a.rec = d_rec; a.dcode = b.dcode; a.pm_edge = d_pm_edge;
    if (hipMemcpyAsync(d_rec, b.small + nwin + m + npo, 4 * T * sizeof(int), hipMemcpyDeviceToDevice, cs) != hipSuccess ||
        hipMemsetAsync(d_pm_edge, 0xff, (size_t)rounds * kLanes * sizeof(int32_t), cs) != hipSuccess) return BT_EHIP;
    hipLaunchKernelGGL(k_plan_fill, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, cs, a);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

// 64-track layouts: the small tables the slot kernels read (queued on `stream`; no synchronisation: the plan's upload follows)
int plan_device_slots_stage(const bt_plan *pl, void *stream) {
    hipStream_t cs = static_cast<hipStream_t>(stream);
    DevPlanBuffers &b = bufs();
    const size_t nwin = pl->trk_win.size(), m = pl->trk_loc.size(), npo = pl->dev_pair_of.size(), noff = pl->dev_off.size(), npb = pl->dev_pbase.size();
    const size_t n_small = nwin + m + npo + noff + npb;
    if (n_small > b.small_cap) {
        (void)hipFree(b.small); (void)hipHostFree(b.h_small);
        const size_t want = n_small + n_small / 4 + 4096;
        if (hipMalloc(reinterpret_cast<void **>(&b.small), want * sizeof(int)) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void **>(&b.h_small), want * sizeof(int), hipHostMallocDefault) != hipSuccess) { b.small_cap = 0; return BT_ENOMEM; }
        b.small_cap = want;
    }
    const size_t ncr = (size_t)pl->info.slots + 2;
    if (ncr > b.crossed_cap) {
        (void)hipFree(b.crossed);
        if (hipMalloc(reinterpret_cast<void **>(&b.crossed), ncr + ncr / 4 + 4096) != hipSuccess) { b.crossed_cap = 0; return BT_ENOMEM; }
        b.crossed_cap = ncr + ncr / 4 + 4096;
    }
    int *h = b.h_small;
    std::copy(pl->trk_win.begin(), pl->trk_win.end(), h);
    std::copy(pl->trk_loc.begin(), pl->trk_loc.end(), h + nwin);
    std::copy(pl->dev_pair_of.begin(), pl->dev_pair_of.end(), h + nwin + m);
    std::copy(pl->dev_off.begin(), pl->dev_off.end(), h + nwin + m + npo);
    std::copy(pl->dev_pbase.begin(), pl->dev_pbase.end(), h + nwin + m + npo + noff);
    if (hipMemcpyAsync(b.small, h, n_small * sizeof(int), hipMemcpyHostToDevice, cs) != hipSuccess ||
        hipMemsetAsync(b.crossed, 0, ncr, cs) != hipSuccess) return BT_EHIP;
    return BT_OK;
}

// ... and the arrays themselves, in the plan's buffer (queued behind the upload of its tables)
int plan_device_em_verdict() { return bufs().h_glob ? bufs().h_glob[12] : 1; }

int plan_device_slots_fill(const bt_plan *pl, int64_t E, int32_t *d_slot_edge, int32_t *d_slot_pair, uint16_t *d_slot_lab, uint8_t *d_slot_lp,
                           uint16_t *d_cut8, uint16_t *d_cut16, void *stream, const DevWptOut *wpt) {
    hipStream_t cs = static_cast<hipStream_t>(stream);
    DevPlanBuffers &b = bufs();
    const size_t nwin = pl->trk_win.size(), m = pl->trk_loc.size(), npo = pl->dev_pair_of.size(), n = (size_t)pl->info.slots * kLanes;
    PlanSlotArgs a{};
    a.keys = b.keys + pl->dev_q0; a.vals = b.vals + pl->dev_q0; a.words = b.words; a.jbits = b.jbits; a.E = E;
    a.trk_win = b.small; a.trk_loc = b.small + nwin; a.pair_of = b.small + nwin + m; a.off = b.small + nwin + m + npo;
    a.pbase = pl->dev_pbase.empty() ? nullptr : b.small + nwin + m + npo + pl->dev_off.size();
    a.f_lo = (int)pl->dev_f_lo; a.nw = (int)pl->dev_nw; a.fixedp = (int)pl->info.fixedp;
    const PlanDev &P = pl->dev;
    a.tile_slot0 = P.tile_slot0; a.tile_cam0 = P.tile_cam0; a.tile_ncam = P.tile_ncam; a.tile_cams = P.tile_cams;
    a.tile_pair0 = P.tile_pair0; a.tile_npair = P.tile_npair; a.tile_pairs = P.tile_pairs; a.tile_nslot = P.tile_nslot;
    a.slot_edge = d_slot_edge; a.slot_pair = d_slot_pair; a.slot_lab = d_slot_lab; a.slot_lp = d_slot_lp;
    a.crossed = b.crossed; a.cut8 = d_cut8; a.cut16 = d_cut16; a.T = (int)pl->info.tiles;
    if (hipMemsetAsync(d_slot_edge, 0xff, n * sizeof(int32_t), cs) != hipSuccess || hipMemsetAsync(d_slot_pair, 0, n * sizeof(int32_t), cs) != hipSuccess ||
        hipMemsetAsync(d_slot_lab, 0xff, n * sizeof(uint16_t), cs) != hipSuccess || hipMemsetAsync(d_slot_lp, 0, n, cs) != hipSuccess) return BT_EHIP;
    if (wpt) {
        a.slot_code = wpt->slot_code; a.tile_la = wpt->tile_la; a.tile_rec = wpt->tile_rec; a.it_edge = wpt->it_edge; a.tile_sinfo = wpt->tile_sinfo;
        a.em_bad = b.glob + 12; a.tile_ntrk = P.tile_ntrk;
        b.h_glob[12] = 0;
        if (hipMemsetAsync(a.slot_code, 0xff, n * sizeof(uint16_t), cs) != hipSuccess || hipMemsetAsync(a.tile_la, 0xff, (size_t)a.T * kLanes, cs) != hipSuccess ||
            hipMemsetAsync(a.em_bad, 0, sizeof(int), cs) != hipSuccess ||
            (a.it_edge && hipMemsetAsync(a.it_edge, 0xff, (size_t)wpt->its * kLanes * sizeof(int32_t), cs) != hipSuccess)) return BT_EHIP;
    }
    hipLaunchKernelGGL(k_plan_slots, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, cs, a);
    hipLaunchKernelGGL(k_plan_cuts, dim3((unsigned)((2 * a.T + 255) / 256)), dim3(256), 0, cs, a);
    if (wpt && a.tile_sinfo) {
        hipLaunchKernelGGL(k_plan_sinfo, dim3((unsigned)((a.T + 3) / 4)), dim3(256), 0, cs, a);
        if (hipMemcpyAsync(b.h_glob + 12, a.em_bad, sizeof(int), hipMemcpyDeviceToHost, cs) != hipSuccess) return BT_EHIP;
    }
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

}  // namespace bt
