// ba_stream3.hip — k_edge: the structure-only step and the depth back-substitution of graphs of many SLOT-UNIFORM tiles
// (gfx950, wave64).  The pose+structure Jacobian kernel of those graphs is k_edge2 (ba_edge2.hip); round 4's pose+structure
// instantiation of this kernel was measured against it in round 5 (profiles/r05_edge2_vs_edge.txt) and removed.
//
// Layout (ba_plan.cpp "edge-major"): a tile is 64 tracks, a wave's 64 lanes are 64 / S consecutive tracks x their S slots
// (S = slots per track, padded to a power of two), lane = track_in_group * S + slot.  On the track-major edge lists the caller
// builds (batrack.py:399-410: every patch of a keyframe x every frame of the window) that is 64 CONSECUTIVE edges per wave
// instruction — the coalesced stream of targets and weights; a track's sums (C, w; ba.py:284-292) are reductions over S adjacent
// lanes (DPP).  Every wave owns whole tiles and walks a contiguous range of them with no barrier (workgroup = one wave, private
// LDS); the operands of iteration i+1 are gathered and the edge ids of iteration i+2 loaded while iteration i is computed — one
// flat pipeline across tile boundaries.
// MODE kEmSO: structure-only steps (C, w -> Q, w' per track; k_update<true> applies them); kEmUpd: the depth back-substitution
// (see k_update in ba_kernels.hip).
// Reference: ba.py:228-337, projective_ops.py:54-100.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>

#include "ba_edge.hpp"
#include "ba_kernels.hpp"

namespace bt {

enum { kEmSO = 1, kEmUpd = 2 };
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
template <int MODE, int LGS>
__global__ __launch_bounds__(64, 4) void k_edge(PlanDev pd, StepArgs a, int tiles_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const int mtp = pd.max_tile_pairs > 0 ? pd.max_tile_pairs : 1;
    constexpr int GS = MODE == kEmUpd ? kEmUpdGeo : kPairGeomFloats;
    // LDS carve-up (wave-private)
    float *geo = lds;                                                  // [mtp][GS]
    float *ptab = geo + mtp * GS;                                      // [64][4]: x, y, disparity of every track of the tile
    float *cw = ptab + 256;                                            // [2][64]: C, w per track (kEmUpd: [0][64] = patch index)

    // (workgroup -> tile range stays plain: handing each XCD a contiguous block of ranges, which helps k_tile and k_stream,
    //  made this kernel 4 % slower at 8.4M edges)
    const int wg_ = blockIdx.x;
    const int t_begin = wg_ * tiles_per_wave, t_end = min(pd.T, t_begin + tiles_per_wave);
    if (t_begin >= t_end) return;

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
    unsigned si_c = pd.tile_sinfo[(unsigned)t_begin * kLanes + (unsigned)(lane & ((1 << rec.lgS) - 1))];
    float px = 0.0f, py = 0.0f, pdisp = 0.0f, mono_v = 0.0f;
    if (kx_c >= 0) {
        px = a.patches[3u * (unsigned)kx_c]; py = a.patches[3u * (unsigned)kx_c + 1u]; pdisp = a.patches[3u * (unsigned)kx_c + 2u];
        if (MODE != kEmUpd) mono_v = a.mono[(unsigned)kx_c * (unsigned)a.mstride];
    }

#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int flags = tile == t_begin ? 0 : rec.flags;
        const bool has_next = tile + 1 < t_end;
        const int lgS = LGS >= 0 ? LGS : rec.lgS, S1 = (1 << lgS) - 1, G = kLanes >> lgS;
        // ---- new pair list
        if (!(flags & 2)) {
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
                }
            }
        }
        // the tracks' patches where every lane of their group can read them
        reinterpret_cast<float4 *>(ptab)[lane] = make_float4(px, py, pdisp, 0.0f);
        if (MODE == kEmUpd) reinterpret_cast<int *>(cw)[lane] = kx_c;
        // this lane's slot of the tile: its pair
        const unsigned lp = (si_c >> 8) & 0xffu;
        const int tl = lane >> lgS;
        const bool lead = (lane & S1) == 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- next tile's context: requested now, lands under this tile's iterations
        int kx_n = -1;
        unsigned si_n = 0u;
        float px_n = 0.0f, py_n = 0.0f, pd_n = 0.0f, mono_n = 0.0f;
        EmRec rec_nn = rec_n;
        if (tile + 2 < t_end) rec_nn = em_load_rec(pd, tile + 2);
        if (has_next) {
            kx_n = pd.tile_kx[(unsigned)(tile + 1) * kLanes + (unsigned)lane];
            si_n = pd.tile_sinfo[(unsigned)(tile + 1) * kLanes + (unsigned)(lane & ((1 << rec_n.lgS) - 1))];
        }
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

            const bool act = e >= 0;
            const int track = it * G + tl;
            const float4 pt = reinterpret_cast<const float4 *>(ptab)[track];
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
            BT_WPT_EDGE_EVAL(g, pt.x, pt.y, pt.z, tu, tv, w0, w1, a, q);
            if (!act) { q.W0 = 0.0f; q.W1 = 0.0f; q.r0 = 0.0f; q.r1 = 0.0f; }
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
            } else {
                float sv[2] = { q.W0 * q.jz0 * q.jz0 + q.W1 * q.jz1 * q.jz1, q.W0 * q.jz0 * q.r0 + q.W1 * q.jz1 * q.r1 };   // ba.py:287,292
                group_sum(sv, lgS);
                if (lead && track < rec.ntrk) { cw[track] = sv[0]; cw[64 + track] = sv[1]; }
            }
            // rotate the pipeline
#pragma unroll
            for (int k = 0; k + 1 < kEmIds; ++k) e_q[k] = e_q[k + 1];
            e_q[kEmIds - 1] = e_nn;
#pragma unroll
            for (int k = 0; k + 1 < kEmDepth; ++k) { tu_q[k] = tu_q[k + 1]; tv_q[k] = tv_q[k + 1]; w0_q[k] = w0_q[k + 1]; w1_q[k] = w1_q[k + 1]; }
            tu_q[kEmDepth - 1] = tu_x; tv_q[kEmDepth - 1] = tv_x; w0_q[kEmDepth - 1] = w0_x; w1_q[kEmDepth - 1] = w1_x;
        }
        if (MODE == kEmSO) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < rec.ntrk) {                                                // ba.py:296-311
                const float C = cw[lane], wv = cw[64 + lane];
                const float pm = mono_v > 1e-2f ? 1.0f : 0.0f;
                float Ca = C + pm * a.alpha;
                Ca = Ca + (a.lmbda_trk ? a.lmbda_trk[(unsigned)pd.trk_off + (unsigned)rec.trk0 + (unsigned)lane] : a.lmbda);
                a.qw[(unsigned)rec.trk0 + (unsigned)lane] = make_float2(1.0f / Ca, wv - pm * a.alpha * (pdisp - mono_v));
            }
        }

        // ---- rotate the tile context
        if (has_next) {
            rec = rec_n; rec_n = rec_nn; kx_c = kx_n; si_c = si_n;
            px = px_n; py = py_n; pdisp = pd_n; mono_v = mono_n;
        }
    }
}
#undef BT_DPPF

// ------------------------------------------------------------------ dispatch

static size_t edge_lds_bytes(const PlanDev &pd, int mode) {
    const size_t mtp = (size_t)(pd.max_tile_pairs > 0 ? pd.max_tile_pairs : 1);
    return (mtp * (mode == kEmUpd ? kEmUpdGeo : kPairGeomFloats) + 256 + 128) * sizeof(float);
}

// k_edge2 / k_edge take graphs of many tiles, all slot-uniform (the plan's em_ok), whose tiles see at most 10 cameras (row
// tiles of the Schur product) and 64 camera pairs (one lane per pair in the prologue).
// Measured on the benchmark generator (whole-step times; profiles/r02_kernel_choice.txt, r05_edge2_vs_edge.txt): k_tile is
// fastest up to ~1500 tiles; from 2048 tiles k_edge2 where the tiles are slot-uniform (whole step 122 against k_stream's 130 us at
// 2048 tiles, 151 against 160 at 4096), k_stream otherwise (BT_EDGE_PREF_TILES: k_stream keeps the graphs below it where both apply).
bool edge_applies(const PlanDev &pd) {
    static const int off = std::getenv("BT_EDGE_OFF") ? std::atoi(std::getenv("BT_EDGE_OFF")) : 0;   // measurement only
    static const int pref = std::getenv("BT_EDGE_PREF_TILES") ? std::atoi(std::getenv("BT_EDGE_PREF_TILES")) : 0;
    if (off || !pd.em_ok || pd.T < pd.em_min || pd.max_cams > 10 || pd.max_cams <= 0 || pd.max_tile_pairs > 64 || pd.max_tile_pairs <= 0)
        return false;
    return pd.T >= pref || !stream_applies(pd);
}

template <int MODE, int LGS>
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
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k_edge<MODE, LGS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return BT_EHIP;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_edge<MODE, LGS>, 64, lds) != hipSuccess || nb < 1) nb = 1;
        per_cu = nb; per_cu_lds = lds;
    }
    const int max_waves = n_cu * per_cu;
    const int tpw = (pd.T + max_waves - 1) / max_waves, nw = (pd.T + tpw - 1) / tpw;
    if (ev0) hipExtLaunchKernelGGL((k_edge<MODE, LGS>), dim3(nw), dim3(64), lds, st, ev0, ev1, 0, pd, a, tpw);
    else hipLaunchKernelGGL((k_edge<MODE, LGS>), dim3(nw), dim3(64), lds, st, pd, a, tpw);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

// mode 0: the pose+structure reduce (k_edge2), 1: structure-only, 2: a pose+structure step's last kernel
int launch_edge(const PlanDev &pd, const StepArgs &a, int mode, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    if ((unsigned long long)pd.e_all * (unsigned long long)a.tstride * 4ull >= (1ull << 32)) return BT_EUNSUPPORTED;   // 32-bit byte offsets into the targets
    if ((unsigned long long)pd.p_tot * (unsigned long long)a.mstride * 4ull >= (1ull << 32)) return BT_EUNSUPPORTED;
    const bool s8 = pd.em_lgs == 3;            // the 8-observation graphs of the benchmark generator
    if (mode == kEmSO) return s8 ? launch_edge_t<kEmSO, 3>(pd, a, st, ev0, ev1) : launch_edge_t<kEmSO, -1>(pd, a, st, ev0, ev1);
    if (mode == kEmUpd) return s8 ? launch_edge_t<kEmUpd, 3>(pd, a, st, ev0, ev1) : launch_edge_t<kEmUpd, -1>(pd, a, st, ev0, ev1);
    return launch_edge2(pd, a, st, ev0, ev1);
}

}  // namespace bt
