// ba_edge2u.hip — k_edge2u: the structure-only step and the depth back-substitution of graphs of many SLOT-UNIFORM tiles, with
// the machinery of k_edge2 (ba_edge2.hip): TWO EDGES PER LANE on the edge-major tables (it_edge, tile_sinfo, tile_rec), the two
// edges' float32 arithmetic packed (v_pk_fma_f32 on explicit 2-vectors), the reprojection in float64 on the pair's
// float32-rounded geometry held in LDS as doubles, a track's normalised source coordinates formed once per tile, the operands of
// a step in one of two register sets by the step's parity (gathers two steps ahead, edge ids four), the next tile's context
// requested a tile ahead and landed where the wait is cheap.  Round 6: these two passes were still round 4's one-edge-per-lane
// k_edge (70.9 us at 8.4M edges — a quarter of the step, more bytes fetched than the Jacobian kernel).
//   MODE kUSO   structure-only step (ba.py:316-317): C, w per track -> (Q, w') for k_update<true>
//   MODE kUUpd  depth back-substitution of a pose+structure step (ba.py:328-334): dZ = Q (w' - sum_edges Jz^T W Jj delta), where
//               delta = dX_j - Ad(G_ij) dX_i per camera pair — the stored E is never needed
// No Schur product, no pair sums, no E: ~100 registers, four waves per SIMD, one wave per workgroup.  What a lead lane would
// have to load or store per step (a track's (Q, w'), its new disparity) goes through LDS and is read / written once per tile by
// the lane that IS the track: coalesced, and no load inside a step whose result the step needs (that is a wait for every gather
// in flight).
// Reference: ba.py:228-337, projective_ops.py:54-100.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>

#include "ba_edge2.hpp"
#include "dev_cache.hpp"

namespace bt {
namespace e2 {

enum { kUSO = 1, kUUpd = 2 };
// floats per pair in LDS: t (3), fx_j | fy_j, delta (6: kUUpd), - ; rows of 48 bytes: the b128 reads of 8 neighbouring pairs
// fall into different bank groups
constexpr int kGeoU = 12;

template <int MODE, int LGS, int LOSS>
__global__ __launch_bounds__(64, 4) void k_edge2u(PlanDev pd, StepArgs a, int tiles_per_wave) {
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    const int lane = threadIdx.x;
    const int mtp = pd.max_tile_pairs > 0 ? pd.max_tile_pairs : 1;
    // LDS carve-up (wave-private)
    double *ptD = lds_d;                                               // [64][2]: X0, Y0 of every track of the tile
    double *srcK = ptD + 128;                                          // 1/fx_i, 1/fy_i, cx_i, cy_i of the tile's source camera
    double *geoD = srcK + 4;                                           // [mtp][kGeoDS]
    float *ptS = reinterpret_cast<float *>(geoD + mtp * kGeoDS);       // [64]: disparity of every track of the tile
    float *ptO = ptS + 64;                                             // [64][2]: kUSO: C, w of the track; kUUpd: [0] = its sum over the edges
    float *geoU = ptO + 128;                                           // [mtp][kGeoU]

    auto olane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    const int gw = blockIdx.x;
    const int t_begin = gw * tiles_per_wave, t_end = min(pd.T, t_begin + tiles_per_wave);
    if (t_begin >= t_end) return;

    // ---- the wave's stream of steps (see k_edge2): step J = the iterations (2J, 2J + 1) of it_edge, one edge of each per lane
    Rec rec = load_rec(pd, t_begin);
    Rec rec_n = t_begin + 1 < t_end ? load_rec(pd, t_begin + 1) : rec;
    int gj = rec.it0 >> 1, gj_end;
    { const Rec last = load_rec(pd, t_end - 1); gj_end = (last.it0 + last.nit + 1) >> 1; }
    auto load_ids = [&](int j, int &ea, int &eb) {
        const unsigned f = (unsigned)min(j, gj_end - 1) * 2u * kLanes + (unsigned)lane;
        ea = pd.it_edge[f]; eb = pd.it_edge[f + kLanes];
    };
    auto gather = [&](int &ea, int &eb, int jn, f2 &tu, f2 &tv, f2 &w0, f2 &w1, int &fl) {
        const unsigned ua = (unsigned)max(ea, 0), ub = (unsigned)max(eb, 0);
        const unsigned ta = ua * (unsigned)a.tstride, tb = ub * (unsigned)a.tstride;     // (launch_edge checks that byte offsets fit 32 bits)
        const float *pa_ = a.targets + ta, *pb_ = a.targets + tb;
        const float2 *wa_ = reinterpret_cast<const float2 *>(a.weights) + ua, *wb_ = reinterpret_cast<const float2 *>(a.weights) + ub;
        fl = (ea >= 0 ? 1 : 0) | (eb >= 0 ? 2 : 0);
        asm volatile("" : "+v"(fl));
        BT_E2_SB;
        load_ids(jn, ea, eb);                // (issued first: whoever waits for ids issued behind the gathers waits for the gathers)
        BT_E2_SB;
        tu.x = pa_[0]; tv.x = pa_[1];
        tu.y = pb_[0]; tv.y = pb_[1];
        const float2 wa = *wa_, wb = *wb_;
        w0 = f2{wa.x, wb.x}; w1 = f2{wa.y, wb.y};
    };
    f2 tu_q[2], tv_q[2], w0_q[2], w1_q[2];
    int fl_q[2], ea_q[2], eb_q[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        load_ids(gj + k, ea_q[k], eb_q[k]);
        gather(ea_q[k], eb_q[k], gj + k + 2, tu_q[k], tv_q[k], w0_q[k], w1_q[k], fl_q[k]);
    }
    int par = 0;

    // ---- per-tile context of the first tile (lane = track of the tile)
    int kx_c = pd.tile_kx[(unsigned)t_begin * kLanes + (unsigned)lane];
    unsigned si_c = pd.tile_sinfo[(unsigned)t_begin * kLanes + (unsigned)(lane & ((1 << rec.lgS) - 1))];
    float px, py, pdisp, mono_v = 0.0f, lm_v = a.lmbda;
    float2 qw_c = make_float2(0.0f, 0.0f);
    {
        const unsigned kq = (unsigned)max(kx_c, 0);
        px = a.patches[3u * kq]; py = a.patches[3u * kq + 1u]; pdisp = a.patches[3u * kq + 2u];
        if (MODE == kUSO) {
            mono_v = a.mono[kq * (unsigned)a.mstride];
            if (a.lmbda_trk) lm_v = a.lmbda_trk[(unsigned)pd.trk_off + (unsigned)min(rec.trk0 + lane, pd.m - 1)];
        } else {
            qw_c = a.qw[(unsigned)min(rec.trk0 + lane, pd.m - 1)];
        }
    }
    const double b0 = (double)a.b0, b1 = (double)a.b1, b2 = (double)a.b2, b3 = (double)a.b3;

#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int flags = tile == t_begin ? 0 : rec.flags;
        const bool has_next = tile + 1 < t_end;
        const int lgS = LGS >= 0 ? LGS : rec.lgS, S1 = (1 << lgS) - 1, G = kLanes >> lgS;
        const unsigned lp = (si_c >> 8) & 0xffu;
        // ---- new pair list: the pairs' geometry (and, for the back-substitution, what dX moves the pair by)
        if (!(flags & 2)) {
            for (int p = olane(); p < rec.npair; p += 64) {
                const int gp = pd.tile_pairs[rec.pair0 + p];
                double *gd = geoD + p * kGeoDS;
                float *gu = geoU + p * kGeoU;
                if (MODE == kUUpd) {
                    // the geometry the step's Jacobian kernel left (k_edge2: R, t rounded to float32 — the linearisation point of
                    // S and y), delta = dX_j - Ad(G_ij) dX_i, Ad(G_ij)(tau, phi) = (R tau + t x (R phi), R phi)   (se3.h:58-67)
                    float gg[kPairGeomFloats];
                    const float4 *src = reinterpret_cast<const float4 *>(a.pairgeo + (size_t)gp * kPairGeomFloats);
#pragma unroll
                    for (int c = 0; c < kPairGeomFloats / 4; ++c) { const float4 t4 = src[c]; gg[4*c] = t4.x; gg[4*c + 1] = t4.y; gg[4*c + 2] = t4.z; gg[4*c + 3] = t4.w; }
                    const int ia = pd.pair_i[gp] - pd.fixedp, ib = pd.pair_j[gp] - pd.fixedp;
                    float xi[6] = {0, 0, 0, 0, 0, 0}, xj[6] = {0, 0, 0, 0, 0, 0};
                    if (ia >= 0) for (int c = 0; c < 6; ++c) xi[c] = a.dx[6 * ia + c];
                    if (ib >= 0) for (int c = 0; c < 6; ++c) xj[c] = a.dx[6 * ib + c];
                    float Rt[3], Rp[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        Rt[r] = gg[3*r] * xi[0] + gg[3*r + 1] * xi[1] + gg[3*r + 2] * xi[2];
                        Rp[r] = gg[3*r] * xi[3] + gg[3*r + 1] * xi[4] + gg[3*r + 2] * xi[5];
                    }
#pragma unroll
                    for (int c = 0; c < 12; ++c) gd[c] = (double)gg[c];
#pragma unroll
                    for (int c = 0; c < 4; ++c) gd[12 + c] = (double)gg[16 + c];
                    gu[0] = gg[9]; gu[1] = gg[10]; gu[2] = gg[11]; gu[3] = gg[16]; gu[4] = gg[17];
                    gu[5] = xj[0] - (Rt[0] + gg[10] * Rp[2] - gg[11] * Rp[1]);
                    gu[6] = xj[1] - (Rt[1] + gg[11] * Rp[0] - gg[9]  * Rp[2]);
                    gu[7] = xj[2] - (Rt[2] + gg[9]  * Rp[1] - gg[10] * Rp[0]);
                    gu[8] = xj[3] - Rp[0]; gu[9] = xj[4] - Rp[1]; gu[10] = xj[5] - Rp[2]; gu[11] = 0.0f;
                    if (p == 0) { srcK[0] = frcp((double)gg[12]); srcK[1] = frcp((double)gg[13]); srcK[2] = (double)gg[14]; srcK[3] = (double)gg[15]; }
                } else {
                    const int ij = pd.tile_ij[(unsigned)tile * (unsigned)mtp + (unsigned)p];
                    double g[kPairGeomFloats];
                    pair_geometry<double, true>(a.poses, a.intr, ij & 0xffff, ij >> 16, g);
                    // (R, t rounded to float32 as in k_edge2: the structure-only step is then the same function of the state
                    //  whichever of the two kernels evaluates the edge)
#pragma unroll
                    for (int c = 0; c < 12; ++c) gd[c] = (double)(float)g[c];
#pragma unroll
                    for (int c = 0; c < 4; ++c) gd[12 + c] = g[16 + c];
                    gu[0] = (float)g[9]; gu[1] = (float)g[10]; gu[2] = (float)g[11]; gu[3] = (float)g[16]; gu[4] = (float)g[17];
                    if (p == 0) { srcK[0] = frcp(g[12]); srcK[1] = frcp(g[13]); srcK[2] = g[14]; srcK[3] = g[15]; }
                }
            }
            BT_E2_WAVE_SYNC();
        }
        {   // the tracks' normalised source coordinates (projective_ops.py:19-29) and disparities where every lane can read them
            const double X0 = ((double)px - srcK[2]) * srcK[0], Y0 = ((double)py - srcK[3]) * srcK[1];
            const int ln = olane();
            reinterpret_cast<double2 *>(ptD)[ln] = make_double2(X0, Y0);
            ptS[ln] = pdisp;
        }
        const int tl = lane >> lgS;
        const bool lead = (lane & S1) == 0;
        const bool hasB = G < 64;
        BT_E2_WAVE_SYNC();
        // ---- next tile's context: requested now, lands under this tile's steps
        int kx_n = -1;
        unsigned si_n = 0u;
        float px_n = 0.0f, py_n = 0.0f, pd_n = 0.0f, mono_n = 0.0f, lm_n = a.lmbda;
        float2 qw_n = make_float2(0.0f, 0.0f);
        RawRec raw_nn = load_raw(pd, min(tile + 2, t_end - 1));
        if (has_next) {
            kx_n = pd.tile_kx[(unsigned)(tile + 1) * kLanes + (unsigned)lane];
            si_n = pd.tile_sinfo[(unsigned)(tile + 1) * kLanes + (unsigned)(lane & ((1 << rec_n.lgS) - 1))];
            if (MODE == kUSO) { if (a.lmbda_trk) lm_n = a.lmbda_trk[(unsigned)pd.trk_off + (unsigned)min(rec_n.trk0 + lane, pd.m - 1)]; }
            else qw_n = a.qw[(unsigned)min(rec_n.trk0 + lane, pd.m - 1)];
        }

        const int nit2 = (rec.nit + 1) >> 1;
        auto step = [&](auto pc, int it) {
            constexpr int P = decltype(pc)::value;
            const f2 tu = tu_q[P], tv = tv_q[P];
            const int fl = fl_q[P];
            const int trA = it * 2 * G + tl, trB = (trA + G) & 63;
            const double2 xyA = reinterpret_cast<const double2 *>(ptD)[trA], xyB = reinterpret_cast<const double2 *>(ptD)[trB];
            const f2 d = f2{ptS[trA], ptS[trB]};
            double gd[kGeoD];
            {
                const double2 *g2 = reinterpret_cast<const double2 *>(geoD + (size_t)lp * kGeoDS);
#pragma unroll
                for (int c = 0; c < kGeoD / 2; ++c) { const double2 t2 = g2[c]; gd[2 * c] = t2.x; gd[2 * c + 1] = t2.y; }
            }
            const Proj pA = project(gd, xyA.x, xyA.y, d.x, tu.x, tv.x, (fl & 1) != 0, b0, b1, b2, b3);
            const Proj pB = project(gd, xyB.x, xyB.y, d.y, tu.y, tv.y, hasB && (fl & 2) != 0, b0, b1, b2, b3);
            float gu[kGeoU];
            {
                const float4 *g4 = reinterpret_cast<const float4 *>(geoU + (size_t)lp * kGeoU);
#pragma unroll
                for (int c = 0; c < (MODE == kUUpd ? 3 : 2); ++c) { const float4 t4 = g4[c]; gu[4 * c] = t4.x; gu[4 * c + 1] = t4.y; gu[4 * c + 2] = t4.z; gu[4 * c + 3] = t4.w; }
            }
            // ---- Jz, robust weights (and for the back-substitution the rows of Jj) for the two edges at once
            //      (projective_ops.py:80-98, ba.py:247-251)
            const f2 X = f2{pA.X, pB.X}, Y = f2{pA.Y, pB.Y}, Z = f2{pA.Z, pB.Z};
            const f2 vld = f2{pA.ok ? 1.0f : 0.0f, pB.ok ? 1.0f : 0.0f};
            const f2 dj = f2{fabsf(Z.x) > 0.2f ? frcp(Z.x) : 0.0f, fabsf(Z.y) > 0.2f ? frcp(Z.y) : 0.0f};
            const f2 t0 = splat(gu[0]), t1 = splat(gu[1]), t2 = splat(gu[2]);
            const f2 A = splat(gu[3]) * dj, C = splat(gu[4]) * dj;
            const f2 Bc = -(A * (X * dj)), Dc = -(C * (Y * dj));
            const f2 jz0 = fma2(A, t0, Bc * t2), jz1 = fma2(C, t1, Dc * t2);
            const f2 r0u = f2{pA.r0, pB.r0}, r1u = f2{pA.r1, pB.r1};
            const f2 s0 = r0u * r0u, s1 = r1u * r1u;
            const f2 rw0 = f2{robust1<LOSS>(s0.x), robust1<LOSS>(s0.y)};
            const f2 rw1 = f2{robust1<LOSS>(s1.x), robust1<LOSS>(s1.y)};
            const f2 W0 = vld * (w0_q[P] * rw0), W1 = vld * (w1_q[P] * rw1);
            // ---- this step's operands are consumed: the gathers of the step after next go into the same registers
            BT_E2_SB;
            if (it == 0 && has_next) {
                asm volatile("" : "+v"(kx_n));
                const unsigned kq = (unsigned)max(kx_n, 0);
                px_n = a.patches[3u * kq]; py_n = a.patches[3u * kq + 1u]; pd_n = a.patches[3u * kq + 2u];
                if (MODE == kUSO) mono_n = a.mono[kq * (unsigned)a.mstride];
            }
            if (it == nit2 - 1 && has_next) {
                asm volatile("" : "+v"(px_n), "+v"(py_n), "+v"(pd_n), "+v"(mono_n), "+v"(lm_n), "+v"(si_n), "+v"(qw_n.x), "+v"(qw_n.y));
                asm volatile("" : "+v"(raw_nn.r0.x), "+v"(raw_nn.r0.w), "+v"(raw_nn.r1.x), "+v"(raw_nn.r1.y), "+v"(raw_nn.r1.z), "+v"(raw_nn.r1.w));
            }
            gather(ea_q[P], eb_q[P], gj + 4, tu_q[P], tv_q[P], w0_q[P], w1_q[P], fl_q[P]);
            BT_E2_SB;
            const f2 wj0 = W0 * jz0, wj1 = W1 * jz1;
            if (MODE == kUSO) {
                const f2 r0 = vld * r0u, r1 = vld * r1u;
                f2 sv[2] = { fma2(wj0, jz0, wj1 * jz1), fma2(wj0, r0, wj1 * r1) };          // C, w (ba.py:287,292)
                group_sum2(sv, lgS);
                if (lead) {
                    reinterpret_cast<float2 *>(ptO)[trA] = make_float2(sv[0].x, sv[1].x);
                    if (hasB) reinterpret_cast<float2 *>(ptO)[trB] = make_float2(sv[0].y, sv[1].y);
                }
            } else {
                const f2 a0 = d * A, a2 = d * Bc, a3 = Bc * Y, a4 = fma2(A, Z, -(Bc * X)), a5 = -(A * Y);
                const f2 b1_ = d * C, b2_ = d * Dc, b3_ = fma2(Dc, Y, -(C * Z)), b4_ = -(Dc * X), b5_ = C * X;
                const f2 d0 = fma2(a0, splat(gu[5]), fma2(a2, splat(gu[7]), fma2(a3, splat(gu[8]), fma2(a4, splat(gu[9]), a5 * splat(gu[10])))));
                const f2 d1 = fma2(b1_, splat(gu[6]), fma2(b2_, splat(gu[7]), fma2(b3_, splat(gu[8]), fma2(b4_, splat(gu[9]), b5_ * splat(gu[10])))));
                f2 sv[1] = { fma2(wj0, d0, wj1 * d1) };                                     // Jz^T W Jj delta of the lane's two edges
                group_sum2(sv, lgS);
                if (lead) {
                    ptO[2 * trA] = sv[0].x;
                    if (hasB) ptO[2 * trB] = sv[0].y;
                }
            }
            ++gj;
        };
        {
            int it = 0;
            if (par) { step(IC<1>{}, it); ++it; }
#pragma unroll 1
            for (; it + 1 < nit2; it += 2) { step(IC<0>{}, it); step(IC<1>{}, it + 1); }
            if (it < nit2) { step(IC<0>{}, it); par = 1; } else par = 0;
        }
        // ---- the tile's tracks, one per lane (coalesced: the tracks of a tile are consecutive patches as a rule)
        BT_E2_WAVE_SYNC();
        {
            const int ln = olane();
            const float2 o = reinterpret_cast<const float2 *>(ptO)[ln];
            if (ln < rec.ntrk && kx_c >= 0) {
                if (MODE == kUSO) {                                                         // ba.py:296-311
                    const float pm = mono_v > 1e-2f ? a.alpha : 0.0f;
                    const float Ca = o.x + pm + lm_v;
                    a.qw[(unsigned)rec.trk0 + (unsigned)ln] = make_float2(rcp_f32(Ca), o.y - pm * (pdisp - mono_v));
                } else {
                    float dd = pdisp + qw_c.x * (qw_c.y - o.x);                             // ba.py:328, :333
                    dd = dd < 1e-3f ? 1e-3f : dd;
                    dd = dd > 10.0f ? 10.0f : dd;
                    float *dst = a.patches_out + 3u * (unsigned)kx_c;
                    dst[0] = px; dst[1] = py; dst[2] = dd;
                }
            }
        }
        // ---- rotate the tile context
        if (has_next) {
            rec = rec_n; rec_n = decode_rec(raw_nn); kx_c = kx_n; si_c = si_n;
            px = px_n; py = py_n; pdisp = pd_n; mono_v = mono_n; lm_v = lm_n; qw_c = qw_n;
        }
    }
}

static size_t lds_bytes_u(const PlanDev &pd) {
    const size_t mtp = (size_t)(pd.max_tile_pairs > 0 ? pd.max_tile_pairs : 1);
    return (128 + 4 + mtp * kGeoDS) * sizeof(double) + (64 + 128 + mtp * kGeoU) * sizeof(float);
}

template <int MODE, int LGS, int LOSS>
static int launch_u(const PlanDev &pd, const StepArgs &a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    const size_t lds = (lds_bytes_u(pd) + 255) & ~(size_t)255;
    static LdsLimit lds_limit;
    static std::atomic<int> per_cu_c[kMaxDevices];
    static std::atomic<size_t> per_cu_lds[kMaxDevices];
    DevProps dp;
    if (!device_props(&dp, pd.dev_id)) return BT_EHIP;
    const int dslot = dp.dev >= 0 && dp.dev < kMaxDevices ? dp.dev : 0;
    int per_cu = per_cu_lds[dslot].load(std::memory_order_acquire) == lds ? per_cu_c[dslot].load(std::memory_order_relaxed) : 0;
    if (!per_cu) {
        if (!lds_limit.ensure(reinterpret_cast<const void *>(&k_edge2u<MODE, LGS, LOSS>), lds, pd.dev_id)) return BT_EHIP;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_edge2u<MODE, LGS, LOSS>, 64, lds) != hipSuccess || nb < 1) nb = 1;
        per_cu = nb;
        per_cu_c[dslot].store(nb, std::memory_order_relaxed); per_cu_lds[dslot].store(lds, std::memory_order_release);
    }
    const int max_waves = dp.n_cu * per_cu;
    const int tpw = (pd.T + max_waves - 1) / max_waves, nw = (pd.T + tpw - 1) / tpw;
    if (ev0) hipExtLaunchKernelGGL((k_edge2u<MODE, LGS, LOSS>), dim3(nw), dim3(64), lds, st, ev0, ev1, 0, pd, a, tpw);
    else hipLaunchKernelGGL((k_edge2u<MODE, LGS, LOSS>), dim3(nw), dim3(64), lds, st, pd, a, tpw);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

}  // namespace e2

// k_edge2 / k_edge2u take graphs of many tiles, all slot-uniform (the plan's em_ok), whose tiles see at most 10 cameras (row
// tiles of the Schur product) and 64 camera pairs (one lane per pair in the prologue).
// Measured on the benchmark generator (whole-step times; profiles/r02_kernel_choice.txt, r05_edge2_vs_edge.txt): k_tile is
// fastest up to ~1500 tiles; from 2048 tiles k_edge2 where the tiles are slot-uniform (whole step 122 against k_stream's 130 us at
// 2048 tiles, 151 against 160 at 4096), k_stream otherwise (tiles of more than 64 slots: the edge-major layout does not hold them).
bool edge_applies(const PlanDev &pd) {
    return pd.em_ok && pd.T >= pd.em_min && pd.max_cams <= 10 && pd.max_cams > 0 && pd.max_tile_pairs <= 64 && pd.max_tile_pairs > 0;
}

// mode 0: the pose+structure reduce (k_edge2), 1: structure-only, 2: a pose+structure step's depth back-substitution (k_edge2u)
int launch_edge(const PlanDev &pd, const StepArgs &a, int mode, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    if ((unsigned long long)pd.e_all * (unsigned long long)a.tstride * 4ull >= (1ull << 32)) return BT_EUNSUPPORTED;   // 32-bit byte offsets into the targets
    if ((unsigned long long)pd.p_tot * (unsigned long long)a.mstride * 4ull >= (1ull << 32)) return BT_EUNSUPPORTED;
    if (mode == 0) return launch_edge2(pd, a, st, ev0, ev1);
    const bool s8 = pd.em_lgs == 3;            // the 8-observation graphs of the benchmark generator
#define BT_E2U_LOSS(MODE, LGS)                                                                             \
    (a.loss == BT_LOSS_HUBER ? e2::launch_u<MODE, LGS, BT_LOSS_HUBER>(pd, a, st, ev0, ev1)                 \
     : a.loss == BT_LOSS_CAUCHY ? e2::launch_u<MODE, LGS, BT_LOSS_CAUCHY>(pd, a, st, ev0, ev1)             \
                                : e2::launch_u<MODE, LGS, BT_LOSS_TRIVIAL>(pd, a, st, ev0, ev1))
    if (mode == e2::kUSO) return s8 ? BT_E2U_LOSS(e2::kUSO, 3) : BT_E2U_LOSS(e2::kUSO, -1);
    return s8 ? BT_E2U_LOSS(e2::kUUpd, 3) : BT_E2U_LOSS(e2::kUUpd, -1);
#undef BT_E2U_LOSS
}

}  // namespace bt
