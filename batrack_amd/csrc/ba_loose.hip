// ba_loose.hip — LOOSE tracks: tracks seen by more than 64 free cameras (kTileCamHard; more than 32 where a plan has only a few such: ba_plan.cpp).  A tile keeps its E as [6 x cameras][tracks] in
// LDS with at most 64 cameras; a hub track — a landmark of a global / loop-closing adjustment seen from a hundred keyframes — has
// no place in one.  The reference's dense E [n, m, 6] (ba.py:268-292) has no such clause, so these tracks take a slow-but-correct
// path of their own, a workgroup per track, everything in double:
//   k_loose_reduce   the track's edges: residual, Jacobians, robust weights (projective_ops.py:54-100, ba.py:228-266); the 27
//                    per-pair sums by atomics into the accumulators k_pair_finalize turns into B and v (it runs behind this
//                    kernel); C, w -> Q, w' (ba.py:296-311); the track's E over ALL free cameras of the graph in LDS
//                    (E[b] += Ej, E[a] += -Ad^T Ej), then its Schur term S -= Q E E^T, y -= Q w' E (ba.py:314-323) by atomics
//   k_loose_update   dZ = Q (w' - sum_edges Jz^T W Jj delta), delta = dX_j - Ad dX_i (ba.py:328-334; the edges re-evaluated as
//                    k_update does for the tiles' tracks)
// The pair geometry is left in the workspace in the plan's own format for k_pair_finalize (a pair that only loose tracks see has
// no tile to compute it), and is what the edge maths here uses — float32-rounded R, t in plans whose tile kernels round them
// (the Ad sandwich of k_pair_finalize and the Jacobians must be of one linearisation point).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "ba_edge.hpp"
#include "ba_kernels.hpp"
#include "dev_cache.hpp"

namespace bt {
namespace lz {

constexpr int kThreads = 256;

// geometry of the edge's pair as the plan's kernels have it; g: the non-RAWK layout edge_eval<double> reads
__device__ __forceinline__ void geometry(const PlanDev &pd, const StepArgs &a, int gp, int rawk, bool store, double (&g)[kPairGeomFloats]) {
    const int i = pd.pair_i[gp], j = pd.pair_j[gp];
    pair_geometry<double, false>(a.poses, a.intr, i, j, g);
    if (!a.prec) {
#pragma unroll
        for (int c = 0; c < 12; ++c) g[c] = (double)(float)g[c];
    }
    if (!store) return;
    if (a.prec) {
        double *dst = reinterpret_cast<double *>(a.pairgeo) + (size_t)gp * kPairGeomFloats;
#pragma unroll
        for (int c = 0; c < kPairGeomFloats; ++c) dst[c] = g[c];
    } else {
        float *dst = a.pairgeo + (size_t)gp * kPairGeomFloats;
#pragma unroll
        for (int c = 0; c < kPairGeomFloats; ++c) dst[c] = (float)g[c];
        if (rawk) { dst[12] = a.intr[4 * i]; dst[13] = a.intr[4 * i + 1]; }          // (pair_geometry<float, true>: fx_i, fy_i themselves)
    }
}

__device__ __forceinline__ double block_sum(double v, double *red) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) s += red[w];
    return s;
}

template <bool SO>
__global__ __launch_bounds__(kThreads) void k_loose_reduce(PlanDev pd, StepArgs a, int rawk) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double red[kThreads / 64], sQ[2];
    __shared__ int s_nl;
    const int l = blockIdx.x, tid = threadIdx.x, n = pd.n;
    const int k = pd.lz_trk[l], q0 = pd.lz_ptr[l], q1 = pd.lz_ptr[l + 1];
    double *Eg = sm;                                         // [n][6]: the track's E by free camera
    int *touched = reinterpret_cast<int *>(Eg + 6 * (size_t)n);    // [n]: 1 = the track sees this camera; then the list of those
    if (!SO) {
        for (int i = tid; i < 6 * n; i += kThreads) Eg[i] = 0.0;
        for (int i = tid; i < n; i += kThreads) touched[i] = 0;
    }
    __syncthreads();
    const int patch = pd.kx[k];
    const double px = a.patches[3 * patch], py = a.patches[3 * patch + 1], d = a.patches[3 * patch + 2];
    double C = 0.0, w = 0.0;
    for (int q = q0 + tid; q < q1; q += kThreads) {
        const int e = pd.lz_edge[q], gp = pd.lz_pair[q];
        double g[kPairGeomFloats];
        geometry(pd, a, gp, rawk, !SO, g);
        const float *tp = a.targets + (size_t)e * a.tstride;
        const float2 wt = reinterpret_cast<const float2 *>(a.weights)[e];
        EdgeQT<double> o;
        edge_eval<double>(g, px, py, d, (double)tp[0], (double)tp[1], (double)wt.x, (double)wt.y, a, o);
        C += o.W0 * o.jz0 * o.jz0 + o.W1 * o.jz1 * o.jz1;                    // ba.py:287
        w += o.W0 * o.jz0 * o.r0 + o.W1 * o.jz1 * o.r1;                      // ba.py:292
        if (SO) continue;
        const double r0[6] = {o.a0, 0.0, o.a2, o.a3, o.a4, o.a5}, r1[6] = {0.0, o.b1, o.b2, o.b3, o.b4, o.b5};
        double wa[6], wb[6], Ej[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) { wa[c] = o.W0 * r0[c]; wb[c] = o.W1 * r1[c]; Ej[c] = wa[c] * o.jz0 + wb[c] * o.jz1; }      // ba.py:263
        // Bjj = Jj^T W Jj (21) and gj = Jj^T W r (6) of the pair (ba.py:260,266): k_pair_finalize's accumulators
        double *acc = a.pairacc + (size_t)gp * kPairAccStride;
        int vi = 0;
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int c = p; c < 6; ++c, ++vi) {
                const double v = wa[p] * r0[c] + wb[p] * r1[c];
                if (v != 0.0) atomicAdd(acc + vi, v);
            }
#pragma unroll
        for (int c = 0; c < 6; ++c) atomicAdd(acc + 21 + c, wa[c] * o.r0 + wb[c] * o.r1);
        const int ia = pd.pair_i[gp] - pd.fixedp, ib = pd.pair_j[gp] - pd.fixedp;
        if (ib >= 0) {
#pragma unroll
            for (int c = 0; c < 6; ++c) atomicAdd(&Eg[6 * ib + c], Ej[c]);
            touched[ib] = 1;
        }
        if (ia >= 0) {
            // Ei = -Ad(Gij)^T Ej:  o_tau = R^T e_tau,  o_phi = R^T (e_tau x t + e_phi)          (se3.h:58-67)
            const double cx = Ej[1] * g[11] - Ej[2] * g[10] + Ej[3], cy = Ej[2] * g[9] - Ej[0] * g[11] + Ej[4], cz = Ej[0] * g[10] - Ej[1] * g[9] + Ej[5];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                atomicAdd(&Eg[6 * ia + c], -(g[c] * Ej[0] + g[3 + c] * Ej[1] + g[6 + c] * Ej[2]));
                atomicAdd(&Eg[6 * ia + 3 + c], -(g[c] * cx + g[3 + c] * cy + g[6 + c] * cz));
            }
            touched[ia] = 1;
        }
    }
    C = block_sum(C, red);
    w = block_sum(w, red);
    if (tid == 0) {                                          // ba.py:296-311
        const double mono = (double)a.mono[(size_t)patch * a.mstride];
        const double pm = mono > 1e-2 ? (double)a.alpha : 0.0;
        const double lm = a.lmbda_trk ? (double)a.lmbda_trk[pd.trk_off + k] : (double)a.lmbda;
        const double Q = 1.0 / (C + pm + lm), wp = w - pm * (d - mono);
        if (a.prec) reinterpret_cast<double2 *>(a.qw)[k] = make_double2(Q, wp);
        else a.qw[k] = make_float2((float)Q, (float)wp);
        sQ[0] = Q; sQ[1] = Q * wp;
        if (!SO) {                                           // the cameras the track sees, in ascending order
            int nl = 0;
            for (int c = 0; c < n; ++c) if (touched[c]) touched[nl++] = c;      // (in place: nl <= c)
            s_nl = nl;
        }
    }
    if (SO) return;
    __syncthreads();
    // ---- the track's Schur term (ba.py:321-323): S[u, v] -= Q E_u E_v^T over its cameras u >= v, y[u] -= Q w' E_u
    const int nl = s_nl;
    const double Q = sQ[0], be = sQ[1];
    const size_t D = (size_t)pd.D;
    const int npair = nl * (nl + 1) / 2;
    for (int idx = tid; idx < npair * 36; idx += kThreads) {
        const int pr = idx / 36, el = idx - 36 * pr, r = el / 6, c = el - 6 * r;
        int u = (int)((sqrtf(8.0f * (float)pr + 1.0f) - 1.0f) * 0.5f);
        while ((u + 1) * (u + 2) / 2 <= pr) ++u;
        while (u * (u + 1) / 2 > pr) --u;
        const int v = pr - u * (u + 1) / 2;
        const int cu = touched[u], cv = touched[v];          // cu >= cv (the list ascends)
        if (u == v && c > r) continue;                       // S holds the lower triangle
        const double val = Q * Eg[6 * cu + r] * Eg[6 * cv + c];
        if (val != 0.0) atomicAdd(&a.S[(size_t)(6 * cu + r) * D + 6 * cv + c], -val);
    }
    for (int idx = tid; idx < nl * 6; idx += kThreads) {
        const int cu = touched[idx / 6];
        atomicAdd(&a.y[6 * cu + idx % 6], -be * Eg[6 * cu + idx % 6]);
    }
}

__global__ __launch_bounds__(kThreads) void k_loose_update(PlanDev pd, StepArgs a) {
    __shared__ double red[kThreads / 64];
    const int l = blockIdx.x, tid = threadIdx.x;
    const int k = pd.lz_trk[l], q0 = pd.lz_ptr[l], q1 = pd.lz_ptr[l + 1];
    const int patch = pd.kx[k];
    const double px = a.patches[3 * patch], py = a.patches[3 * patch + 1], d = a.patches[3 * patch + 2];
    double acc = 0.0;
    for (int q = q0 + tid; q < q1; q += kThreads) {
        const int e = pd.lz_edge[q], gp = pd.lz_pair[q];
        double g[kPairGeomFloats];
        geometry(pd, a, gp, 0, false, g);
        const float *tp = a.targets + (size_t)e * a.tstride;
        const float2 wt = reinterpret_cast<const float2 *>(a.weights)[e];
        EdgeQT<double> o;
        edge_eval<double>(g, px, py, d, (double)tp[0], (double)tp[1], (double)wt.x, (double)wt.y, a, o);
        const int ia = pd.pair_i[gp] - pd.fixedp, ib = pd.pair_j[gp] - pd.fixedp;
        double xi[6] = {0, 0, 0, 0, 0, 0}, xj[6] = {0, 0, 0, 0, 0, 0};
        if (ia >= 0) for (int c = 0; c < 6; ++c) xi[c] = (double)a.dx[6 * ia + c];
        if (ib >= 0) for (int c = 0; c < 6; ++c) xj[c] = (double)a.dx[6 * ib + c];
        // delta = dX_j - Ad(Gij) dX_i,  Ad (tau, phi) = (R tau + t x (R phi), R phi)                (se3.h:58-67)
        double Rt[3], Rp[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            Rt[r] = g[3 * r] * xi[0] + g[3 * r + 1] * xi[1] + g[3 * r + 2] * xi[2];
            Rp[r] = g[3 * r] * xi[3] + g[3 * r + 1] * xi[4] + g[3 * r + 2] * xi[5];
        }
        const double dl[6] = { xj[0] - (Rt[0] + g[10] * Rp[2] - g[11] * Rp[1]), xj[1] - (Rt[1] + g[11] * Rp[0] - g[9] * Rp[2]),
                               xj[2] - (Rt[2] + g[9] * Rp[1] - g[10] * Rp[0]), xj[3] - Rp[0], xj[4] - Rp[1], xj[5] - Rp[2] };
        const double d0 = o.a0 * dl[0] + o.a2 * dl[2] + o.a3 * dl[3] + o.a4 * dl[4] + o.a5 * dl[5];
        const double d1 = o.b1 * dl[1] + o.b2 * dl[2] + o.b3 * dl[3] + o.b4 * dl[4] + o.b5 * dl[5];
        acc += o.W0 * o.jz0 * d0 + o.W1 * o.jz1 * d1;
    }
    acc = block_sum(acc, red);
    if (tid == 0) {
        double Q, wp;
        if (a.prec) { const double2 qw = reinterpret_cast<const double2 *>(a.qw)[k]; Q = qw.x; wp = qw.y; }
        else { const float2 qw = a.qw[k]; Q = qw.x; wp = qw.y; }
        float dd = (float)(d + Q * (wp - acc));                              // ba.py:328, :333
        dd = dd < 1e-3f ? 1e-3f : dd;
        dd = dd > 10.0f ? 10.0f : dd;
        a.patches_out[3 * patch] = (float)px; a.patches_out[3 * patch + 1] = (float)py; a.patches_out[3 * patch + 2] = dd;
    }
}

}  // namespace lz

// the loose tracks' share of the reduce phase (behind the Jacobian kernel, in front of k_pair_finalize) ...
int launch_loose_reduce(const PlanDev &pd, const StepArgs &a, bool so, hipStream_t st) {
    if (pd.nlz <= 0) return BT_OK;
    const int rawk = (edge_applies(pd) || stream_applies(pd)) ? 1 : 0;
    const size_t lds = so ? 0 : (size_t)pd.n * (6 * sizeof(double) + sizeof(int));
    if (lds > 150 * 1024) return BT_EUNSUPPORTED;
    if (so) hipLaunchKernelGGL(lz::k_loose_reduce<true>, dim3(pd.nlz), dim3(lz::kThreads), 0, st, pd, a, rawk);
    else {
        static LdsLimit lds_limit;
        if (!lds_limit.ensure(reinterpret_cast<const void *>(&lz::k_loose_reduce<false>), lds, pd.dev_id)) return BT_EHIP;
        hipLaunchKernelGGL(lz::k_loose_reduce<false>, dim3(pd.nlz), dim3(lz::kThreads), lds, st, pd, a, rawk);
    }
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

// ... and of a pose+structure step's last kernel
int launch_loose_update(const PlanDev &pd, const StepArgs &a, hipStream_t st) {
    if (pd.nlz <= 0) return BT_OK;
    hipLaunchKernelGGL(lz::k_loose_update, dim3(pd.nlz), dim3(lz::kThreads), 0, st, pd, a);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

}  // namespace bt
