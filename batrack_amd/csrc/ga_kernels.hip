// ga_kernels.hip — losses of the dense global-alignment stage and their gradients (include/batrack_ga.h), gfx950.
//   k_ga_scale     one workgroup per (frame t, slot s): mean of the track scales over the N tracks
//                  (refine_net.py:123-127), then per track: exp scale, the frame's scale grid sampled bilinearly at the
//                  track position (:148-174), the scaled mono disparity (written out: the other terms read it) and the
//                  masked smooth-L1 depth residual (:252-268), optionally formed in float16.
//   k_ga_pairwise  the O(Q S N^2) rigidity term (:199-225): one workgroup per (query frame, slot) and a strip of tracks n;
//                  the slot's and the centre slot's 3-D points of ALL tracks staged in LDS once, every thread walks m.
//                  Compute-bound on the f32 vector pipe (2 distances = 2 sqrt per pair); no N x N tensor exists.
//   k_ga_pts3d     3-D point consistency through pose_j^-1 pose_t (:300-345), one thread per (t, n, s).
// Sums are float32 inside a workgroup, float64 across workgroups (one atomic per workgroup).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "../../include/batrack_ba.h"
#include "../../include/batrack_ga.h"
#include "dev_cache.hpp"

namespace bt {

__device__ __forceinline__ float ga_disp(const void *p, size_t i, bool half) {
    return half ? __half2float(reinterpret_cast<const __half *>(p)[i]) : reinterpret_cast<const float *>(p)[i];
}

__device__ __forceinline__ float block_sum(float v, float *red) {          // blockDim.x <= 1024
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.0f;
    if (threadIdx.x < 64) {
        t = (int)threadIdx.x < nw ? red[threadIdx.x] : 0.0f;
        for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o);
    }
    __syncthreads();
    return t;                                                              // valid in thread 0
}

__global__ __launch_bounds__(256) void k_ga_scale(bt_ga_args a, float *mono_scaled, double *losses) {
    __shared__ float red[16];
    __shared__ float s_mean;
    const int t = blockIdx.x / (int)a.S, s = blockIdx.x % (int)a.S;
    const int T = (int)a.T, N = (int)a.N, S = (int)a.S, gh = (int)a.gh, gw = (int)a.gw;
    const bool half = a.half_disp != 0;
    float part = 0.0f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) part += a.trajs_scales[((size_t)t * N + n) * S + s];
    const float tot = block_sum(part, red);
    if (threadIdx.x == 0) s_mean = tot / (float)N;
    __syncthreads();
    const float mean = s_mean;
    const long long jraw = a.jj[(size_t)t * S + s];
    const int jc = (int)(jraw < 0 ? 0 : (jraw > T - 1 ? T - 1 : jraw));
    const bool patch_ok = jraw >= 0 && jraw < T;
    bool is_query = false;
    for (int q = 0; q < (int)a.Q; ++q) is_query |= a.query[q] == t;
    const float *grid = a.frame_scales + (size_t)jc * gh * gw;
    const float shift = a.frame_shifts[jc];
    float acc = 0.0f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const size_t e = ((size_t)t * N + n) * S + s;
        const float x = a.trajs_2d[2 * e], y = a.trajs_2d[2 * e + 1];
        // F.grid_sample(align_corners=True, zeros padding) of exp(grid / 10) at (x / (W-1), y / (H-1))
        const float gx = x / (float)(a.W - 1) * (float)(gw - 1), gy = y / (float)(a.H - 1) * (float)(gh - 1);
        const float fx0 = floorf(gx), fy0 = floorf(gy);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float wx = gx - fx0, wy = gy - fy0;
        float fs = 0.0f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int yy = y0 + dy, xx = x0 + dx;
                if (yy >= 0 && yy < gh && xx >= 0 && xx < gw)
                    fs += expf(grid[yy * gw + xx] / 10.0f) * (dy ? wy : 1.0f - wy) * (dx ? wx : 1.0f - wx);
            }
        const float mono = ga_disp(a.trajs_disp_mono, e, half), disp = ga_disp(a.trajs_disp, e, half);
        const float ms = mono * fs + shift;
        mono_scaled[e] = ms;
        if (is_query) {
            const float sexp = expf((a.trajs_scales[e] - mean) / a.pw_break);
            const float aligned = sexp * disp;
            float err;
            if (half) err = fabsf(__half2float(__hsub(__float2half(ms), __float2half(aligned))));   // the residual itself in float16
            else err = fabsf(ms - aligned);
            const bool m = a.trajs_vis[e] > 0.9f && patch_ok && sqrtf(x * x + y * y) > 5.0f && disp > 1e-2f;
            acc += m ? (err < 1.0f ? 0.5f * err * err : err - 0.5f) : 0.0f;
        }
    }
    const float bs = block_sum(acc, red);
    if (threadIdx.x == 0 && is_query) atomicAdd(&losses[0], (double)bs / ((double)a.Q * N * S));
}

__device__ __forceinline__ void ga_iproj(float x, float y, float d, const float *K, float *P) {     // geomeotry.py:3-18
    const float depth = 1.0f / fmaxf(d, 1e-2f);
    P[0] = (x - K[2]) / K[0] * depth; P[1] = (y - K[3]) / K[1] * depth; P[2] = depth;
}

// ---- the rigidity term's tracks, staged two by two.
// A SUPER-TRACK p is the tracks 2p and 2p + 1 of the (query frame, slot); its numbers sit in four float4 arrays of LDS so
// that both tracks of a partner arrive as the two halves of packed-float32 operands (v_pk_add_f32 / v_pk_fma_f32):
//   L[5 p + 0] = (Xs0, Xs1, Ys0, Ys1)   + 1 = (Zs0, Zs1, Xm0, Xm1)   + 2 = (Ym0, Ym1, Zm0, Zm1)   + 3 = (vis0, vis1, static0, static1)
//   + 4 = (disparity above iproj's clamp: slot 0, slot 1, centre 0, centre 1 — read by the backward only)
// (a record is 80 B: one address per partner, the reads are immediate offsets, and 16 consecutive lanes' 16 B words fall
//  into 16 different bank groups because 5 is odd)
// (s: the point at the slot, m: at the centre slot; vis is zeroed where the mono disparity fails its test or the track does
// not exist — N odd — so the pair mask min(vis vis', static static') > 0.5 is refine_net.py:213-217's three tests in one).
// A thread owns one super-track and meets a partner super-track per step: two 4 x 16 B LDS reads feed FOUR track pairs
// (16 B of LDS per pair instead of 40), every difference / square / sum is one packed instruction per two pairs.
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kGaStrip = 256;           // super-tracks (2 tracks each) per workgroup of the pairwise kernels

__device__ __forceinline__ f2 ga_fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

struct GaSuper { float4 *rec; };
constexpr int kGaRec = 5;              // float4 per super-track

__device__ __forceinline__ void ga_stage(const bt_ga_args &a, const float *mono_scaled, int i, int s, int mid, const float *Ks, const float *Km,
                                         const GaSuper &L, int M) {
    const int N = (int)a.N, S = (int)a.S;
    const bool half = a.half_disp != 0;
    for (int p = threadIdx.x; p < M; p += blockDim.x) {
        float Ps[2][3] = {{0, 0, 0}, {0, 0, 0}}, Pm[2][3] = {{0, 0, 0}, {0, 0, 0}}, vis[2] = {0, 0}, sta[2] = {0, 0}, oks[2] = {0, 0}, okm[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = 2 * p + h;
            if (n < N) {
                const size_t es = ((size_t)i * N + n) * S + s, em = ((size_t)i * N + n) * S + mid;
                ga_iproj(a.trajs_2d[2 * es], a.trajs_2d[2 * es + 1], mono_scaled[es], Ks, Ps[h]);
                ga_iproj(a.trajs_2d[2 * em], a.trajs_2d[2 * em + 1], mono_scaled[em], Km, Pm[h]);
                vis[h] = ga_disp(a.trajs_disp_mono, es, half) > 1e-2f ? a.trajs_vis[es] : 0.0f;
                sta[h] = a.trajs_static[es];
                oks[h] = mono_scaled[es] > 1e-2f ? 1.0f : 0.0f;       // (iproj clamps the disparity there: no gradient below)
                okm[h] = mono_scaled[em] > 1e-2f ? 1.0f : 0.0f;
            }
        }
        float4 *r = L.rec + kGaRec * p;
        r[0] = make_float4(Ps[0][0], Ps[1][0], Ps[0][1], Ps[1][1]);
        r[1] = make_float4(Ps[0][2], Ps[1][2], Pm[0][0], Pm[1][0]);
        r[2] = make_float4(Pm[0][1], Pm[1][1], Pm[0][2], Pm[1][2]);
        r[3] = make_float4(vis[0], vis[1], sta[0], sta[1]);
        r[4] = make_float4(oks[0], oks[1], okm[0], okm[1]);
    }
}

// one own track (its six coordinates and two mask factors) against the two tracks of a partner super-track
struct GaOwn { float xs, ys, zs, xm, ym, zm, vis, sta; };
struct GaPartner { f2 xs, ys, zs, xm, ym, zm, vis, sta; };

__device__ __forceinline__ GaPartner ga_partner(const float4 *r) {
    const float4 qa = r[0], qb = r[1], qc = r[2], qd = r[3];
    GaPartner q;
    q.xs = f2{qa.x, qa.y}; q.ys = f2{qa.z, qa.w}; q.zs = f2{qb.x, qb.y};
    q.xm = f2{qb.z, qb.w}; q.ym = f2{qc.x, qc.y}; q.zm = f2{qc.z, qc.w};
    q.vis = f2{qd.x, qd.y}; q.sta = f2{qd.z, qd.w};
    return q;
}
template <int H>
__device__ __forceinline__ GaOwn ga_own(const GaPartner &q) {
    GaOwn o;
    o.xs = q.xs[H]; o.ys = q.ys[H]; o.zs = q.zs[H]; o.xm = q.xm[H]; o.ym = q.ym[H]; o.zm = q.zm[H]; o.vis = q.vis[H]; o.sta = q.sta[H];
    return o;
}

// masked |d_s - d_mid| of own track o with the partner's two tracks (refine_net.py:205-222).  Hardware square roots (1 ulp):
// the IEEE-exact sequence is ten instructions per root, two roots per pair
__device__ __forceinline__ f2 ga_pair_loss(const GaOwn &o, const GaPartner &q) {
    const f2 dxs = q.xs - o.xs, dys = q.ys - o.ys, dzs = q.zs - o.zs;
    const f2 dxm = q.xm - o.xm, dym = q.ym - o.ym, dzm = q.zm - o.zm;
    const f2 ss = ga_fma2(dzs, dzs, ga_fma2(dys, dys, dxs * dxs)), sq = ga_fma2(dzm, dzm, ga_fma2(dym, dym, dxm * dxm));
    const f2 ds = f2{__builtin_amdgcn_sqrtf(ss.x), __builtin_amdgcn_sqrtf(ss.y)}, dm = f2{__builtin_amdgcn_sqrtf(sq.x), __builtin_amdgcn_sqrtf(sq.y)};
    const f2 df = ds - dm, pv = q.vis * o.vis, pt = q.sta * o.sta;
    f2 r;
    r.x = fminf(pv.x, pt.x) > 0.5f ? fabsf(df.x) : 0.0f;
    r.y = fminf(pv.y, pt.y) > 0.5f ? fabsf(df.y) : 0.0f;
    return r;
}

__global__ __launch_bounds__(kGaStrip) void k_ga_pairwise(bt_ga_args a, const float *mono_scaled, double *losses) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[16];
    const int T = (int)a.T, N = (int)a.N, S = (int)a.S, mid = S / 2, M = (N + 1) >> 1;
    const int qi = blockIdx.x / S, s = blockIdx.x % S;
    const int i = (int)a.query[qi];
    const long long jraw = a.jj[(size_t)i * S + s];
    if (jraw < 0 || jraw >= T || s == mid) return;             // t_mask, and the centre slot's own difference is zero
    GaSuper L;
    L.rec = reinterpret_cast<float4 *>(sm);
    const long long jm = a.jj[(size_t)i * S + mid];
    ga_stage(a, mono_scaled, i, s, mid, a.intrinsics + 4 * (size_t)jraw, a.intrinsics + 4 * (size_t)(jm < 0 ? 0 : (jm > T - 1 ? T - 1 : jm)), L, M);
    __syncthreads();
    const int j = blockIdx.y * kGaStrip + threadIdx.x;
    float acc = 0.0f;
    if (j < M) {
        // |d_s(n, m) - d_mid(n, m)| is symmetric in (n, m) and zero on the diagonal.  Super-track j takes the pair inside it
        // and the partners p = j + 1 .. j + M / 2 (mod M) — every unordered pair of super-tracks once, the same trip count
        // for every thread — and the sum counts each pair twice (for even M the antipodal partner is met from both ends:
        // half weight).
        const GaPartner me = ga_partner(L.rec + kGaRec * j);
        const GaOwn oa = ga_own<0>(me), ob = ga_own<1>(me);
        acc = 2.0f * ga_pair_loss(oa, me).y;
        const int half_m = M >> 1, full = (M & 1) ? half_m : half_m - 1;
        int p = j + 1 >= M ? 0 : j + 1;
        const float4 *rp = L.rec + kGaRec * p;                 // (walked by increments: no multiply per step)
        f2 acc2 = {0.0f, 0.0f};
#pragma unroll 2
        for (int k = 0; k < full; ++k) {
            const GaPartner q = ga_partner(rp);
            acc2 += ga_pair_loss(oa, q) + ga_pair_loss(ob, q);
            ++p; rp += kGaRec;
            if (p >= M) { p = 0; rp = L.rec; }
        }
        acc += 2.0f * (acc2.x + acc2.y);
        if (!(M & 1) && half_m >= 1) {
            const GaPartner q = ga_partner(rp);
            const f2 r = ga_pair_loss(oa, q) + ga_pair_loss(ob, q);
            acc += r.x + r.y;
        }
    }
    const float bs = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(&losses[1], (double)bs / ((double)a.Q * S * N * N));
}

__device__ __forceinline__ void ga_qrot(const float *q, const float *v, float *o) {
    const float ux = 2.0f * (q[1] * v[2] - q[2] * v[1]), uy = 2.0f * (q[2] * v[0] - q[0] * v[2]), uz = 2.0f * (q[0] * v[1] - q[1] * v[0]);
    o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}

__global__ __launch_bounds__(256) void k_ga_pts3d(bt_ga_args a, const float *mono_scaled, double *losses) {
    __shared__ float red[16];
    const int T = (int)a.T, N = (int)a.N, S = (int)a.S, mid = S / 2;
    const size_t total = (size_t)T * N * S, e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float val = 0.0f;
    if (e < total) {
        const int s = (int)(e % S), n = (int)((e / S) % N), t = (int)(e / ((size_t)S * N));
        const long long jraw = a.jj[(size_t)t * S + s];
        const int jc = (int)(jraw < 0 ? 0 : (jraw > T - 1 ? T - 1 : jraw));
        const bool half = a.half_disp != 0;
        const bool m = a.trajs_vis[e] > 0.9f && jraw >= 0 && jraw < T && ga_disp(a.trajs_disp, e, half) > 1e-2f && a.trajs_static[e] > 0.3f;
        if (m) {
            const size_t em = ((size_t)t * N + n) * S + mid;
            float src[3], trg[3], rel_t[3], tmp[3], from_src[3];
            ga_iproj(a.trajs_2d[2 * em], a.trajs_2d[2 * em + 1], mono_scaled[em], a.intrinsics + 4 * (size_t)t, src);
            ga_iproj(a.trajs_2d[2 * e], a.trajs_2d[2 * e + 1], mono_scaled[e], a.intrinsics + 4 * (size_t)jc, trg);
            // pose_j^-1 * pose_t acting on src:  R_j^T (R_t src + t_t - t_j)
            const float *pt = a.pose + 7 * (size_t)t, *pj = a.pose + 7 * (size_t)jc;
            ga_qrot(pt + 3, src, tmp);
            for (int c = 0; c < 3; ++c) rel_t[c] = tmp[c] + pt[c] - pj[c];
            const float qji[4] = {-pj[3], -pj[4], -pj[5], pj[6]};
            ga_qrot(qji, rel_t, from_src);
            const float dx = from_src[0] - trg[0], dy = from_src[1] - trg[1], dz = from_src[2] - trg[2];
            val = sqrtf(dx * dx + dy * dy + dz * dz);
        }
    }
    const float bs = block_sum(val, red);
    if (threadIdx.x == 0) atomicAdd(&losses[2], (double)bs / (double)total);
}

// ------------------------------------------------------------------ backward of  w_sp * spatial + w_rg * inter_frame
// with respect to trajs_scales [T,N,S] and frame_scales [T,gh,gw] (the parameters RefineNet.forward reaches,
// refine_net.py:252-293; trainer.py:23-77 steps them with Adam).  Three kernels behind bt_ga_backward:
//   k_ga_bwd_spatial   per (query frame, slot): d/d(mono_scaled) of the smooth-L1 term into g_ms, d/d(trajs_scales) through
//                      exp((p - mean_n p) / pw_break) incl. the mean's own derivative (g - mean_n g)
//   k_ga_bwd_pairwise  the O(Q S N^2) term: thread (q, s, n) walks ALL partners m (its own gradient needs every pair it is
//                      in; the forward's half walk would have to scatter to m) and adds d/d(mono_scaled) of its slot and,
//                      with the opposite sign, of the centre slot
//   k_ga_bwd_grid      g_ms -> d/d(frame_scales): the four bilinear cells of every track, summed per workgroup in LDS
//                      (all tracks of a (frame, slot) sample the same frame's grid), one global atomic per cell
__global__ __launch_bounds__(256) void k_ga_bwd_spatial(bt_ga_args a, const float *mono_scaled, float w_sp, float *g_ms, float *g_ts) {
    __shared__ float red[16];
    __shared__ float s_mean, s_gmean;
    const int S = (int)a.S, N = (int)a.N, T = (int)a.T;
    const int qi = blockIdx.x / S, s = blockIdx.x % S;
    const int t = (int)a.query[qi];
    const bool half = a.half_disp != 0;
    float part = 0.0f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) part += a.trajs_scales[((size_t)t * N + n) * S + s];
    const float tot = block_sum(part, red);
    if (threadIdx.x == 0) s_mean = tot / (float)N;
    __syncthreads();
    const float mean = s_mean;
    const long long jraw = a.jj[(size_t)t * S + s];
    const bool patch_ok = jraw >= 0 && jraw < T;
    const float c = w_sp / (float)((double)a.Q * N * S);
    float gsum = 0.0f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const size_t e = ((size_t)t * N + n) * S + s;
        const float x = a.trajs_2d[2 * e], y = a.trajs_2d[2 * e + 1];
        const float disp = ga_disp(a.trajs_disp, e, half), ms = mono_scaled[e];
        const float sexp = expf((a.trajs_scales[e] - mean) / a.pw_break);
        const float aligned = sexp * disp;
        const float diff = half ? __half2float(__hsub(__float2half(ms), __float2half(aligned))) : ms - aligned;
        const bool m = a.trajs_vis[e] > 0.9f && patch_ok && sqrtf(x * x + y * y) > 5.0f && disp > 1e-2f;
        const float h = m ? c * fminf(fmaxf(diff, -1.0f), 1.0f) : 0.0f;          // smooth-L1' = clamp(diff, -1, 1)
        g_ms[e] = h;
        const float gp = -h * aligned / a.pw_break;                              // d aligned / d p = aligned / pw_break
        g_ts[e] = gp;
        gsum += gp;
    }
    const float gt = block_sum(gsum, red);
    if (threadIdx.x == 0) s_gmean = gt / (float)N;
    __syncthreads();
    const float gmean = s_gmean;
    for (int n = threadIdx.x; n < N; n += blockDim.x) g_ts[((size_t)t * N + n) * S + s] -= gmean;
}

// d/d(points) of the masked |d_s - d_mid| of own track o with the partner's two tracks: cs = sign / d_s * (o_s - q_s) is what
// the pair adds to o's slot point (and, negated, to the partner's), cm = -sign / d_mid * (o_m - q_m) to o's centre point.
// 1 / distance by v_rsq on max(|.|^2, 1e-30): a zero distance has zero differences, its products vanish; sign(0) = 0 as
// torch.sign has it
__device__ __forceinline__ void ga_pair_grad(const GaOwn &o, const GaPartner &q, float wgt, f2 (&cs)[3], f2 (&cm)[3]) {
    const f2 dxs = o.xs - q.xs, dys = o.ys - q.ys, dzs = o.zs - q.zs;
    const f2 dxm = o.xm - q.xm, dym = o.ym - q.ym, dzm = o.zm - q.zm;
    const f2 ss = ga_fma2(dzs, dzs, ga_fma2(dys, dys, dxs * dxs)), sq = ga_fma2(dzm, dzm, ga_fma2(dym, dym, dxm * dxm));
    const f2 is = f2{__builtin_amdgcn_rsqf(fmaxf(ss.x, 1e-30f)), __builtin_amdgcn_rsqf(fmaxf(ss.y, 1e-30f))};
    const f2 im = f2{__builtin_amdgcn_rsqf(fmaxf(sq.x, 1e-30f)), __builtin_amdgcn_rsqf(fmaxf(sq.y, 1e-30f))};
    const f2 dd = ga_fma2(ss, is, -(sq * im));                                   // d_s - d_mid
    const f2 pv = q.vis * o.vis, pt = q.sta * o.sta;
    f2 sg;
    sg.x = (fminf(pv.x, pt.x) > 0.5f && dd.x != 0.0f) ? __builtin_copysignf(wgt, dd.x) : 0.0f;
    sg.y = (fminf(pv.y, pt.y) > 0.5f && dd.y != 0.0f) ? __builtin_copysignf(wgt, dd.y) : 0.0f;
    const f2 as = sg * is, am = -(sg * im);
    cs[0] = as * dxs; cs[1] = as * dys; cs[2] = as * dzs;
    cm[0] = am * dxm; cm[1] = am * dym; cm[2] = am * dzm;
}

// lane l <- lane (l + 1) mod 64
__device__ __forceinline__ float ga_rol1(float x) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0x134 /* wave_rol:1 */, 0xf, 0xf, false));
}

// The O(Q S N^2) term's gradient, every pair of tracks ONCE: what a pair adds to one of its tracks it takes from the other.
// A WAVE owns a block of 64 super-tracks (lane = super-track) and meets the blocks J = I + 1 .. I + Mb / 2 (mod Mb) — and its
// own — one after the other; inside a block pair lane l meets partner (l + t) mod 64 at step t, so the 12 sums of the
// PARTNER's tracks can live in registers too: they move one lane down per step (v_mov_b32_dpp wave_rol:1) and arrive, after
// the block's 64 steps, complete.  Only then — and for the own sums after the walk — does a lane turn the twelve numbers into
// d/d(mono_scaled) of the two tracks' slot and centre-slot disparities (a point is its ray times the depth: linear in the
// sums, so partial sums may leave separately) with four global float atomics, and into its share of the intrinsics'
// gradient.  (An LDS float atomic per step and value — the straightforward scatter — measured 25x slower than the whole
// walk, ≈250 cycles per wave instruction; per-track sums in LDS flushed per block pair cost the second workgroup per CU.
// Round 4: every thread walked all N partners — twice the pairs, one track pair per 40 B of LDS.)
#ifndef BT_GA_BWD_WAVES
#define BT_GA_BWD_WAVES 4
#endif
__global__ __launch_bounds__(kGaStrip, BT_GA_BWD_WAVES) void k_ga_bwd_pairwise(bt_ga_args a, const float *mono_scaled, float w_rg, float *g_ms, float *g_intr) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[16];
    const int T = (int)a.T, N = (int)a.N, S = (int)a.S, mid = S / 2, M = (N + 1) >> 1, Mb = (M + 63) >> 6, Mp = Mb << 6;
    const int qi = blockIdx.x / S, s = blockIdx.x % S;
    const int i = (int)a.query[qi];
    const long long jraw = a.jj[(size_t)i * S + s];
    if (jraw < 0 || jraw >= T || s == mid) return;
    GaSuper L;
    L.rec = reinterpret_cast<float4 *>(sm);                      // [Mp] (super-tracks M .. Mp - 1: tracks that do not exist)
    const long long jm = a.jj[(size_t)i * S + mid];
    const float *Ks = a.intrinsics + 4 * (size_t)jraw;
    const float *Km = a.intrinsics + 4 * (size_t)(jm < 0 ? 0 : (jm > T - 1 ? T - 1 : jm));
    ga_stage(a, mono_scaled, i, s, mid, Ks, Km, L, Mp);
    __syncthreads();
    // each pair sits twice in the mean over [S, N, N]
    const float c = 2.0f * w_rg / (float)((double)a.Q * S * N * N);
    const int lane = threadIdx.x & 63, I = blockIdx.y * (kGaStrip / 64) + (threadIdx.x >> 6);
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // the sums acc (slot x y z, centre x y z, each for the two tracks) of super-track p, times sgn, leave the workgroup
    auto emit = [&](int p, const float (&acc)[12], float sgn) {
        const float4 *r = L.rec + kGaRec * p;
        const float4 qa = r[0], qb = r[1], qc = r[2], ok = r[4];
        const float k = -c * sgn;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float psx = h ? qa.y : qa.x, psy = h ? qa.w : qa.z, psz = h ? qb.y : qb.x;
            const float pmx = h ? qb.w : qb.z, pmy = h ? qc.y : qc.x, pmz = h ? qc.w : qc.z;
            const float gs0 = k * acc[h], gs1 = k * acc[2 + h], gs2 = k * acc[4 + h], gm0 = k * acc[6 + h], gm1 = k * acc[8 + h], gm2 = k * acc[10 + h];
            const size_t es = ((size_t)i * N + (2 * p + h)) * S + s, em = ((size_t)i * N + (2 * p + h)) * S + mid;
            // depth = 1 / max(disparity, 1e-2): d P / d d = -P D above the clamp (the factor -1 is in k)
            if ((h ? ok.y : ok.x) != 0.0f) atomicAdd(&g_ms[es], (gs0 * psx + gs1 * psy + gs2 * psz) * psz);
            if ((h ? ok.w : ok.z) != 0.0f) atomicAdd(&g_ms[em], (gm0 * pmx + gm1 * pmy + gm2 * pmz) * pmz);
            if (g_intr) {
                v[0] += gs0 * psx; v[1] += gs1 * psy; v[2] += gs0 * psz; v[3] += gs1 * psz;
                v[4] += gm0 * pmx; v[5] += gm1 * pmy; v[6] += gm0 * pmz; v[7] += gm1 * pmz;
            }
        }
    };
    if (I < Mb) {
        const GaPartner me = ga_partner(L.rec + kGaRec * (I * 64 + lane));
        const GaOwn oa = ga_own<0>(me), ob = ga_own<1>(me);
        f2 gas[3], gam[3], gbs[3], gbm[3], cs[3], cm[3], ds[3], dm[3];
        float pacc[12];
        // the pair inside the super-track (lane x of the result is the track with itself: zero)
        ga_pair_grad(oa, me, 1.0f, cs, cm);
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
            gas[c3] = cs[c3]; gam[c3] = cm[c3];
            gbs[c3] = f2{-cs[c3].y, 0.0f}; gbm[c3] = f2{-cm[c3].y, 0.0f};
        }
        // block J, steps t0 .. t1 (lane l meets super-track 64 J + (l + t) mod 64), the last one with weight w_last
        auto block_pair = [&](int J, int t0, int t1, float w_last) {
#pragma unroll
            for (int k = 0; k < 12; ++k) pacc[k] = 0.0f;
            const float4 *rb = L.rec + kGaRec * 64 * J;
#pragma unroll 1
            for (int t = t0; t <= t1; ++t) {
                const GaPartner q = ga_partner(rb + kGaRec * ((lane + t) & 63));
                const float wgt = t == t1 ? w_last : 1.0f;
                ga_pair_grad(oa, q, wgt, cs, cm);
                ga_pair_grad(ob, q, wgt, ds, dm);
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3) {
                    gas[c3] += cs[c3]; gam[c3] += cm[c3]; gbs[c3] += ds[c3]; gbm[c3] += dm[c3];
                    const f2 ts = cs[c3] + ds[c3], tm = cm[c3] + dm[c3];
                    // the partner's sums came from the lane above (which met this partner one step ago)
                    pacc[2 * c3] = ga_rol1(pacc[2 * c3]) + ts.x;         pacc[2 * c3 + 1] = ga_rol1(pacc[2 * c3 + 1]) + ts.y;
                    pacc[6 + 2 * c3] = ga_rol1(pacc[6 + 2 * c3]) + tm.x; pacc[6 + 2 * c3 + 1] = ga_rol1(pacc[6 + 2 * c3 + 1]) + tm.y;
                }
            }
            emit(64 * J + ((lane + t1) & 63), pacc, -1.0f);                 // (the pair's vector, negated, is the partner's)
        };
        block_pair(I, 1, 32, 0.5f);                              // inside the block: the antipodal lane is met from both ends
        const int nfull = (Mb & 1) ? Mb >> 1 : (Mb >> 1) - 1;
        int J = I;
        for (int d = 0; d < nfull; ++d) {
            J = J + 1 >= Mb ? 0 : J + 1;
            block_pair(J, 0, 63, 1.0f);
        }
        if (!(Mb & 1) && I < (Mb >> 1)) block_pair(I + (Mb >> 1), 0, 63, 1.0f);       // the antipodal block: by the lower wave of the two
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
            pacc[2 * c3] = gas[c3].x + gas[c3].y;     pacc[2 * c3 + 1] = gbs[c3].x + gbs[c3].y;
            pacc[6 + 2 * c3] = gam[c3].x + gam[c3].y; pacc[6 + 2 * c3 + 1] = gbm[c3].x + gbm[c3].y;
        }
        emit(I * 64 + lane, pacc, 1.0f);
    }
    if (g_intr) {
        // the points' gradients through iproj to the intrinsics of the slot's frame and of the centre slot's frame
        // (refine_intrinsics: RefineNet.intrinsics is K * K_scale for every frame, refine_net.py:131-136): dX/dfx = -X / fx,
        // dX/dcx = -D / fx, dY/dfy = -Y / fy, dY/dcy = -D / fy   (the factor -c is in v already)
        const size_t js = (size_t)jraw, jmc = (size_t)(jm < 0 ? 0 : (jm > T - 1 ? T - 1 : jm));
        for (int k = 0; k < 8; ++k) {
            const float tsum = block_sum(v[k], red);
            if (threadIdx.x == 0) atomicAdd(&g_intr[4 * (k < 4 ? js : jmc) + (k & 3)], tsum / (k < 4 ? Ks : Km)[k & 1]);
        }
    }
}

__global__ __launch_bounds__(256) void k_ga_bwd_grid(bt_ga_args a, const float *g_ms, float *g_fs, int all_frames) {
    extern __shared__ __attribute__((aligned(16))) float cells[];
    const int S = (int)a.S, N = (int)a.N, T = (int)a.T, gh = (int)a.gh, gw = (int)a.gw;
    // all_frames: one workgroup per (t, s) of every frame (pts_3d_loss reaches them all), else per (query frame, s)
    const int qi = blockIdx.x / S, s = blockIdx.x % S;
    const int t = all_frames ? qi : (int)a.query[qi];
    const long long jraw = a.jj[(size_t)t * S + s];
    const int jc = (int)(jraw < 0 ? 0 : (jraw > T - 1 ? T - 1 : jraw));
    const float *grid = a.frame_scales + (size_t)jc * gh * gw;
    const bool half = a.half_disp != 0;
    for (int k = threadIdx.x; k < gh * gw; k += blockDim.x) cells[k] = 0.0f;
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const size_t e = ((size_t)t * N + n) * S + s;
        const float g = g_ms[e] * ga_disp(a.trajs_disp_mono, e, half);
        if (g == 0.0f) continue;
        const float x = a.trajs_2d[2 * e], y = a.trajs_2d[2 * e + 1];
        const float gx = x / (float)(a.W - 1) * (float)(gw - 1), gy = y / (float)(a.H - 1) * (float)(gh - 1);
        const float fx0 = floorf(gx), fy0 = floorf(gy);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float wx = gx - fx0, wy = gy - fy0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int yy = y0 + dy, xx = x0 + dx;
                if (yy >= 0 && yy < gh && xx >= 0 && xx < gw)
                    atomicAdd(&cells[yy * gw + xx], g * (dy ? wy : 1.0f - wy) * (dx ? wx : 1.0f - wx));
            }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < gh * gw; k += blockDim.x) {
        const float v = cells[k];
        if (v != 0.0f) atomicAdd(&g_fs[(size_t)jc * gh * gw + k], v * expf(grid[k] / 10.0f) / 10.0f);   // d exp(g / 10) / d g
    }
}


// ------------------------------------------------------------------ the terms loss_weight_dict adds (refine_net.py:274-297)
// cam_smooth_vec_loss (:356-360): mean_t |t_t - t_{t+1}| + 0.3 mean_t |q_t - q_{t+1}| over the RAW pose components;
// scale_grid_smoothness_loss (:362-392): mean over horizontal neighbours of f(s_a - s_b) + the same over vertical ones,
// s = exp(grid / 10), f = | | (l1), square (l2) or smooth-L1 (huber).  One workgroup: T is a few hundred, the grid 4..10 wide.
__device__ __forceinline__ float ga_smooth_f(float d, int mode) {
    const float ad = fabsf(d);
    if (mode == BT_GA_SMOOTH_L2) return d * d;
    if (mode == BT_GA_SMOOTH_HUBER) return ad < 1.0f ? 0.5f * d * d : ad - 0.5f;
    return ad;
}
__device__ __forceinline__ float ga_smooth_df(float d, int mode) {
    if (mode == BT_GA_SMOOTH_L2) return 2.0f * d;
    if (mode == BT_GA_SMOOTH_HUBER) return fminf(fmaxf(d, -1.0f), 1.0f);
    return d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
}

// G = false: losses[3] (camera smoothness), losses[4] (scale-grid smoothness).  G = true: their gradients, weighted, ADDED to
// g_pose [T,7] and g_fs [T,gh,gw] (plain adds: one workgroup, every element has one writer per phase).
template <bool G>
__global__ __launch_bounds__(256) void k_ga_smooth(bt_ga_args a, int mode, double *losses, float w_cam, float w_ss, float *g_pose, float *g_fs) {
    __shared__ float red[16];
    const int T = (int)a.T, gh = (int)a.gh, gw = (int)a.gw;
    // ---- camera smoothness
    if (!G || w_cam != 0.0f) {
        float lt = 0.0f, lr = 0.0f;
        for (int t = threadIdx.x; t + 1 < T; t += blockDim.x) {
            const float *p = a.pose + 7 * (size_t)t, *q = p + 7;
            float dt[3], dr[4], nt = 0.0f, nr = 0.0f;
            for (int c = 0; c < 3; ++c) { dt[c] = p[c] - q[c]; nt += dt[c] * dt[c]; }
            for (int c = 0; c < 4; ++c) { dr[c] = p[3 + c] - q[3 + c]; nr += dr[c] * dr[c]; }
            nt = sqrtf(nt); nr = sqrtf(nr);
            lt += nt; lr += nr;
            if (G) {                                             // d |v| / d v = v / |v| (0 at 0, as torch.norm)
                const float ct = nt > 0.0f ? w_cam / ((float)(T - 1) * nt) : 0.0f, cr = nr > 0.0f ? 0.3f * w_cam / ((float)(T - 1) * nr) : 0.0f;
                for (int c = 0; c < 3; ++c) { atomicAdd(&g_pose[7 * (size_t)t + c], ct * dt[c]); atomicAdd(&g_pose[7 * (size_t)(t + 1) + c], -ct * dt[c]); }
                for (int c = 0; c < 4; ++c) { atomicAdd(&g_pose[7 * (size_t)t + 3 + c], cr * dr[c]); atomicAdd(&g_pose[7 * (size_t)(t + 1) + 3 + c], -cr * dr[c]); }
            }
        }
        if (!G) {
            const float st = block_sum(lt, red), sr = block_sum(lr, red);
            if (threadIdx.x == 0 && T > 1) losses[3] = (double)st / (T - 1) + 0.3 * (double)sr / (T - 1);
        }
    }
    // ---- scale-grid smoothness
    if (!G || w_ss != 0.0f) {
        const int nh = T * gh * (gw - 1), nv = T * (gh - 1) * gw;
        float lh = 0.0f, lv = 0.0f;
        for (int k = threadIdx.x; k < nh; k += blockDim.x) {
            const int x = k % (gw - 1), y = (k / (gw - 1)) % gh, t = k / ((gw - 1) * gh);
            const size_t i0 = ((size_t)t * gh + y) * gw + x, i1 = i0 + 1;
            const float s0 = expf(a.frame_scales[i0] / 10.0f), s1 = expf(a.frame_scales[i1] / 10.0f), d = s0 - s1;
            lh += ga_smooth_f(d, mode);
            if (G) { const float g = w_ss * ga_smooth_df(d, mode) / (float)nh; atomicAdd(&g_fs[i0], g * s0 / 10.0f); atomicAdd(&g_fs[i1], -g * s1 / 10.0f); }
        }
        for (int k = threadIdx.x; k < nv; k += blockDim.x) {
            const int x = k % gw, y = (k / gw) % (gh - 1), t = k / (gw * (gh - 1));
            const size_t i0 = ((size_t)t * gh + y) * gw + x, i1 = i0 + gw;
            const float s0 = expf(a.frame_scales[i0] / 10.0f), s1 = expf(a.frame_scales[i1] / 10.0f), d = s0 - s1;
            lv += ga_smooth_f(d, mode);
            if (G) { const float g = w_ss * ga_smooth_df(d, mode) / (float)nv; atomicAdd(&g_fs[i0], g * s0 / 10.0f); atomicAdd(&g_fs[i1], -g * s1 / 10.0f); }
        }
        if (!G) {
            const float sh = block_sum(lh, red), sv = block_sum(lv, red);
            if (threadIdx.x == 0) losses[4] = (nh > 0 ? (double)sh / nh : 0.0) + (nv > 0 ? (double)sv / nv : 0.0);
        }
    }
}

// ------------------------------------------------------------------ backward of pts_3d_loss (refine_net.py:314-354)
// One workgroup per (frame t, slot s), threads over the tracks n: the frame pair (t, j), its poses and intrinsics are
// workgroup-uniform, so their gradients are summed in the workgroup and leave as one atomic each.
//   q = R_j^T (R_t p_src + t_t - t_j),  loss += w |q - p_trg| / (T N S)
// Gradients: d/d mono_scaled of the source (centre-slot) and of the target disparity (into g_ms, then through k_ga_bwd_grid
// to frame_scales_), d/d intrinsics of frames t and j (g_intr [T,4]), and d/d pose in the convention of pypose's own
// backward (pypose/lietensor/operation.py: SE3_Act / SE3_Mul / SE3_Inv return the gradient of the LEFT perturbation
// Exp(delta) X, order (tau, phi), padded with a zero to the 7 stored numbers): with Z = T_j^-1 T_t and g the gradient at q,
//   g_Z = (g, q x g),   g_{T_t} = g_Z Ad(T_j^-1) = -g_{T_j}.
__global__ __launch_bounds__(256) void k_ga_bwd_pts3d(bt_ga_args a, const float *mono_scaled, float w, float *g_ms, float *g_pose, float *g_intr) {
    __shared__ float red[16];
    const int T = (int)a.T, N = (int)a.N, S = (int)a.S, mid = S / 2;
    const int t = blockIdx.x / S, s = blockIdx.x % S;
    const long long jraw = a.jj[(size_t)t * S + s];
    if (jraw < 0 || jraw >= T) return;                              // patch_mask
    const int jc = (int)jraw;
    const bool half = a.half_disp != 0;
    const float *Kt = a.intrinsics + 4 * (size_t)t, *Kj = a.intrinsics + 4 * (size_t)jc;
    const float *pt = a.pose + 7 * (size_t)t, *pj = a.pose + 7 * (size_t)jc;
    const float qji[4] = {-pj[3], -pj[4], -pj[5], pj[6]};
    const float c = w / (float)((double)T * N * S);
    float gz[6] = {0, 0, 0, 0, 0, 0}, gkt[4] = {0, 0, 0, 0}, gkj[4] = {0, 0, 0, 0};
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const size_t e = ((size_t)t * N + n) * S + s, em = ((size_t)t * N + n) * S + mid;
        const bool m = a.trajs_vis[e] > 0.9f && ga_disp(a.trajs_disp, e, half) > 1e-2f && a.trajs_static[e] > 0.3f;
        if (!m) continue;
        float src[3], trg[3], tmp[3], rel_t[3], q[3];
        const float xs = a.trajs_2d[2 * em], ys = a.trajs_2d[2 * em + 1], xt = a.trajs_2d[2 * e], yt = a.trajs_2d[2 * e + 1];
        const float ds = mono_scaled[em], dt = mono_scaled[e];
        ga_iproj(xs, ys, ds, Kt, src);
        ga_iproj(xt, yt, dt, Kj, trg);
        ga_qrot(pt + 3, src, tmp);
        for (int k = 0; k < 3; ++k) rel_t[k] = tmp[k] + pt[k] - pj[k];
        ga_qrot(qji, rel_t, q);
        const float dx = q[0] - trg[0], dy = q[1] - trg[1], dz = q[2] - trg[2];
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
        if (!(nrm > 0.0f)) continue;
        const float g[3] = {c * dx / nrm, c * dy / nrm, c * dz / nrm};               // d loss / d q = - d loss / d p_trg
        // pose (tangent of Z): (g, q x g)
        gz[0] += g[0]; gz[1] += g[1]; gz[2] += g[2];
        gz[3] += q[1] * g[2] - q[2] * g[1]; gz[4] += q[2] * g[0] - q[0] * g[2]; gz[5] += q[0] * g[1] - q[1] * g[0];
        // source point: d q / d p_src = R_j^T R_t  ->  g_src = R_t^T R_j g
        float gs[3], t1[3];
        ga_qrot(pj + 3, g, t1);
        const float qti[4] = {-pt[3], -pt[4], -pt[5], pt[6]};
        ga_qrot(qti, t1, gs);
        // a point is (X, Y, D) = ((x - cx) / fx D, (y - cy) / fy D, D), D = 1 / max(d, 1e-2): d P / d d = -P D for d > 1e-2
        if (ds > 1e-2f) atomicAdd(&g_ms[em], -(gs[0] * src[0] + gs[1] * src[1] + gs[2] * src[2]) * src[2]);
        if (dt > 1e-2f) atomicAdd(&g_ms[e], (g[0] * trg[0] + g[1] * trg[1] + g[2] * trg[2]) * trg[2]);
        // intrinsics: dX/dfx = -X / fx, dX/dcx = -D / fx, dY/dfy = -Y / fy, dY/dcy = -D / fy
        gkt[0] += -gs[0] * src[0] / Kt[0]; gkt[1] += -gs[1] * src[1] / Kt[1]; gkt[2] += -gs[0] * src[2] / Kt[0]; gkt[3] += -gs[1] * src[2] / Kt[1];
        gkj[0] += g[0] * trg[0] / Kj[0];   gkj[1] += g[1] * trg[1] / Kj[1];   gkj[2] += g[0] * trg[2] / Kj[0];   gkj[3] += g[1] * trg[2] / Kj[1];
    }
    float tot[14];
    for (int k = 0; k < 6; ++k) tot[k] = block_sum(gz[k], red);
    for (int k = 0; k < 4; ++k) { tot[6 + k] = block_sum(gkt[k], red); tot[10 + k] = block_sum(gkj[k], red); }
    if (threadIdx.x == 0) {
        if (g_intr) for (int k = 0; k < 4; ++k) { atomicAdd(&g_intr[4 * (size_t)t + k], tot[6 + k]); atomicAdd(&g_intr[4 * (size_t)jc + k], tot[10 + k]); }
        if (g_pose && jc != t) {
            // g_Z Ad(A), A = T_j^-1 = (R_j^T, -R_j^T t_j):  (g_tau R_A, g_tau [t_A]x R_A + g_phi R_A);  v R_A = R_A^T v = R_j v
            float tA[3], gt[3], u[3], gp[3];
            const float ntj[3] = {-pj[0], -pj[1], -pj[2]};
            ga_qrot(qji, ntj, tA);                                   // t_A = -R_j^T t_j
            ga_qrot(pj + 3, tot, gt);                                // g_tau R_A
            // the row vector g_tau [t_A]x is (g_tau x t_A)^T  (v^T [t]x = ([t]x^T v)^T = (-t x v)^T = (v x t)^T)
            u[0] = tot[1] * tA[2] - tot[2] * tA[1]; u[1] = tot[2] * tA[0] - tot[0] * tA[2]; u[2] = tot[0] * tA[1] - tot[1] * tA[0];
            for (int k = 0; k < 3; ++k) u[k] += tot[3 + k];          // g_tau [t_A]x + g_phi
            ga_qrot(pj + 3, u, gp);
            for (int k = 0; k < 3; ++k) { atomicAdd(&g_pose[7 * (size_t)t + k], gt[k]); atomicAdd(&g_pose[7 * (size_t)jc + k], -gt[k]); }
            for (int k = 0; k < 3; ++k) { atomicAdd(&g_pose[7 * (size_t)t + 3 + k], gp[k]); atomicAdd(&g_pose[7 * (size_t)jc + 3 + k], -gp[k]); }
        }
    }
}
}  // namespace bt

extern "C" int bt_ga_backward_total(const bt_ga_args *a, const float *mono_scaled, const bt_ga_weights *w, float *g_mono_scaled,
                                    float *grad_trajs_scales, float *grad_frame_scales, float *grad_pose, float *grad_intrinsics, void *stream) {
    if (!a || !w || !mono_scaled || !g_mono_scaled || !grad_trajs_scales || !grad_frame_scales) return BT_EINVAL;
    if (a->T <= 0 || a->N <= 0 || a->S <= 0 || a->gh <= 0 || a->gw <= 0 || a->H <= 1 || a->W <= 1 || a->Q <= 0) return BT_EINVAL;
    if (!a->trajs_2d || !a->trajs_disp || !a->trajs_disp_mono || !a->trajs_vis || !a->trajs_static || !a->jj || !a->intrinsics ||
        !a->query || !a->trajs_scales || !a->frame_scales || !a->frame_shifts) return BT_EINVAL;
    if ((w->pts3d != 0.0f || w->cam_smooth != 0.0f) && !a->pose) return BT_EINVAL;
    if (w->smooth_mode < BT_GA_SMOOTH_L1 || w->smooth_mode > BT_GA_SMOOTH_HUBER) return BT_EINVAL;
    if (a->T * a->S > 0x7fffffff || a->T * a->N * a->S > ((int64_t)1 << 40) || a->gh * a->gw > 12 * 1024) return BT_EUNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t tns = (size_t)(a->T * a->N * a->S);
    if (hipMemsetAsync(g_mono_scaled, 0, tns * sizeof(float), st) != hipSuccess ||
        hipMemsetAsync(grad_trajs_scales, 0, tns * sizeof(float), st) != hipSuccess ||
        hipMemsetAsync(grad_frame_scales, 0, (size_t)(a->T * a->gh * a->gw) * sizeof(float), st) != hipSuccess ||
        (grad_pose && hipMemsetAsync(grad_pose, 0, (size_t)a->T * 7 * sizeof(float), st) != hipSuccess) ||
        (grad_intrinsics && hipMemsetAsync(grad_intrinsics, 0, (size_t)a->T * 4 * sizeof(float), st) != hipSuccess)) return BT_EHIP;
    const dim3 qs((unsigned)(a->Q * a->S)), ts((unsigned)(a->T * a->S));
    hipLaunchKernelGGL(bt::k_ga_bwd_spatial, qs, dim3(256), 0, st, *a, mono_scaled, w->spatial, g_mono_scaled, grad_trajs_scales);
    if (w->rigid != 0.0f) {
        const size_t M = (size_t)(a->N + 1) / 2, Mb = (M + 63) / 64, lds = Mb * 64 * bt::kGaRec * sizeof(float4);
        if (lds > 160 * 1024) return BT_EUNSUPPORTED;             // N <= 4096 tracks per frame
        static bt::LdsLimit lds_limit;
        if (!lds_limit.ensure(reinterpret_cast<const void *>(&bt::k_ga_bwd_pairwise), lds)) return BT_EHIP;
        hipLaunchKernelGGL(bt::k_ga_bwd_pairwise, dim3((unsigned)(a->Q * a->S), (unsigned)((Mb + bt::kGaStrip / 64 - 1) / (bt::kGaStrip / 64))), dim3(bt::kGaStrip),
                           lds, st, *a, mono_scaled, w->rigid, g_mono_scaled, grad_intrinsics);
    }
    if (w->pts3d != 0.0f)
        hipLaunchKernelGGL(bt::k_ga_bwd_pts3d, ts, dim3(256), 0, st, *a, mono_scaled, w->pts3d, g_mono_scaled, grad_pose, grad_intrinsics);
    // d total / d mono_scaled -> frame_scales_: over the query frames, or over every frame when the 3-D point term is in
    hipLaunchKernelGGL(bt::k_ga_bwd_grid, w->pts3d != 0.0f ? ts : qs, dim3(256), (size_t)(a->gh * a->gw) * sizeof(float), st, *a, g_mono_scaled,
                       grad_frame_scales, w->pts3d != 0.0f ? 1 : 0);
    if ((w->cam_smooth != 0.0f && grad_pose) || w->scale_smooth != 0.0f)
        hipLaunchKernelGGL(bt::k_ga_smooth<true>, dim3(1), dim3(256), 0, st, *a, w->smooth_mode, (double *)nullptr, grad_pose ? w->cam_smooth : 0.0f,
                           w->scale_smooth, grad_pose, grad_frame_scales);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

extern "C" int bt_ga_backward(const bt_ga_args *a, const float *mono_scaled, float w_spatial, float w_rigid, float *g_mono_scaled,
                              float *grad_trajs_scales, float *grad_frame_scales, void *stream) {
    const bt_ga_weights w = {w_spatial, w_rigid, 0.0f, 0.0f, 0.0f, BT_GA_SMOOTH_L1};
    return bt_ga_backward_total(a, mono_scaled, &w, g_mono_scaled, grad_trajs_scales, grad_frame_scales, nullptr, nullptr, stream);
}

// ------------------------------------------------------------------ hand-off from the sparse-SLAM stage (refine_net.py:53-121)
__global__ __launch_bounds__(256) void k_ga_mat_to_se3(const float *mats, float *poses, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const float *m = mats + (size_t)t * 16;
    const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[4], m11 = m[5], m12 = m[6], m20 = m[8], m21 = m[9], m22 = m[10];
    const float tr = m00 + m11 + m22;
    float qx, qy, qz, qw;
    if (tr > 0.0f) {
        const float sq = sqrtf(tr + 1.0f) * 2.0f;
        qw = 0.25f * sq; qx = (m21 - m12) / sq; qy = (m02 - m20) / sq; qz = (m10 - m01) / sq;
    } else if (m00 > m11 && m00 > m22) {
        const float sq = sqrtf(1.0f + m00 - m11 - m22) * 2.0f;
        qw = (m21 - m12) / sq; qx = 0.25f * sq; qy = (m01 + m10) / sq; qz = (m02 + m20) / sq;
    } else if (m11 > m22) {
        const float sq = sqrtf(1.0f + m11 - m00 - m22) * 2.0f;
        qw = (m02 - m20) / sq; qx = (m01 + m10) / sq; qy = 0.25f * sq; qz = (m12 + m21) / sq;
    } else {
        const float sq = sqrtf(1.0f + m22 - m00 - m11) * 2.0f;
        qw = (m10 - m01) / sq; qx = (m02 + m20) / sq; qy = (m12 + m21) / sq; qz = 0.25f * sq;
    }
    const float inv = 1.0f / sqrtf(qx * qx + qy * qy + qz * qz + qw * qw);
    float *o = poses + (size_t)t * 7;
    o[0] = m[3]; o[1] = m[7]; o[2] = m[11];
    o[3] = qx * inv; o[4] = qy * inv; o[5] = qz * inv; o[6] = qw * inv;
}

__global__ __launch_bounds__(256) void k_ga_sample_disp_mono(const float *dmaps, const float *trajs_2d, float *out, int T, int N, int S, int H, int W) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)T * N * S) return;
    const int s = (int)(idx % S), t = (int)(idx / ((long long)N * S));
    int f = t + s - S / 2;
    f = f < 0 ? 0 : (f > T - 1 ? T - 1 : f);
    const float x = trajs_2d[2 * idx], y = trajs_2d[2 * idx + 1];
    const int x0 = (int)floorf(x), y0 = (int)floorf(y), x1 = x0 + 1, y1 = y0 + 1;
    const int x0c = min(max(x0, 0), W - 1), x1c = min(max(x1, 0), W - 1), y0c = min(max(y0, 0), H - 1), y1c = min(max(y1, 0), H - 1);
    const float *im = dmaps + (size_t)f * H * W;
    const float i00 = im[(size_t)y0c * W + x0c], i01 = im[(size_t)y0c * W + x1c], i10 = im[(size_t)y1c * W + x0c], i11 = im[(size_t)y1c * W + x1c];
    const float x0f = (float)x0, x1f = (float)x1, y0f = (float)y0, y1f = (float)y1;
    const float w00 = (x1f - x) * (y1f - y), w01 = (x - x0f) * (y1f - y), w10 = (x1f - x) * (y - y0f), w11 = (x - x0f) * (y - y0f);
    const float depth = w00 * i00 + w01 * i01 + w10 * i10 + w11 * i11;
    out[idx] = 1.0f / fmaxf(depth, 1e-2f);
}

extern "C" int bt_ga_mat_to_se3(const float *mats, float *poses, int64_t T, void *stream) {
    if (!mats || !poses || T < 0 || T > (1 << 24)) return BT_EINVAL;
    if (T == 0) return BT_OK;
    hipLaunchKernelGGL(k_ga_mat_to_se3, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), mats, poses, (int)T);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

extern "C" int bt_ga_sample_disp_mono(const float *dmaps, const float *trajs_2d, float *out, int64_t T, int64_t N, int64_t S, int64_t H, int64_t W, void *stream) {
    if (!dmaps || !trajs_2d || !out || T < 1 || N < 1 || S < 1 || H < 1 || W < 1 || T > (1 << 24) || (double)T * (double)N * (double)S > 4e11) return BT_EINVAL;
    if (H > 0x7fffffffll || W > 0x7fffffffll || N > 0x7fffffffll || S > 0x7fffffffll || (double)T * (double)H * (double)W > 9e18) return BT_EINVAL;   // (the kernel's ints)
    const long long total = (long long)T * N * S;
    hipLaunchKernelGGL(k_ga_sample_disp_mono, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       dmaps, trajs_2d, out, (int)T, (int)N, (int)S, (int)H, (int)W);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

extern "C" int bt_ga_forward(const bt_ga_args *a, float *mono_scaled_out, double *losses, int32_t which, void *stream) {
    if (!a || !mono_scaled_out || !losses) return BT_EINVAL;
    if (a->T <= 0 || a->N <= 0 || a->S <= 0 || a->gh <= 0 || a->gw <= 0 || a->H <= 1 || a->W <= 1 || a->Q <= 0) return BT_EINVAL;
    if (!a->trajs_2d || !a->trajs_disp || !a->trajs_disp_mono || !a->trajs_vis || !a->trajs_static || !a->jj || !a->intrinsics ||
        !a->pose || !a->query || !a->trajs_scales || !a->frame_scales || !a->frame_shifts) return BT_EINVAL;
    if (a->T * a->S > 0x7fffffff || a->T * a->N * a->S > ((int64_t)1 << 40)) return BT_EUNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(losses, 0, 5 * sizeof(double), st) != hipSuccess) return BT_EHIP;
    hipLaunchKernelGGL(bt::k_ga_scale, dim3((unsigned)(a->T * a->S)), dim3(256), 0, st, *a, mono_scaled_out, losses);
    if (which & 2) {
        const size_t M = (size_t)(a->N + 1) / 2, lds = M * bt::kGaRec * sizeof(float4);
        if ((M + 63) / 64 * 64 * bt::kGaRec * sizeof(float4) > 160 * 1024) return BT_EUNSUPPORTED;      // N <= 4096 tracks per frame (the backward's staging: blocks of 64 track pairs)
        static bt::LdsLimit lds_limit;
        if (!lds_limit.ensure(reinterpret_cast<const void *>(&bt::k_ga_pairwise), lds)) return BT_EHIP;
        hipLaunchKernelGGL(bt::k_ga_pairwise, dim3((unsigned)(a->Q * a->S), (unsigned)((M + bt::kGaStrip - 1) / bt::kGaStrip)), dim3(bt::kGaStrip),
                           lds, st, *a, mono_scaled_out, losses);
    }
    if (which & 4) {
        const size_t total = (size_t)(a->T * a->N * a->S);
        hipLaunchKernelGGL(bt::k_ga_pts3d, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *a, mono_scaled_out, losses);
    }
    if (which & 8) {
        const int mode = (which >> 8) & 3;
        if (mode > BT_GA_SMOOTH_HUBER) return BT_EINVAL;
        hipLaunchKernelGGL(bt::k_ga_smooth<false>, dim3(1), dim3(256), 0, st, *a, mode, losses, 0.0f, 0.0f, (float *)nullptr, (float *)nullptr);
    }
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}
