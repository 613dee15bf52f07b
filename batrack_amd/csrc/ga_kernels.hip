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

constexpr int kGaStrip = 256;           // tracks n per workgroup of k_ga_pairwise

__global__ __launch_bounds__(kGaStrip) void k_ga_pairwise(bt_ga_args a, const float *mono_scaled, double *losses) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[16];
    const int T = (int)a.T, N = (int)a.N, S = (int)a.S, mid = S / 2;
    const int qi = blockIdx.x / S, s = blockIdx.x % S, n0 = blockIdx.y * kGaStrip;
    const int i = (int)a.query[qi];
    const long long jraw = a.jj[(size_t)i * S + s];
    if (jraw < 0 || jraw >= T || s == mid) return;             // t_mask, and the centre slot's own difference is zero
    // all N tracks of the slot and of the centre slot: point (3) | point at the centre slot (3) | vis, static, mono-ok
    float4 *Ps = reinterpret_cast<float4 *>(sm), *Pm = Ps + N;
    float2 *Vs = reinterpret_cast<float2 *>(Pm + N);
    const long long jm = a.jj[(size_t)i * S + mid];
    const float *Ks = a.intrinsics + 4 * (size_t)jraw;
    const float *Km = a.intrinsics + 4 * (size_t)(jm < 0 ? 0 : (jm > T - 1 ? T - 1 : jm));
    const bool half = a.half_disp != 0;
    for (int m = threadIdx.x; m < N; m += blockDim.x) {
        const size_t es = ((size_t)i * N + m) * S + s, em = ((size_t)i * N + m) * S + mid;
        float P[3];
        ga_iproj(a.trajs_2d[2 * es], a.trajs_2d[2 * es + 1], mono_scaled[es], Ks, P);
        const bool okd = ga_disp(a.trajs_disp_mono, es, half) > 1e-2f;
        Ps[m] = make_float4(P[0], P[1], P[2], okd ? 1.0f : 0.0f);
        ga_iproj(a.trajs_2d[2 * em], a.trajs_2d[2 * em + 1], mono_scaled[em], Km, P);
        Pm[m] = make_float4(P[0], P[1], P[2], 0.0f);
        Vs[m] = make_float2(a.trajs_vis[es], a.trajs_static[es]);
    }
    __syncthreads();
    const int n = n0 + threadIdx.x;
    float acc = 0.0f;
    if (n < N) {
        // |d_s(n, m) - d_mid(n, m)| is symmetric in (n, m) and zero on the diagonal: thread n takes the partners
        // m = n + 1 .. n + N / 2 (mod N) — every unordered pair once, the same trip count for every thread — and the sum
        // counts each pair twice (for even N the antipodal partner is met from both ends: half weight).
        const float4 ps = Ps[n], pm = Pm[n];
        const float2 vn = Vs[n];
        const int half_n = N >> 1;
        int m = n + 1 >= N ? n + 1 - N : n + 1;
#pragma unroll 4
        for (int k = 1; k <= half_n; ++k) {
            const float4 qs = Ps[m], qm = Pm[m];
            const float2 vm = Vs[m];
            const float dxs = ps.x - qs.x, dys = ps.y - qs.y, dzs = ps.z - qs.z;
            const float dxm = pm.x - qm.x, dym = pm.y - qm.y, dzm = pm.z - qm.z;
            // hardware square roots (1 ulp): the IEEE-exact sequence is ten instructions per root, two roots per pair
            const float ds = __builtin_amdgcn_sqrtf(dxs * dxs + dys * dys + dzs * dzs), dm = __builtin_amdgcn_sqrtf(dxm * dxm + dym * dym + dzm * dzm);
            const bool mk = vn.x * vm.x > 0.5f && vn.y * vm.y > 0.5f && ps.w * qs.w > 0.5f;
            const float wgt = (2 * k == N) ? 1.0f : 2.0f;
            acc += mk ? wgt * fabsf(ds - dm) : 0.0f;
            m = m + 1 >= N ? 0 : m + 1;
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

__global__ __launch_bounds__(kGaStrip) void k_ga_bwd_pairwise(bt_ga_args a, const float *mono_scaled, float w_rg, float *g_ms) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int T = (int)a.T, N = (int)a.N, S = (int)a.S, mid = S / 2;
    const int qi = blockIdx.x / S, s = blockIdx.x % S, n0 = blockIdx.y * kGaStrip;
    const int i = (int)a.query[qi];
    const long long jraw = a.jj[(size_t)i * S + s];
    if (jraw < 0 || jraw >= T || s == mid) return;
    float4 *Ps = reinterpret_cast<float4 *>(sm), *Pm = Ps + N;
    float2 *Vs = reinterpret_cast<float2 *>(Pm + N);
    const long long jm = a.jj[(size_t)i * S + mid];
    const float *Ks = a.intrinsics + 4 * (size_t)jraw;
    const float *Km = a.intrinsics + 4 * (size_t)(jm < 0 ? 0 : (jm > T - 1 ? T - 1 : jm));
    const bool half = a.half_disp != 0;
    for (int m = threadIdx.x; m < N; m += blockDim.x) {
        const size_t es = ((size_t)i * N + m) * S + s, em = ((size_t)i * N + m) * S + mid;
        float P[3];
        ga_iproj(a.trajs_2d[2 * es], a.trajs_2d[2 * es + 1], mono_scaled[es], Ks, P);
        const bool okd = ga_disp(a.trajs_disp_mono, es, half) > 1e-2f;
        Ps[m] = make_float4(P[0], P[1], P[2], okd ? 1.0f : 0.0f);
        ga_iproj(a.trajs_2d[2 * em], a.trajs_2d[2 * em + 1], mono_scaled[em], Km, P);
        Pm[m] = make_float4(P[0], P[1], P[2], 0.0f);
        Vs[m] = make_float2(a.trajs_vis[es], a.trajs_static[es]);
    }
    __syncthreads();
    const int n = n0 + threadIdx.x;
    if (n >= N) return;
    const float4 ps = Ps[n], pm = Pm[n];
    const float2 vn = Vs[n];
    float gs0 = 0.0f, gs1 = 0.0f, gs2 = 0.0f, gm0 = 0.0f, gm1 = 0.0f, gm2 = 0.0f;
#pragma unroll 4
    for (int m = 0; m < N; ++m) {
        const float4 qs = Ps[m], qm = Pm[m];
        const float2 vm = Vs[m];
        const float dxs = ps.x - qs.x, dys = ps.y - qs.y, dzs = ps.z - qs.z;
        const float dxm = pm.x - qm.x, dym = pm.y - qm.y, dzm = pm.z - qm.z;
        const float ss = dxs * dxs + dys * dys + dzs * dzs, sq = dxm * dxm + dym * dym + dzm * dzm;
        const float is = ss > 0.0f ? __builtin_amdgcn_rsqf(ss) : 0.0f, im = sq > 0.0f ? __builtin_amdgcn_rsqf(sq) : 0.0f;   // 1 / distance (0 at distance 0)
        const float dd = ss * is - sq * im;                                      // d_s - d_mid
        const bool mk = vn.x * vm.x > 0.5f && vn.y * vm.y > 0.5f && ps.w * qs.w > 0.5f;
        const float sg = mk ? (dd > 0.0f ? 1.0f : (dd < 0.0f ? -1.0f : 0.0f)) : 0.0f;
        const float as = sg * is, am = sg * im;
        gs0 += as * dxs; gs1 += as * dys; gs2 += as * dzs;
        gm0 -= am * dxm; gm1 -= am * dym; gm2 -= am * dzm;
    }
    // each pair sits twice in the mean over [S, N, N]; a point is its ray times the depth, depth = 1 / max(disparity, 1e-2)
    const float c = 2.0f * w_rg / (float)((double)a.Q * S * N * N);
    const size_t es = ((size_t)i * N + n) * S + s, em = ((size_t)i * N + n) * S + mid;
    if (mono_scaled[es] > 1e-2f) atomicAdd(&g_ms[es], -c * (gs0 * ps.x + gs1 * ps.y + gs2 * ps.z) * ps.z);
    if (mono_scaled[em] > 1e-2f) atomicAdd(&g_ms[em], -c * (gm0 * pm.x + gm1 * pm.y + gm2 * pm.z) * pm.z);
}

__global__ __launch_bounds__(256) void k_ga_bwd_grid(bt_ga_args a, const float *g_ms, float *g_fs) {
    extern __shared__ __attribute__((aligned(16))) float cells[];
    const int S = (int)a.S, N = (int)a.N, T = (int)a.T, gh = (int)a.gh, gw = (int)a.gw;
    const int qi = blockIdx.x / S, s = blockIdx.x % S;
    const int t = (int)a.query[qi];
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

}  // namespace bt

extern "C" int bt_ga_backward(const bt_ga_args *a, const float *mono_scaled, float w_spatial, float w_rigid, float *g_mono_scaled,
                              float *grad_trajs_scales, float *grad_frame_scales, void *stream) {
    if (!a || !mono_scaled || !g_mono_scaled || !grad_trajs_scales || !grad_frame_scales) return BT_EINVAL;
    if (a->T <= 0 || a->N <= 0 || a->S <= 0 || a->gh <= 0 || a->gw <= 0 || a->H <= 1 || a->W <= 1 || a->Q <= 0) return BT_EINVAL;
    if (!a->trajs_2d || !a->trajs_disp || !a->trajs_disp_mono || !a->trajs_vis || !a->trajs_static || !a->jj || !a->intrinsics ||
        !a->query || !a->trajs_scales || !a->frame_scales || !a->frame_shifts) return BT_EINVAL;
    if (a->T * a->S > 0x7fffffff || a->T * a->N * a->S > ((int64_t)1 << 40) || a->gh * a->gw > 12 * 1024) return BT_EUNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t tns = (size_t)(a->T * a->N * a->S);
    if (hipMemsetAsync(g_mono_scaled, 0, tns * sizeof(float), st) != hipSuccess ||
        hipMemsetAsync(grad_trajs_scales, 0, tns * sizeof(float), st) != hipSuccess ||
        hipMemsetAsync(grad_frame_scales, 0, (size_t)(a->T * a->gh * a->gw) * sizeof(float), st) != hipSuccess) return BT_EHIP;
    const dim3 qs((unsigned)(a->Q * a->S));
    hipLaunchKernelGGL(bt::k_ga_bwd_spatial, qs, dim3(256), 0, st, *a, mono_scaled, w_spatial, g_mono_scaled, grad_trajs_scales);
    if (w_rigid != 0.0f) {
        const size_t lds = (size_t)a->N * (2 * sizeof(float4) + sizeof(float2));
        if (lds > 64 * 1024) return BT_EUNSUPPORTED;
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(&bt::k_ga_bwd_pairwise), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess)
            return BT_EHIP;
        hipLaunchKernelGGL(bt::k_ga_bwd_pairwise, dim3((unsigned)(a->Q * a->S), (unsigned)((a->N + bt::kGaStrip - 1) / bt::kGaStrip)), dim3(bt::kGaStrip),
                           lds, st, *a, mono_scaled, w_rigid, g_mono_scaled);
    }
    hipLaunchKernelGGL(bt::k_ga_bwd_grid, qs, dim3(256), (size_t)(a->gh * a->gw) * sizeof(float), st, *a, g_mono_scaled, grad_frame_scales);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

extern "C" int bt_ga_forward(const bt_ga_args *a, float *mono_scaled_out, double *losses, int32_t which, void *stream) {
    if (!a || !mono_scaled_out || !losses) return BT_EINVAL;
    if (a->T <= 0 || a->N <= 0 || a->S <= 0 || a->gh <= 0 || a->gw <= 0 || a->H <= 1 || a->W <= 1 || a->Q <= 0) return BT_EINVAL;
    if (!a->trajs_2d || !a->trajs_disp || !a->trajs_disp_mono || !a->trajs_vis || !a->trajs_static || !a->jj || !a->intrinsics ||
        !a->pose || !a->query || !a->trajs_scales || !a->frame_scales || !a->frame_shifts) return BT_EINVAL;
    if (a->T * a->S > 0x7fffffff || a->T * a->N * a->S > ((int64_t)1 << 40)) return BT_EUNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(losses, 0, 3 * sizeof(double), st) != hipSuccess) return BT_EHIP;
    hipLaunchKernelGGL(bt::k_ga_scale, dim3((unsigned)(a->T * a->S)), dim3(256), 0, st, *a, mono_scaled_out, losses);
    if (which & 2) {
        const size_t lds = (size_t)a->N * (2 * sizeof(float4) + sizeof(float2));
        if (lds > 64 * 1024) return BT_EUNSUPPORTED;              // N <= 1638 tracks per frame
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(&bt::k_ga_pairwise), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess)
            return BT_EHIP;
        hipLaunchKernelGGL(bt::k_ga_pairwise, dim3((unsigned)(a->Q * a->S), (unsigned)((a->N + bt::kGaStrip - 1) / bt::kGaStrip)), dim3(bt::kGaStrip),
                           lds, st, *a, mono_scaled_out, losses);
    }
    if (which & 4) {
        const size_t total = (size_t)(a->T * a->N * a->S);
        hipLaunchKernelGGL(bt::k_ga_pts3d, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *a, mono_scaled_out, losses);
    }
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}
