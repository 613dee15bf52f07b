// ba_dense.hip — the reduced camera system of MORE THAN 255 free poses (plans with `wide` set, ba_plan.cpp): a dense, blocked,
// right-looking Cholesky in double in the global workspace, then the two triangular solves.  The block-sparse solvers
// (ba_kernels.hip) keep pose numbers in 8 bits and live in one CU's LDS; the reference's solve (ba.py:60-70: torch.linalg.cholesky
// of the dense 6n x 6n matrix) has no size clause, so neither may this backend — a global / loop-closing adjustment of 300 or 1000
// keyframes must step, if at a fraction of the windowed solver's rate.  Slow-but-correct by design: one kernel per phase and
// panel (3 launches per 48 columns), plain LDS-tiled products on the vector pipe.
//   damping    diag += ep + lm * diag, lm = 1e-4                                   (ba.py:67)
//   failure    a non-positive pivot => dX = 0                                      (ba.py:9-13)
//   NaN in dX  => once more with lm = 1e-3                                         (ba.py:324-325)
// The second attempt's kernels are enqueued with the first's (nothing here waits for the host) and return at once unless the
// first left the retry flag.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>

#include "ba_kernels.hpp"
#include "dev_cache.hpp"

namespace bt {
namespace dn {

constexpr int NB = 48;                       // panel width: 8 poses
constexpr int kCtlFail = 0, kCtlRetry = 1;   // ints behind the status words (StepArgs::status + kCtlOffset)
constexpr int kCtlOffset = 200;

struct Dense { double *L, *z; int *ctl; int D; };

__device__ __forceinline__ bool skip(const Dense &d, int attempt) {
    // attempt 1 runs only if attempt 0 asked for it; nobody runs after a failed factorisation of the current attempt
    return attempt == 1 && d.ctl[kCtlRetry] == 0;
}

// L = lower(S) with the damped diagonal, z = y; the flags of this attempt
__global__ __launch_bounds__(256) void k_dn_load(Dense d, const double *S, const double *y, float ep, int attempt) {
    if (skip(d, attempt)) return;
    const double lm = attempt == 0 ? 1e-4 : 1e-3;
    const size_t D = (size_t)d.D, total = D * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / D, c = i - r * D;
        if (c > r) continue;
        double v = S[i];
        if (c == r) v = v + ((double)ep + lm * v);
        d.L[i] = v;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < D; i += (size_t)gridDim.x * blockDim.x) d.z[i] = y[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) d.ctl[kCtlFail] = 0;
}

// the diagonal block of panel p, factored in LDS
__global__ __launch_bounds__(256) void k_dn_panel(Dense d, int p0, int attempt) {
    if (skip(d, attempt) || d.ctl[kCtlFail]) return;
    __shared__ double A[NB][NB + 1];
    __shared__ int bad;
    const int tid = threadIdx.x, nb = min(NB, d.D - p0);
    if (tid == 0) bad = 0;
    for (int i = tid; i < nb * nb; i += 256) { const int r = i / nb, c = i - r * nb; A[r][c] = c <= r ? d.L[(size_t)(p0 + r) * d.D + p0 + c] : 0.0; }
    __syncthreads();
    for (int k = 0; k < nb; ++k) {
        const double piv = A[k][k];
        if (!(piv > 0.0)) { if (tid == 0) bad = 1; break; }          // (every thread sees the same pivot: a uniform exit)
        const double il = 1.0 / sqrt(piv);
        __syncthreads();
        if (tid == 0) A[k][k] = sqrt(piv);
        for (int r = k + 1 + tid; r < nb; r += 256) A[r][k] *= il;
        __syncthreads();
        const int m = nb - k - 1;                                      // trailing block: rows, cols k+1 .. nb-1, lower part
        for (int i = tid; i < m * m; i += 256) {
            const int r = k + 1 + i / m, c = k + 1 + i % m;
            if (c <= r) A[r][c] -= A[r][k] * A[c][k];
        }
        __syncthreads();
    }
    __syncthreads();
    if (bad) { if (tid == 0) d.ctl[kCtlFail] = 1; return; }
    for (int i = tid; i < nb * nb; i += 256) { const int r = i / nb, c = i - r * nb; if (c <= r) d.L[(size_t)(p0 + r) * d.D + p0 + c] = A[r][c]; }
}

// the rows below the panel: L[r, panel] = A[r, panel] L_pp^-T, one thread per row
__global__ __launch_bounds__(256) void k_dn_trsm(Dense d, int p0, int attempt) {
    if (skip(d, attempt) || d.ctl[kCtlFail]) return;
    __shared__ double Lp[NB][NB + 1];
    const int tid = threadIdx.x, nb = min(NB, d.D - p0), p1 = p0 + nb;
    for (int i = tid; i < nb * nb; i += 256) { const int r = i / nb, c = i - r * nb; Lp[r][c] = c <= r ? d.L[(size_t)(p0 + r) * d.D + p0 + c] : 0.0; }
    __syncthreads();
    const int r = p1 + blockIdx.x * 256 + tid;
    if (r >= d.D) return;
    double *row = d.L + (size_t)r * d.D + p0;
    double x[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) x[c] = c < nb ? row[c] : 0.0;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        if (c < nb) {
            double t = x[c];
#pragma unroll
            for (int k = 0; k < c; ++k) t -= x[k] * Lp[c][k];
            x[c] = t / Lp[c][c];
        }
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) if (c < nb) row[c] = x[c];
}

// the trailing matrix: A[bi, bj] -= L[bi, p] L[bj, p]^T over the lower block tiles behind the panel
__global__ __launch_bounds__(256) void k_dn_syrk(Dense d, int p0, int attempt) {
    if (skip(d, attempt) || d.ctl[kCtlFail]) return;
    __shared__ double P1[NB][NB + 1], P2[NB][NB + 1];
    const int tid = threadIdx.x, nb = min(NB, d.D - p0), p1 = p0 + nb;
    // linear tile index -> (bi >= bj), rows / columns counted from p1 in tiles of NB
    const int t = blockIdx.x;
    int bi = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while (bi * (bi + 1) / 2 > t) --bi;
    const int bj = t - bi * (bi + 1) / 2;
    const int r0 = p1 + bi * NB, c0 = p1 + bj * NB, nr = min(NB, d.D - r0), nc = min(NB, d.D - c0);
    for (int i = tid; i < NB * NB; i += 256) {
        const int r = i / NB, k = i - r * NB;
        P1[r][k] = (r < nr && k < nb) ? d.L[(size_t)(r0 + r) * d.D + p0 + k] : 0.0;
        P2[r][k] = (r < nc && k < nb) ? d.L[(size_t)(c0 + r) * d.D + p0 + k] : 0.0;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;                  // 16 x 16 threads, 3 x 3 outputs each
    double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int k = 0; k < NB; ++k) {
        double a[3], b[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) { a[u] = P1[3 * ty + u][k]; b[u] = P2[3 * tx + u][k]; }
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) acc[u][v] += a[u] * b[v];
    }
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int r = r0 + 3 * ty + u, c = c0 + 3 * tx + v;
            if (3 * ty + u < nr && 3 * tx + v < nc && c <= r) d.L[(size_t)r * d.D + c] -= acc[u][v];
        }
}

// forward and backward substitution, the right-hand side in LDS; then dX, the status, the retry flag
__global__ __launch_bounds__(1024) void k_dn_solve(Dense d, float *dx, int *status, int attempt) {
    if (skip(d, attempt)) return;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, D = d.D;
    double *z = sm;                                          // [D]
    double *part = z + D;                                    // [NB][kSl]
    double *t = part + NB * 21;                              // [NB]
    constexpr int kSl = 21;                                  // slices per panel row / column: 48 x 21 = 1008 threads
    __shared__ double Lp[NB][NB + 1];
    __shared__ int has_nan;
    if (tid == 0) has_nan = 0;
    if (d.ctl[kCtlFail]) {
        // failed factorisation: the pose update is skipped (ba.py:9-13)
        for (int i = tid; i < D; i += 1024) dx[i] = 0.0f;
        if (tid == 0) { status[0] = BT_SOLVE_CHOL_FAILED; d.ctl[kCtlRetry] = 0; }
        return;
    }
    for (int i = tid; i < D; i += 1024) z[i] = d.z[i];
    __syncthreads();
    const int q = tid / kSl, sl = tid - q * kSl;             // q: row (forward) / column (backward) of the panel
    // ---- L w = y
    for (int p0 = 0; p0 < D; p0 += NB) {
        const int nb = min(NB, D - p0);
        for (int i = tid; i < nb * nb; i += 1024) { const int r = i / nb, c = i - r * nb; Lp[r][c] = c <= r ? d.L[(size_t)(p0 + r) * D + p0 + c] : 0.0; }
        if (q < nb) {
            const double *row = d.L + (size_t)(p0 + q) * D;
            double acc = 0.0;
            for (int c = sl; c < p0; c += kSl) acc += row[c] * z[c];
            part[q * kSl + sl] = acc;
        }
        __syncthreads();
        if (tid < nb) { double acc = z[p0 + tid]; for (int s = 0; s < kSl; ++s) acc -= part[tid * kSl + s]; t[tid] = acc; }
        __syncthreads();
        if (tid < 64) {                                      // one wave: the panel's triangle, a lane per row
            for (int k = 0; k < nb; ++k) {
                const double zk = t[k] / Lp[k][k];
                if (tid == k) t[k] = zk;
                if (tid > k && tid < nb) t[tid] -= Lp[tid][k] * zk;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        __syncthreads();
        if (tid < nb) z[p0 + tid] = t[tid];
        __syncthreads();
    }
    // ---- L^T x = w
    for (int p0 = ((D - 1) / NB) * NB; p0 >= 0; p0 -= NB) {
        const int nb = min(NB, D - p0), p1 = p0 + nb;
        for (int i = tid; i < nb * nb; i += 1024) { const int r = i / nb, c = i - r * nb; Lp[r][c] = c <= r ? d.L[(size_t)(p0 + r) * D + p0 + c] : 0.0; }
        if (q < nb) {
            double acc = 0.0;
            for (int r = p1 + sl; r < D; r += kSl) acc += d.L[(size_t)r * D + p0 + q] * z[r];
            part[q * kSl + sl] = acc;
        }
        __syncthreads();
        if (tid < nb) { double acc = z[p0 + tid]; for (int s = 0; s < kSl; ++s) acc -= part[tid * kSl + s]; t[tid] = acc; }
        __syncthreads();
        if (tid < 64) {
            for (int k = nb - 1; k >= 0; --k) {
                const double xk = t[k] / Lp[k][k];
                if (tid == k) t[k] = xk;
                if (tid < k) t[tid] -= Lp[k][tid] * xk;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        __syncthreads();
        if (tid < nb) z[p0 + tid] = t[tid];
        __syncthreads();
    }
    for (int i = tid; i < D; i += 1024) if (z[i] != z[i]) has_nan = 1;
    __syncthreads();
    if (attempt == 0 && has_nan) {                           // once more with lm = 1e-3 (ba.py:324-325)
        if (tid == 0) d.ctl[kCtlRetry] = 1;
        return;
    }
    for (int i = tid; i < D; i += 1024) dx[i] = (float)z[i];
    if (tid == 0) { status[0] = attempt == 0 ? BT_SOLVE_OK : BT_SOLVE_RETRIED; d.ctl[kCtlRetry] = 0; }
}

}  // namespace dn

size_t dense_solve_lds_bytes(const PlanDev &pd) { return ((size_t)pd.D + dn::NB * 21 + dn::NB) * sizeof(double); }

// [S | y] -> dX for a wide plan.  ev0 / ev1: start of the first and stop of the last kernel (measurement), or null.
int launch_solve_dense(const PlanDev &pd, const StepArgs &a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    using namespace dn;
    Dense d;
    d.L = reinterpret_cast<double *>(a.lfac); d.z = d.L + (size_t)pd.D * pd.D; d.ctl = a.status + kCtlOffset; d.D = pd.D;
    const int D = pd.D;
    const size_t lds = dense_solve_lds_bytes(pd);
    static LdsLimit lds_limit;
    if (!lds_limit.ensure(reinterpret_cast<const void *>(&k_dn_solve), lds + 24 * 1024, pd.dev_id)) return BT_EHIP;     // (+ the kernel's static arrays)
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int nload = (int)std::min<size_t>(4096, ((size_t)D * D + 255) / 256);
        if (attempt == 0 && ev0) hipExtLaunchKernelGGL(k_dn_load, dim3(nload), dim3(256), 0, st, ev0, nullptr, 0, d, a.S, a.y, a.ep, attempt);
        else hipLaunchKernelGGL(k_dn_load, dim3(nload), dim3(256), 0, st, d, a.S, a.y, a.ep, attempt);
        for (int p0 = 0; p0 < D; p0 += NB) {
            const int p1 = std::min(D, p0 + NB), below = D - p1;
            hipLaunchKernelGGL(k_dn_panel, dim3(1), dim3(256), 0, st, d, p0, attempt);
            if (below > 0) {
                hipLaunchKernelGGL(k_dn_trsm, dim3((below + 255) / 256), dim3(256), 0, st, d, p0, attempt);
                const int nt = (below + NB - 1) / NB;
                hipLaunchKernelGGL(k_dn_syrk, dim3(nt * (nt + 1) / 2), dim3(256), 0, st, d, p0, attempt);
            }
        }
        if (attempt == 1 && ev1) hipExtLaunchKernelGGL(k_dn_solve, dim3(1), dim3(1024), lds, st, nullptr, ev1, 0, d, a.dx, a.status, attempt);
        else hipLaunchKernelGGL(k_dn_solve, dim3(1), dim3(1024), lds, st, d, a.dx, a.status, attempt);
    }
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

}  // namespace bt
