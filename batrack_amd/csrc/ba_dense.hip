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

// the diagonal block of panel p, factored by ONE wave: lane r holds row r in registers, a pivot's scaled column reaches the other
// lanes through LDS as broadcast reads (the loops are unrolled: every register index and LDS offset is a constant), no workgroup
// barrier.  (The first version — 256 threads on the block in LDS, three barriers and an IEEE sqrt + divide per pivot — took 49 us a
// panel, a third of the solver's time; the column by v_readlane instead of LDS 26 us: two readlanes and their hazard slots per product.)
__device__ __forceinline__ double readlane_d(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffll), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// 1 / sqrt(x) in double: the float32 seed and three Newton steps (the third is there for the seed's worst case; x > 0, and scaled so
// that the float32 seed neither overflows nor flushes)
__device__ __forceinline__ double rsqrt_d(double x) {
    int e;
    const double m = frexp(x, &e);                             // x = m 2^e, m in [0.5, 1)
    const int h = e >> 1;                                       // x = (m 2^(e - 2h)) 4^h, the bracket in [0.5, 2)
    const double xs = ldexp(m, e - 2 * h);
    double y = (double)__builtin_amdgcn_rsqf((float)xs);
    y = y * (1.5 - 0.5 * xs * y * y);
    y = y * (1.5 - 0.5 * xs * y * y);
    y = y * (1.5 - 0.5 * xs * y * y);
    return ldexp(y, -h);
}
#define BT_DN_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
__global__ __launch_bounds__(64) void k_dn_panel(Dense d, int p0, int attempt) {
    if (skip(d, attempt) || d.ctl[kCtlFail]) return;
    // a pivot's scaled column, for every lane to read (the same address in all lanes: a broadcast); two copies taking turns, so that
    // a pivot's store never meets the reads of the pivot before it
    __shared__ __attribute__((aligned(16))) double col[2][NB + 16];
    const int lane = threadIdx.x, nb = min(NB, d.D - p0);
    const bool act = lane < nb;
    // (rows and columns beyond the block: the identity, so that their pivots are 1 and nothing of them reaches the block)
    double row[NB];
    const double *src = d.L + (size_t)(p0 + min(lane, nb - 1)) * d.D + p0;
#pragma unroll
    for (int c = 0; c < NB; ++c) row[c] = (act && c <= lane) ? src[c] : (c == lane ? 1.0 : 0.0);
    bool bad = false;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const double akk = readlane_d(row[k], k);
        if (!(akk > 0.0)) { bad = true; break; }                  // (the same value in every lane: a uniform exit)
        const double il = rsqrt_d(akk);
        const double lrk = lane > k ? row[k] * il : lane == k ? akk * il : 0.0;
        row[k] = lrk;
        double *ck = col[k & 1];
        ck[lane] = lrk;                                           // (lanes 48 .. 63 store zeros behind the column)
        BT_DN_WAVE_SYNC();
#pragma unroll
        for (int c = k + 1; c < NB; ++c) row[c] = fma(-lrk, ck[c], row[c]);      // (used for c <= lane only)
    }
    if (bad) { if (lane == 0) d.ctl[kCtlFail] = 1; return; }
    if (act) {
        double *dst = d.L + (size_t)(p0 + lane) * d.D + p0;
#pragma unroll
        for (int c = 0; c < NB; ++c) if (c <= lane) dst[c] = row[c];
    }
}

// the rows below the panel: L[r, panel] = A[r, panel] L_pp^-T, one thread per row (the diagonal's reciprocals once per workgroup).
// 35 us a panel and now the solver's largest kernel: a wave's time is the LDS latency under each of its 1128 dependent
// multiply-adds.  (Column by column with the finished x[c] taken out of the columns behind it at once — independent multiply-adds —
// was written and not kept: the compiler hoists the 1128 LDS reads of the unrolled triangle to the top, 256 registers and 7.8 KB of
// scratch a lane, memory clobbers or scheduling barriers between the columns notwithstanding.  The form that would pay is the
// inverse of the diagonal block from the panel's wave and this kernel as a plain tile product.)
__global__ __launch_bounds__(256) void k_dn_trsm(Dense d, int p0, int attempt) {
    if (skip(d, attempt) || d.ctl[kCtlFail]) return;
    __shared__ double Lp[NB][NB + 1];
    __shared__ double inv[NB];
    const int tid = threadIdx.x, nb = min(NB, d.D - p0), p1 = p0 + nb;
    for (int i = tid; i < nb * nb; i += 256) { const int r = i / nb, c = i - r * nb; Lp[r][c] = c <= r ? d.L[(size_t)(p0 + r) * d.D + p0 + c] : 0.0; }
    __syncthreads();
    if (tid < NB) inv[tid] = tid < nb ? 1.0 / Lp[tid][tid] : 0.0;
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
            x[c] = t * inv[c];
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
    // (not unrolled all the way: 48 x 6 operands in flight took 256 registers and 388 bytes of scratch a lane — the largest launch 121 us)
#pragma unroll 4
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
        if (tid < 64) {                                      // one wave: the panel's triangle, a lane per row, its unknown in a register
            double tv = 0.0, invd = 0.0;
            if (tid < nb) { tv = z[p0 + tid]; for (int s = 0; s < kSl; ++s) tv -= part[tid * kSl + s]; invd = 1.0 / Lp[tid][tid]; }
#pragma unroll 4
            for (int k = 0; k < nb; ++k) {
                const double lk = Lp[min(tid, NB - 1)][k];
                const double zk = readlane_d(tv, k) * readlane_d(invd, k);
                if (tid == k) tv = zk;
                else if (tid > k && tid < nb) tv = fma(-lk, zk, tv);
            }
            if (tid < nb) z[p0 + tid] = tv;
        }
        __syncthreads();
    }
    // ---- L^T x = w
    for (int p0 = ((D - 1) / NB) * NB; p0 >= 0; p0 -= NB) {
        const int nb = min(NB, D - p0), p1 = p0 + nb;
        for (int i = tid; i < nb * nb; i += 1024) { const int r = i / nb, c = i - r * nb; Lp[r][c] = c <= r ? d.L[(size_t)(p0 + r) * D + p0 + c] : 0.0; }
        {   // (a column per LANE here: the wave's loads of a row of L are then contiguous — with the forward sweep's mapping every lane
            //  of a load sat in a row of its own, and this kernel took 1.2 ms of a 255-pose solve)
            const int qb = tid % NB, sb = tid / NB;
            if (sb < kSl && qb < nb) {
                double acc = 0.0;
                for (int r = p1 + sb; r < D; r += kSl) acc += d.L[(size_t)r * D + p0 + qb] * z[r];
                part[qb * kSl + sb] = acc;
            }
        }
        __syncthreads();
        if (tid < 64) {
            double tv = 0.0, invd = 0.0;
            if (tid < nb) { tv = z[p0 + tid]; for (int s = 0; s < kSl; ++s) tv -= part[tid * kSl + s]; invd = 1.0 / Lp[tid][tid]; }
            for (int k = nb - 1; k >= 0; --k) {
                const double lk = Lp[k][min(tid, NB - 1)];
                const double xk = readlane_d(tv, k) * readlane_d(invd, k);
                if (tid == k) tv = xk;
                else if (tid < k) tv = fma(-lk, xk, tv);
            }
            if (tid < nb) z[p0 + tid] = tv;
        }
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
            hipLaunchKernelGGL(k_dn_panel, dim3(1), dim3(64), 0, st, d, p0, attempt);
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
