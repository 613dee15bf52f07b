// plan_pack.hip — the edge list as the planner reads it, packed on the device before it is copied back.
// The caller's indices are three int64 tensors (batrack.py:100-102): 24 bytes per edge.  The planner needs 8: one word
// kk << 32 | ii << 16 | jj (ba_plan.cpp).  This kernel packs and range-checks them, so that a third of the bytes
// crosses PCIe and the host never walks the int64 arrays.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "ba_plan.hpp"

namespace bt {

__global__ __launch_bounds__(256) void k_pack_edges(const long long *ii, const long long *jj, const long long *kk, long long E,
                                                    long long n_buf, long long p_tot, unsigned long long *out, int *bad) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const long long i = ii[e], j = jj[e], k = kk[e];
    if (i < 0 || j < 0 || i >= n_buf || j >= n_buf || k < 0 || k >= p_tot) { *bad = 1; out[e] = 0; return; }
    out[e] = ((unsigned long long)k << 32) | ((unsigned long long)i << 16) | (unsigned long long)j;
}

int launch_pack_edges(const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E, int64_t n_buf, int64_t p_tot,
                      uint64_t *out, int *bad, void *stream) {
    if (E <= 0) return BT_OK;
    hipLaunchKernelGGL(k_pack_edges, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const long long *>(ii), reinterpret_cast<const long long *>(jj), reinterpret_cast<const long long *>(kk),
                       (long long)E, (long long)n_buf, (long long)p_tot, reinterpret_cast<unsigned long long *>(out), bad);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

// ------------------------------------------------------------------ plans of shifted edge lists (bt_plan_create_shifted)
// Is the new packed edge list the old one plus one constant word?  out[0] |= 1 on any mismatch; thread 0 leaves the
// difference of the first edge in out[2..3] (the host splits it into the frame and patch shifts and checks their ranges).
__global__ __launch_bounds__(256) void k_shift_match(const unsigned long long *nw, const unsigned long long *ow, long long E, int *out) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const unsigned long long d0 = nw[0] - ow[0];
    if (nw[e] - ow[e] != d0) atomicOr(out, 1);
    if (e == 0) { out[2] = (int)(unsigned)(d0 & 0xffffffffull); out[3] = (int)(unsigned)(d0 >> 32); }
}

// The tables of a plan that hold absolute frame / patch numbers, shifted in the clone's buffer: kx [m], tile_kx [nkx]
// (-1 = empty lane), tile_ij [nij] (i | j << 16), pair_i / pair_j [P].
__global__ __launch_bounds__(256) void k_plan_shift(int32_t *kx, int m, int32_t *tile_kx, int nkx, int32_t *tile_ij, int nij,
                                                    int32_t *pair_i, int32_t *pair_j, int P, int df, int dk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) kx[i] += dk;
    if (i < nkx && tile_kx[i] >= 0) tile_kx[i] += dk;
    if (i < nij) tile_ij[i] += df | (df << 16);
    if (i < P) { pair_i[i] += df; pair_j[i] += df; }
}

// Bitmap of the patches that carry a track, moved up by dk patch slots (read from the source plan's buffer), and its rank
// table (set bits below each word) rebuilt by one block.
__global__ __launch_bounds__(1024) void k_act_shift(const uint32_t *old_bits, uint32_t *new_bits, int32_t *new_rank, int nwords, int dk) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, per = (nwords + 1023) / 1024, w0 = tid * per, w1 = min(nwords, w0 + per);
    const int q = dk >> 5, r = dk & 31;
    int cnt = 0;
    for (int w = w0; w < w1; ++w) {
        const int s = w - q;
        const uint32_t lo = s >= 0 && s < nwords ? old_bits[s] : 0u, below = s - 1 >= 0 && s - 1 < nwords ? old_bits[s - 1] : 0u;
        const uint32_t v = r ? (lo << r) | (below >> (32 - r)) : lo;
        new_bits[w] = v;
        cnt += __popc(v);
    }
    part[tid] = cnt;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                      // inclusive scan of the 1024 partial counts
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid ? part[tid - 1] : 0;
    for (int w = w0; w < w1; ++w) { new_rank[w] = run; run += __popc(new_bits[w]); }
}

// Pack the new list (as k_pack_edges) and compare it with the old packed list plus `delta` in the same pass; the verdict goes
// straight into pinned host memory (flags[0]: the lists differ, flags[1]: an index out of range) — one launch instead of
// memset + pack + compare + copy back (each HIP call is ~8 us of host time, which is what this path is made of).
__global__ __launch_bounds__(256) void k_pack_match_expect(const long long *ii, const long long *jj, const long long *kk, long long E, long long n_buf, long long p_tot,
                                                           const unsigned long long *ow, unsigned long long delta, unsigned long long *out, int *flags) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const long long i = ii[e], j = jj[e], k = kk[e];
    if (i < 0 || j < 0 || i >= n_buf || j >= n_buf || k < 0 || k >= p_tot) { flags[1] = 1; flags[0] = 1; out[e] = 0; return; }
    const unsigned long long w = ((unsigned long long)k << 32) | ((unsigned long long)i << 16) | (unsigned long long)j;
    out[e] = w;
    if (w - ow[e] != delta) flags[0] = 1;
}

// The same comparison for a clone that was made AHEAD of its list (bt_plan_spec_bind), launched on the caller's own stream in
// front of its first step: no packed copy, and no event either — the last workgroup to finish (a ticket) writes `epoch` behind the
// verdict, and the host polls that word.  In the first call of an update() every HIP call costs 10-25 us of host time (idle
// queues): this is the only one the new list needs before its step.
__global__ __launch_bounds__(256) void k_match_done(const long long *ii, const long long *jj, const long long *kk, long long E, long long n_buf, long long p_tot,
                                                    const unsigned long long *ow, int *flags, unsigned *ticket, int epoch) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) {
        const long long i = ii[e], j = jj[e], k = kk[e];
        const bool bad = i < 0 || j < 0 || i >= n_buf || j >= n_buf || k < 0 || k >= p_tot;
        const unsigned long long w = ((unsigned long long)k << 32) | ((unsigned long long)i << 16) | (unsigned long long)j;
        if (bad || w != ow[e]) {
            if (bad) flags[1] = 1;
            flags[0] = 1;
            __threadfence_system();                 // (in host memory before this workgroup takes its ticket)
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            *ticket = 0u;                           // (for the slot's next use)
            __threadfence_system();
            __hip_atomic_store(flags + 2, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

int launch_match_done(const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E, int64_t n_buf, int64_t p_tot,
                      const uint64_t *ow, int *host_flags, unsigned *ticket, int epoch, void *stream) {
    hipLaunchKernelGGL(k_match_done, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const long long *>(ii), reinterpret_cast<const long long *>(jj), reinterpret_cast<const long long *>(kk),
                       (long long)E, (long long)n_buf, (long long)p_tot, reinterpret_cast<const unsigned long long *>(ow), host_flags, ticket, epoch);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

int launch_pack_match_expect(const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E, int64_t n_buf, int64_t p_tot,
                             const uint64_t *ow, uint64_t delta, uint64_t *out, int *host_flags, void *stream) {
    hipLaunchKernelGGL(k_pack_match_expect, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const long long *>(ii), reinterpret_cast<const long long *>(jj), reinterpret_cast<const long long *>(kk),
                       (long long)E, (long long)n_buf, (long long)p_tot, reinterpret_cast<const unsigned long long *>(ow), (unsigned long long)delta,
                       reinterpret_cast<unsigned long long *>(out), host_flags);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

// the packed list a shifted clone EXPECTS: the source's words moved by `delta` (bt_plan_preshift: the new list is not there yet)
__global__ __launch_bounds__(256) void k_words_add(const unsigned long long *ow, unsigned long long delta, unsigned long long *out, long long E) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) out[e] = ow[e] + delta;
}

int launch_words_add(const uint64_t *ow, uint64_t delta, uint64_t *out, int64_t E, void *stream) {
    hipLaunchKernelGGL(k_words_add, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const unsigned long long *>(ow), (unsigned long long)delta, reinterpret_cast<unsigned long long *>(out), (long long)E);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

int launch_shift_match(const uint64_t *nw, const uint64_t *ow, int64_t E, int *out, void *stream) {
    hipLaunchKernelGGL(k_shift_match, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const unsigned long long *>(nw), reinterpret_cast<const unsigned long long *>(ow), (long long)E, out);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

int launch_plan_shift(int32_t *kx, int m, int32_t *tile_kx, int nkx, int32_t *tile_ij, int nij, int32_t *pair_i, int32_t *pair_j, int P,
                      const uint32_t *old_bits, uint32_t *new_bits, int32_t *new_rank, int nwords, int df, int dk, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int n = std::max(std::max(m, nkx), std::max(nij, P));
    if (n > 0) hipLaunchKernelGGL(k_plan_shift, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, kx, m, tile_kx, nkx, tile_ij, nij, pair_i, pair_j, P, df, dk);
    if (nwords > 0) hipLaunchKernelGGL(k_act_shift, dim3(1), dim3(1024), 0, st, old_bits, new_bits, new_rank, nwords, dk);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

}  // namespace bt
