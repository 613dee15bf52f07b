// plan_pack.hip — the edge list as the planner reads it, packed on the device before it is copied back.
// The caller's indices are three int64 tensors (batrack.py:100-102): 24 bytes per edge.  The planner needs 8: one word
// kk << 32 | ii << 16 | jj (ba_plan.cpp).  This kernel packs and range-checks them, so that a third of the bytes
// crosses PCIe and the host never walks the int64 arrays.
#include <hip/hip_runtime.h>

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

}  // namespace bt
