// projective_kernels.hip — fused reprojection of patches along the edge list for gfx950:
// iproj -> G_j * G_i^-1 -> act4 -> proj in one pass, one thread per (edge, patch pixel).
// Formulas: /root/reference/main/backend/projective_ops.py:19-75, lietorch/include/se3.h:36-56, so3.h:31-60.
// HBM-bound gather: 24 B of indices + 12 B of patch per edge read, 8-16 B written; the pose and
// intrinsics tables (<= 1024 rows) stay in cache.
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/batrack_ba.h"
#include "../../include/batrack_projective.h"

namespace bt {

struct Q4 { float x, y, z, w; };

__device__ __forceinline__ Q4 q_unit(Q4 q) {                                   // so3.h:35-37
    const float n = 1.0f / sqrtf(q.x*q.x + q.y*q.y + q.z*q.z + q.w*q.w);
    return {q.x*n, q.y*n, q.z*n, q.w*n};
}
__device__ __forceinline__ Q4 q_mul(Q4 a, Q4 b) {
    return { a.w*b.x + a.x*b.w + a.y*b.z - a.z*b.y,
             a.w*b.y - a.x*b.z + a.y*b.w + a.z*b.x,
             a.w*b.z + a.x*b.y - a.y*b.x + a.z*b.w,
             a.w*b.w - a.x*b.x - a.y*b.y - a.z*b.z };
}
__device__ __forceinline__ void q_rot(Q4 q, const float *p, float *o) {       // so3.h:55-60
    float ux = q.y*p[2] - q.z*p[1], uy = q.z*p[0] - q.x*p[2], uz = q.x*p[1] - q.y*p[0];
    ux += ux; uy += uy; uz += uz;
    o[0] = p[0] + q.w*ux + (q.y*uz - q.z*uy);
    o[1] = p[1] + q.w*uy + (q.z*ux - q.x*uz);
    o[2] = p[2] + q.w*uz + (q.x*uy - q.y*ux);
}

template <bool DEPTH, bool TONLY>
__global__ __launch_bounds__(256) void k_reproject(const float *__restrict__ poses, int64_t n_poses,
                                                   const float *__restrict__ patches, int64_t n_patches, int pe,
                                                   const float *__restrict__ intr, const int64_t *__restrict__ ii,
                                                   const int64_t *__restrict__ jj, const int64_t *__restrict__ kk,
                                                   int64_t total, float *__restrict__ coords, float *__restrict__ valid) {
    constexpr int NO = DEPTH ? 3 : 2;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = pe == 1 ? t : t / pe;
        const int pix = pe == 1 ? 0 : (int)(t - e * pe);
        const int64_t i = ii[e], j = jj[e], k = kk[e];
        float *out = coords + t * NO;
        if (i < 0 || j < 0 || k < 0 || i >= n_poses || j >= n_poses || k >= n_patches) {
            out[0] = out[1] = NAN;
            if (DEPTH) out[2] = NAN;
            if (valid) valid[t] = 0.0f;
            continue;
        }
        const float *pi = poses + 7 * i, *pj = poses + 7 * j, *Ki = intr + 4 * i, *Kj = intr + 4 * j;
        const float *pat = patches + (size_t)k * 3 * pe + pix;
        // X0 = ((x - cx)/fx, (y - cy)/fy, 1, d)                                           projective_ops.py:19-29
        const float d = pat[2 * pe];
        const float X0[3] = { (pat[0] - Ki[2]) / Ki[0], (pat[pe] - Ki[3]) / Ki[1], 1.0f };
        // Gij = G_j * G_i^-1                                                               se3.h:36-47
        const Q4 qi = q_unit({pi[3], pi[4], pi[5], pi[6]}), qj = q_unit({pj[3], pj[4], pj[5], pj[6]});
        const Q4 qiv = {-qi.x, -qi.y, -qi.z, qi.w};
        const float ti[3] = {pi[0], pi[1], pi[2]};
        float tiv[3], tr[3];
        q_rot(qiv, ti, tiv);
        tiv[0] = -tiv[0]; tiv[1] = -tiv[1]; tiv[2] = -tiv[2];
        q_rot(qj, tiv, tr);
        const float tij[3] = { pj[0] + tr[0], pj[1] + tr[1], pj[2] + tr[2] };
        float R[3];
        if (TONLY) { R[0] = X0[0]; R[1] = X0[1]; R[2] = X0[2]; }                           // projective_ops.py:61-64
        else q_rot(q_unit(q_mul(qj, qiv)), X0, R);
        // X1 = (R X0 + t d, d)                                                              se3.h:53-56
        const float X = R[0] + tij[0] * d, Y = R[1] + tij[1] * d, Z = R[2] + tij[2] * d;
        const float iz = 1.0f / fmaxf(Z, 1e-2f);                                            // projective_ops.py:43
        out[0] = Kj[0] * (iz * X) + Kj[2];
        out[1] = Kj[1] * (iz * Y) + Kj[3];
        if (DEPTH) out[2] = iz * d;
        if (valid) valid[t] = Z > 0.2f ? 1.0f : 0.0f;                                       // projective_ops.py:103
    }
}

}  // namespace bt

extern "C" int bt_reproject(const float *poses, int64_t n_poses, const float *patches, int64_t n_patches, int64_t patch_elems,
                            const float *intrinsics, const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E,
                            int32_t mode, float *coords, float *valid, void *stream) {
    if (E < 0 || n_poses < 0 || n_patches < 0 || patch_elems <= 0 || patch_elems > 4096 || (mode & ~3)) return BT_EINVAL;
    if (E == 0) return BT_OK;
    if (!poses || !patches || !intrinsics || !ii || !jj || !kk || !coords) return BT_EINVAL;
    const int64_t total = E * patch_elems;
    int64_t nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)nb), blk(256);
    const int pe = (int)patch_elems;
    switch (mode) {
    case 0: hipLaunchKernelGGL((bt::k_reproject<false, false>), grid, blk, 0, st, poses, n_poses, patches, n_patches, pe, intrinsics, ii, jj, kk, total, coords, valid); break;
    case 1: hipLaunchKernelGGL((bt::k_reproject<true, false>), grid, blk, 0, st, poses, n_poses, patches, n_patches, pe, intrinsics, ii, jj, kk, total, coords, valid); break;
    case 2: hipLaunchKernelGGL((bt::k_reproject<false, true>), grid, blk, 0, st, poses, n_poses, patches, n_patches, pe, intrinsics, ii, jj, kk, total, coords, valid); break;
    default: hipLaunchKernelGGL((bt::k_reproject<true, true>), grid, blk, 0, st, poses, n_poses, patches, n_patches, pe, intrinsics, ii, jj, kk, total, coords, valid); break;
    }
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}
