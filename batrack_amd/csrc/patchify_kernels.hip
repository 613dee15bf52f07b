// patchify_kernels.hip — integer-aligned patch gather with the bilinear blend fused in.
// One thread per output element; the C channel planes of a point are strided by H*W, the
// taps of one thread are at most two adjacent rows x two adjacent columns.
#include <hip/hip_runtime.h>

#include "../../include/batrack_ba.h"
#include "../../include/batrack_patchify.h"

namespace bt {

__device__ __forceinline__ float tap(const float *plane, int i, int j, int H, int W) {
    return (i >= 0 && i < H && j >= 0 && j < W) ? plane[(size_t)i * W + j] : 0.0f;
}

template <bool BILINEAR>
__global__ void k_patchify(const float *net, const float *coords, float *out,
                           int64_t total, int C, int H, int W, int M, int R) {
    // the blend reproduces the reference's float32 products and sums one by one (correlation.py:55-66): no fused multiply-add
#pragma clang fp contract(off)
    const int d = BILINEAR ? 2 * R + 1 : 2 * R + 2;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < total; n += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = n;
        const int e = (int)(t % d); t /= d;          // column offset
        const int a = (int)(t % d); t /= d;          // row offset
        const int c = (int)(t % C); t /= C;
        const int m = (int)(t % M); t /= M;
        const int b = (int)t;
        const float x = coords[((size_t)b * M + m) * 2], y = coords[((size_t)b * M + m) * 2 + 1];
        const float fx = floorf(x), fy = floorf(y);
        const int i = (int)fy + a - R, j = (int)fx + e - R;
        const float *plane = net + ((size_t)b * C + c) * H * W;
        if (!BILINEAR) {
            out[n] = tap(plane, i, j, H, W);
        } else {
            // correlation.py:55-66, same association: ((1-dy)(1-dx)) p00 + ((1-dy) dx) p01 + (dy (1-dx)) p10 + (dy dx) p11
            const float dx = x - fx, dy = y - fy;
            const float p00 = tap(plane, i, j, H, W), p01 = tap(plane, i, j + 1, H, W);
            const float p10 = tap(plane, i + 1, j, H, W), p11 = tap(plane, i + 1, j + 1, H, W);
            const float w00 = (1.0f - dy) * (1.0f - dx), w01 = (1.0f - dy) * dx;
            const float w10 = dy * (1.0f - dx), w11 = dy * dx;
            const float t00 = w00 * p00, t01 = w01 * p01, t10 = w10 * p10, t11 = w11 * p11;     // (contract(off): products rounded, then summed)
            out[n] = ((t00 + t01) + t10) + t11;
        }
    }
}

}  // namespace bt

extern "C" int bt_patchify(const float *net, int64_t B, int64_t C, int64_t H, int64_t W, const float *coords,
                           int64_t M, int32_t radius, int32_t bilinear, float *out, void *stream) {
    if (B < 0 || C <= 0 || H <= 0 || W <= 0 || M < 0 || radius < 0 || radius > 64) return BT_EINVAL;
    const int64_t d = bilinear ? 2 * radius + 1 : 2 * radius + 2;
    const int64_t total = B * M * C * d * d;
    if (total == 0) return BT_OK;
    if (!net || !coords || !out) return BT_EINVAL;
    int64_t nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (bilinear) hipLaunchKernelGGL(bt::k_patchify<true>, dim3((unsigned)nb), dim3(256), 0, st, net, coords, out, total, (int)C, (int)H, (int)W, (int)M, (int)radius);
    else          hipLaunchKernelGGL(bt::k_patchify<false>, dim3((unsigned)nb), dim3(256), 0, st, net, coords, out, total, (int)C, (int)H, (int)W, (int)M, (int)radius);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}
