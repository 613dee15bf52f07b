/* batrack_patchify.h — C ABI of the patch gather.
 *
 * Replaces `cuda_corr.patchify_forward` + the Python bilinear blend of
 * `altcorr.patchify(net, coords, radius, mode)`
 *     /root/reference/main/backend/altcorr/correlation_kernel.cu:16-47,288-307
 *     /root/reference/main/backend/altcorr/correlation.py:33-68
 * used by the caller every frame for patch coordinates, colours and depths
 * (/root/reference/main/batrack.py:321,323,438; radius 0 or P//2 = 0).
 *
 * net [B,C,H,W] float32, coords [B,M,2] float32 (x, y), all DEVICE pointers, contiguous.
 *   bilinear = 0: out [B,M,C,D,D], D = 2*radius + 2: out[b,m,c,a,e] = net[b,c,floor(y)+a-r, floor(x)+e-r],
 *                 0 outside the image (the raw gather the reference's kernel produces);
 *   bilinear = 1: out [B,M,C,d,d], d = 2*radius + 1: the reference's four-tap blend of that gather
 *                 with weights from the fractional parts of (x, y) — fused here, the D x D tensor is
 *                 never written.
 * `stream` is a hipStream_t as void*.  Returns BT_OK / BT_EINVAL / BT_EHIP.  The reference's
 * patchify_backward and the `corr` kernels are not on the inference path (dead code in the caller). */
#ifndef BATRACK_PATCHIFY_H
#define BATRACK_PATCHIFY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int bt_patchify(const float *net, int64_t B, int64_t C, int64_t H, int64_t W,
                const float *coords, int64_t M, int32_t radius, int32_t bilinear,
                float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BATRACK_PATCHIFY_H */
