/* batrack_se3.h — C ABI of the element-wise SE(3) operations.
 *
 * Replaces, for the SE3 group (group_id 3) and its forward operations, the reference's
 * compiled module `lietorch_backends`
 *     expm logm inv mul adj adjT act act4 as_matrix
 *                /root/reference/main/backend/lietorch/src/lietorch.cpp:286-316
 *     kernels    /root/reference/main/backend/lietorch/src/lietorch_gpu.cu:20-294
 * which the caller reaches through lietorch/group_ops.py:28-66 (outside BA: init_motion
 * batrack.py:176-187, reproject :337, terminate :904-906, update_point_cloud :823-850).
 *
 * Conventions of the reference's headers (include/se3.h, so3.h, common.h): an element is 7
 * contiguous scalars (tx ty tz qx qy qz qw); the quaternion is re-normalised on every load and
 * after every product; tangent order (tau, phi); small-angle switch EPS = 1e-6.
 * All pointers are DEVICE pointers to contiguous [B, dim] arrays; `stream` is a hipStream_t as
 * void*.  dtype: 0 = float32, 1 = float64.  Return: BT_OK (0) / BT_EINVAL (-1) / BT_EHIP (-3).
 */
#ifndef BATRACK_SE3_H
#define BATRACK_SE3_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int bt_se3_exp(const void *xi /*[B,6]*/, void *X /*[B,7]*/, int64_t B, int dtype, void *stream);
int bt_se3_log(const void *X /*[B,7]*/, void *xi /*[B,6]*/, int64_t B, int dtype, void *stream);
int bt_se3_inv(const void *X, void *Y, int64_t B, int dtype, void *stream);
int bt_se3_mul(const void *X, const void *Y, void *Z, int64_t B, int dtype, void *stream);
int bt_se3_act(const void *X, const void *p /*[B,3]*/, void *q /*[B,3]*/, int64_t B, int dtype, void *stream);
int bt_se3_act4(const void *X, const void *p /*[B,4]*/, void *q /*[B,4]*/, int64_t B, int dtype, void *stream);
int bt_se3_adj(const void *X, const void *a /*[B,6]*/, void *b /*[B,6]*/, int64_t B, int dtype, void *stream);
int bt_se3_adjT(const void *X, const void *a /*[B,6]*/, void *b /*[B,6]*/, int64_t B, int dtype, void *stream);
int bt_se3_matrix(const void *X, void *M /*[B,16] row-major 4x4*/, int64_t B, int dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BATRACK_SE3_H */
