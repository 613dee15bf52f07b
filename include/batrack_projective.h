/* batrack_projective.h — C ABI of the fused reprojection.
 *
 * Replaces, for the non-Jacobian call of the reference's
 *     pops.transform(poses, patches, intrinsics, ii, jj, kk, depth=, valid=, tonly=)
 *                /root/reference/main/backend/projective_ops.py:54-75,102-105
 * the chain iproj (:19-29) -> Gij = G_j * G_i^-1 (lietorch inv, mul) -> act4 -> proj (:32-52), which the caller
 * runs once per frame over the whole edge list for its map filtering (batrack.py:327-338), for `flow_mag`
 * (:1017, projective_ops.py:112-122) and for the point-cloud export.  One thread per (edge, patch pixel); the
 * group arithmetic follows the same sequence of float32 operations as the element-wise kernels of batrack_se3.h.
 *
 * All pointers are DEVICE pointers, contiguous:
 *   poses       [n_poses, 7]   tx ty tz qx qy qz qw (re-normalised on load)
 *   patches     [n_patches, 3, p, p]   planes x, y, inverse depth; patch_elems = p*p
 *   intrinsics  [n_poses, 4]   fx fy cx cy
 *   ii, jj, kk  [E] int64      source frame, target frame, patch of every edge
 *   coords      [E, p, p, 2]   (u, v), or [E, p, p, 3] = (u, v, projected inverse depth) with BT_REPROJECT_DEPTH
 *   valid       [E, p, p]      1.0 where Z > 0.2 else 0.0; may be NULL
 * An edge with an index out of range yields NaN coordinates and valid = 0 (the reference's gather would raise).
 * Z is clamped at 1e-2 before the division (projective_ops.py:43).
 * Return: BT_OK (0) / BT_EINVAL (-1) / BT_EHIP (-3), as in batrack_ba.h.
 */
#ifndef BATRACK_PROJECTIVE_H
#define BATRACK_PROJECTIVE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BT_REPROJECT_DEPTH 1   /* also emit the projected inverse depth (proj(..., depth=True))        */
#define BT_REPROJECT_TONLY 2   /* translation-only relative motion (transform(..., tonly=True), :61-64) */

int bt_reproject(const float *poses, int64_t n_poses, const float *patches, int64_t n_patches, int64_t patch_elems,
                 const float *intrinsics, const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E,
                 int32_t mode, float *coords, float *valid, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BATRACK_PROJECTIVE_H */
