/* batrack_ga.h — C ABI of the dense global-alignment LOSSES and their gradients, SURVEY.md §8 row f-4.
 *
 * Reference: RefineNet.forward and the losses it calls,
 *   /root/reference/main/global_refine/model/refine_net.py:123-127 (get_trajs_scales), :148-174 (get_frame_scaled_depth),
 *   :252-268 (spatial huber term), :199-225 (inter_frame_loss, O(Q S N^2)), :300-345 (pts_3d_loss),
 * driven by the Adam loop of model/trainer.py:23-77.  The reference is a Python/autograd module (pypose for the poses);
 * it has no FFI.  This header is what a binding of those forward computations binds; batrack_amd/global_refine.py does
 * so.  bt_ga_forward gives the values (all five terms of refine_net.py:252-392), bt_ga_backward_total the gradients of their
 * weighted total with respect to every parameter of the Adam loop.  Device pointers, sizes, integer status
 * codes (include/batrack_ba.h); nothing allocates or synchronises.
 */
#ifndef BATRACK_GA_H
#define BATRACK_GA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int64_t T, N, S;               /* frames, tracks per frame, local window (slots per track)      */
    int64_t gh, gw, H, W, Q;       /* scale grid, image size (refine_net.py:158-159), query frames */
    const float *trajs_2d;         /* [T,N,S,2]                                                    */
    const void *trajs_disp;        /* [T,N,S] float32, or float16 when half_disp                   */
    const void *trajs_disp_mono;   /* [T,N,S] float32, or float16 when half_disp                   */
    const float *trajs_vis;        /* [T,N,S]                                                      */
    const float *trajs_static;     /* [T,N,S]                                                      */
    const int64_t *jj;             /* [T,S] target frame of slot s, UNclamped (refine_net.py:92-97) */
    const float *intrinsics;       /* [T,4] fx fy cx cy                                            */
    const float *pose;             /* [T,7] tx ty tz qx qy qz qw (pypose SE3 layout)               */
    const int64_t *query;          /* [Q] grid_query_frames                                        */
    const float *trajs_scales;     /* [T,N,S] parameter (refine_net.py:42)                          */
    const float *frame_scales;     /* [T,gh,gw] parameter, raw: exp(x / 10) is applied (scale_mode 'exp') */
    const float *frame_shifts;     /* [T]                                                          */
    float pw_break;                /* refine_net.py:39                                             */
    int32_t half_disp;             /* 1: disparities are float16 and the depth residual of the spatial term is formed in
                                      float16 (BASELINE.json configs[4]); everything else stays float32 */
} bt_ga_args;

/* mono_scaled_out [T,N,S] float32 = get_frame_scaled_depth(); losses[5] (device, float64) = spatial huber term,
 * inter_frame_loss, pts_3d_loss, cam_smooth_vec_loss (refine_net.py:356-360), scale_grid_smoothness_loss (:362-392) — each
 * the reference's mean.  `which` selects: bit 0 spatial (always computes mono_scaled_out), bit 1 inter-frame, bit 2 3-D
 * points, bit 3 the two smoothness terms, with the mode of the scale-grid term in bits 8-9 (BT_GA_SMOOTH_*; forward() itself
 * always passes 'l1', refine_net.py:289,299).  Entries not selected read 0.  Enqueued on `stream`. */
#define BT_GA_SMOOTH_L1 0
#define BT_GA_SMOOTH_L2 1
#define BT_GA_SMOOTH_HUBER 2
int bt_ga_forward(const bt_ga_args *args, float *mono_scaled_out, double *losses, int32_t which, void *stream);

/* Weights of RefineNet.forward's total: with loss_weight_dict (refine_net.py:274-297; run_global_refine.py:61-67 passes
 * {spatial 5.0, inter_frame 0.3, pts_3d 1.0, cam_smooth_vec 1.0, scale_smoothness 0.3}) the entries of the dict; with
 * loss_weight_dict = None (:299-301) {1, alpha, 0, 0, scale_smoothness_weight (default 0.1)}. */
typedef struct {
    float spatial, rigid, pts3d, cam_smooth, scale_smooth;
    int32_t smooth_mode;            /* BT_GA_SMOOTH_* of the scale-grid term */
} bt_ga_weights;

/* Gradient of the weighted total with respect to every parameter the reference's Adam loop steps (trainer.py:33-43):
 *   grad_trajs_scales [T,N,S]   through exp((p - mean_n p) / pw_break)                  (refine_net.py:123-127)
 *   grad_frame_scales [T,gh,gw] through exp(g / 10), the bilinear sample (:139-174) and the smoothness term
 *   grad_pose [T,7]             pts_3d_loss in the convention of pypose's backward — the gradient of the LEFT perturbation
 *                               Exp(delta) X in the first six numbers (tau, phi), 0 in the seventh — plus the plain
 *                               derivative of cam_smooth_vec_loss with respect to the seven stored numbers (it reads them
 *                               through .tensor()); NULL: not wanted
 *   grad_intrinsics [T,4]       per frame (fx fy cx cy), inter_frame_loss and pts_3d_loss through iproj; the reference's K
 *                               (refine_intrinsics: intrinsics = K * K_scale for every frame, :131-136) gets
 *                               K_scale * sum_t of it; NULL: not wanted
 * `mono_scaled` is the [T,N,S] output of bt_ga_forward for the same arguments; `g_mono_scaled` is [T,N,S] float scratch (on
 * return: d total / d mono_scaled).  Query frames must be distinct (bt_ga_forward too).  All outputs are overwritten.
 * Enqueued on `stream`. */
int bt_ga_backward_total(const bt_ga_args *args, const float *mono_scaled, const bt_ga_weights *weights, float *g_mono_scaled,
                         float *grad_trajs_scales, float *grad_frame_scales, float *grad_pose, float *grad_intrinsics, void *stream);

/* The same for  w_spatial * spatial + w_rigid * inter_frame  and the two parameters that total reaches. */
int bt_ga_backward(const bt_ga_args *args, const float *mono_scaled, float w_spatial, float w_rigid, float *g_mono_scaled,
                   float *grad_trajs_scales, float *grad_frame_scales, void *stream);

/* ---- the hand-off from the sparse-SLAM stage: RefineNet._init_from_ba, refine_net.py:53-121
 *
 * bt_ga_mat_to_se3: `pp.mat2SE3(cams_T_world)` (refine_net.py:61): T row-major 4x4 matrices [T,16] -> poses [T,7]
 * (tx ty tz qx qy qz qw).  The quaternion by the trace / largest-diagonal-entry rule, normalised; its SIGN is whatever the
 * branch gives (q and -q are the same pose; pypose is not in the image, see tests/golden/refstubs/pypose).
 *
 * bt_ga_sample_disp_mono: the loop refine_net.py:99-110 — for every frame t, slot s and track n the depth map of frame
 * clamp(t + s - S / 2, 0, T - 1) is sampled bilinearly at trajs_2d[t, n, s] exactly as model/utils.py:bilinear_sample2d does
 * (corner indices floor(x), floor(x) + 1 clamped to the image, weights from the UNclamped corners), and the result is
 * 1 / max(depth, 1e-2).  dmaps [T,H,W] float32 (channel 0 of results['dmaps']), out [T,N,S] float32. */
int bt_ga_mat_to_se3(const float *mats, float *poses, int64_t T, void *stream);
int bt_ga_sample_disp_mono(const float *dmaps, const float *trajs_2d, float *disp_mono_out,
                           int64_t T, int64_t N, int64_t S, int64_t H, int64_t W, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BATRACK_GA_H */
