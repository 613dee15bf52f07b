/* batrack_ba.h — C ABI of the MI355X-native bundle-adjustment backend.
 *
 * Drop-in boundary for the reference's BA hot path.  The reference exposes a
 * Python function, not an FFI:
 *
 *   BA_rgbd_droid(poses, patches, patches_monodisp, intrinsics, targets_2d,
 *                 targets_disp, weights, lmbda, ii, jj, kk, bounds, ep, PRINT,
 *                 fixedp, structure_only, loss, alpha) -> (poses, patches)
 *                                   /root/reference/main/backend/ba.py:217
 *   called from BATRACK.update()    /root/reference/main/batrack.py:869-875
 *
 * This header is what a binding for that function binds (batrack_amd/backend/ba.py
 * does so through ctypes; INTEGRATION.md shows the stub).  Plain pointers and
 * sizes only; every pointer in bt_ba_args is a DEVICE pointer; `stream` is a
 * hipStream_t passed as void*.  Nothing here allocates in the step functions,
 * synchronises the device, or throws: integer status codes only.
 *
 * Two objects:
 *   bt_plan   the structure of one edge list (ii, jj, kk, fixedp): unique tracks
 *             (replaces torch.unique at ba.py:276), camera pairs, wave tiles, the
 *             block-sparsity of the reduced camera system.  Built once per edge
 *             list (the reference's caller keeps one list for 2*ITER calls,
 *             batrack.py:869-875), reused by every step on it.
 *   workspace caller-owned device scratch of bt_plan_workspace_bytes() bytes.
 *
 * Arithmetic: inputs and outputs are float32 as the reference's (batrack.py:74-91).  The per-edge maths (reprojection,
 * Jacobians, robust weights, ba.py:228-266 / projective_ops.py:54-100) is float64 on those inputs for every plan of fewer than
 * 2048 tiles (bt_plan_edge_precision() == 8: every window of the real pipeline); beyond, the wave-per-tile kernels keep the
 * reprojection and the residual in float64 and do the Jacobians and their products in float32 (== 6; switched off by
 * bt_config_wave_per_tile_kernels(0)); float32 throughout (== 4) only where a tile's E does not fit LDS as double: plans with
 * MORE THAN 64 tracks seen by 33 .. 64 free cameras each (a few such landmarks are stepped on a float64 path of their own);
 * sums across edges, the reduced system and the retraction are float64 always.  The factorisation is float64 wherever the
 * block-sparse factor fits LDS as double (every window of the real pipeline, the 64-keyframe benchmark graph) and wherever the
 * system is solved dense (more than 255 free poses; fewer poses whose factor fills in beyond LDS and is cheaper dense); long thin
 * bands beyond LDS are factored in float32 and refined twice against the float64 system, which leaves dX at the float32 rounding
 * it is stored in.  The update agrees with the reference's float64 run to ~1e-6 (north_star: 1e-5).
 */
#ifndef BATRACK_BA_H
#define BATRACK_BA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bt_plan bt_plan;

enum { BT_OK = 0, BT_EINVAL = -1, BT_ENOMEM = -2, BT_EHIP = -3, BT_EUNSUPPORTED = -4 };
enum { BT_NO_MATCH = 1 };   /* bt_plan_create_shifted: not an error, the edge list is no shifted copy of the source plan's */
enum { BT_LOSS_TRIVIAL = 0, BT_LOSS_HUBER = 1, BT_LOSS_CAUCHY = 2 };   /* ba.py:81-100 */

/* status word written by the solver (bt_ba_status): */
enum { BT_SOLVE_OK = 0, BT_SOLVE_CHOL_FAILED = 1,   /* dX = 0, ba.py:9-13            */
       BT_SOLVE_RETRIED = 2 };                      /* NaN -> lm = 1e-3, ba.py:324-325 */

typedef struct {
    int64_t E;            /* edges assembled by this plan (owned tracks)          */
    int64_t n_buf;        /* pose / intrinsics buffer length                   */
    int64_t p_tot;        /* patch slots                                       */
    int64_t fixedp;       /* poses [0, fixedp) are held fixed                  */
    int64_t n_all;        /* max(ii, jj) + 1               (ba.py:219)         */
    int64_t n;            /* free poses = n_all - fixedp   (ba.py:272)         */
    int64_t m;            /* distinct tracks in kk         (ba.py:276-277)     */
    int64_t pairs;        /* distinct (ii, jj) camera pairs                    */
    int64_t tiles;        /* wave tiles (<= 64 tracks each)                    */
    int64_t slots;        /* sum over tiles of edge slots (each 64 lanes wide) */
    int64_t erows;        /* sum over tiles of 6 * (free cameras of the tile)  */
    int64_t max_tile_cams;
    int64_t nnz_blocks;   /* 6x6 blocks of the Cholesky factor incl. fill      */
    int64_t updates;      /* block-update triples of the factorisation         */
    int64_t workspace_bytes;
    int64_t sorted_input; /* 1 if kk was already non-decreasing                */
} bt_plan_info;

/* Build the plan.  ii/jj/kk are int64[E] (the dtype the reference's caller
 * holds, batrack.py:100-102), on the device (on_device=1; copied back once, on an
 * internal non-blocking stream that is synchronised before the call returns — the
 * caller makes sure the arrays are complete, i.e. synchronises the stream that
 * produced them; the call may then come from any host thread while other streams
 * are busy) or on the host (on_device=0).  upload=0 keeps the plan
 * host-only (no HIP call is made: CPU tests).  Errors: BT_EINVAL for indices
 * out of range, BT_EUNSUPPORTED for more than 2048 free poses (beyond 255 — and for fewer poses
 * with a filled-in factor that outgrows LDS — the step uses a dense solver in the global
 * workspace), more than 32768 pose slots, or the edges of one
 * track naming different source frames (the caller's invariant ii = ix[kk], batrack.py:199,
 * is relied upon).  No limit on the cameras a track is seen by: beyond 64 free ones (32 where
 * a plan has few such tracks) it is stepped on a per-track path of its own, slower per edge.
 * n_all_min: lower bound for n_all (0 = derive from the edges).
 * own_lo, own_hi: multi-GPU sharding.  Every rank passes the FULL edge
 * list; the plan assembles only the tracks with own_lo <= kk < own_hi (own_hi = 0: all),
 * while the size and block-sparsity of the reduced system come from all edges, so the
 * all-reduced [S | y] has the same layout and pattern on every rank.  Per-edge inputs
 * (targets, weights) stay indexed by the full edge list. */
int bt_plan_create(const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E,
                   int64_t n_buf, int64_t p_tot, int64_t fixedp, int64_t n_all_min,
                   int64_t own_lo, int64_t own_hi, int on_device, int upload, bt_plan **out);
/* The plan of a SHIFTED edge list, without the host analysis.  In the steady state of the caller's sliding window the edge
 * list of an update() is the list of an earlier one with every frame index moved up by df and every patch index by dk
 * (batrack.py:189-212: factors are appended for the newest frames and dropped for the oldest, patch p of frame f is f * M + p)
 * and fixedp moved along: same tiles, slots, camera pairs and sparsity, only the absolute frame / patch numbers differ.
 * ii/jj/kk: DEVICE int64[E].  If they equal src's edge list plus constants (df for ii and jj, dk for kk, df >= 0, dk >= 0,
 * not both 0), fixedp == src.fixedp + df and n_buf, p_tot are src's, *out becomes a copy of src's device tables with those
 * numbers shifted (one comparison kernel, one device-to-device copy, two small kernels: ~0.1 ms instead of ~1 ms) and BT_OK
 * is returned; otherwise BT_NO_MATCH (> 0) and *out = NULL: call bt_plan_create.  src must be an uploaded, unsharded plan
 * built from device indices (plans of up to 4M edges keep their packed edge list for this comparison); it is only read.
 * The new plan has no host arrays (bt_plan_array returns empty ones). */
int bt_plan_create_shifted(const bt_plan *src, const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E,
                           int64_t n_buf, int64_t p_tot, int64_t fixedp, bt_plan **out);
/* The same against up to four candidate source plans at once (the caller's most recent plans, most likely first): the list is
 * packed once and compared with every candidate in one queue — one host wait for all of them, where a candidate that does not
 * match costs a round trip of its own through bt_plan_create_shifted.  *which (optional) = index of the source that matched.
 * The clone's tables are copied on the library's plan stream without a host wait; the plan's first launches are ordered
 * behind them whatever stream they are given. */
int bt_plan_create_shifted_any(const bt_plan *const *srcs, int nsrc, const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E,
                               int64_t n_buf, int64_t p_tot, int64_t fixedp, int *which, bt_plan **out);
/* The same, SPECULATIVELY, with no host wait at all.  The caller's sliding window shifts its edge list by a whole number of
 * frames and fixedp moves along (batrack.py:189-212, :858): the shift is df = fixedp - src.fixedp frames and dk = df * (p_tot /
 * n_buf) patches.  The clone is made for THAT shift — copies and shift kernels enqueued on the plan stream behind an event
 * recorded on `in_stream` (the stream that produced the index tensors: no synchronisation of it either) — while a kernel compares
 * the packed list with src's under the same assumption and leaves its verdict in pinned memory.  The plan is usable at once; what it
 * computes is meaningless (but memory-safe: every index it holds is inside the caller's buffers) unless the verdict is good, so
 * the caller launches its first step, then calls bt_plan_spec_confirm — by then the verdict has usually arrived — and on BT_NO_MATCH
 * destroys the plan, builds the list's plan with bt_plan_create and repeats the step (the inputs of a step are never modified).
 * BT_NO_MATCH from the create call itself: the shift is not of that form (no speculation possible).  src as for the above. */
int bt_plan_create_shifted_spec(const bt_plan *src, const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E,
                                int64_t n_buf, int64_t p_tot, int64_t fixedp, void *in_stream, bt_plan **out);
/* The two halves of bt_plan_create_shifted_spec apart (round 6): the window moves by the same number of frames update() after
 * update(), so the clone for the NEXT list — its tables depend on the source and the shift only — can be enqueued while the steps of
 * the current list run (bt_plan_preshift: `df` frames, fixedp = src.fixedp + df; BT_NO_MATCH where no such shift fits the buffers),
 * and the call that brings the list pays for one comparison kernel (bt_plan_spec_bind: the list against the one the clone was made
 * for, verdict as above; BT_NO_MATCH, without a kernel, if the list's size / buffers / fixedp are not the clone's).  A clone that
 * was never bound cannot be stepped (BT_EINVAL) and confirms as BT_NO_MATCH; after a successful bind it is the plan
 * bt_plan_create_shifted_spec would have returned: step, then bt_plan_spec_confirm. */
int bt_plan_preshift(const bt_plan *src, int64_t df, bt_plan **out);
int bt_plan_spec_bind(bt_plan *plan, const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E, int64_t n_buf, int64_t p_tot,
                      int64_t fixedp, void *in_stream);
/* BT_OK: the plan is what bt_plan_create would have built (or was not speculative); BT_NO_MATCH: it is not — destroy it; BT_EINVAL:
 * the new list holds an index outside [0, n_buf) / [0, p_tot).  Waits for the verdict if it has not arrived. */
int bt_plan_spec_confirm(bt_plan *plan);

/* The device buffer and the host arrays of a destroyed plan are kept (up to eight of each) for the next
 * bt_plan_create: the caller replaces its edge list every frame (batrack.py:189-212), and a
 * hipMalloc/hipFree pair plus fresh host pages per frame cost more than the analysis itself.
 * bt_plan_pool_trim() releases everything that is kept. */
void bt_plan_destroy(bt_plan *plan);
void bt_plan_pool_trim(void);
int bt_plan_get_info(const bt_plan *plan, bt_plan_info *info);
size_t bt_plan_workspace_bytes(const bt_plan *plan);

/* Host view of a plan array by name (for tests / tooling); returns the element
 * count, or -1 for an unknown name.  Element type is int32 unless noted in
 * DESIGN.md ("slot_lab" is uint16). */
int64_t bt_plan_array(const bt_plan *plan, const char *name, const void **data);

typedef struct {
    const float *poses;        /* [n_buf,7] tx ty tz qx qy qz qw                      */
    const float *patches;      /* [p_tot,3] x y inverse-depth (patch size P = 1)      */
    const float *mono_disp;    /* [p_tot]   depth prior (patches_monodisp)            */
    const float *intrinsics;   /* [n_buf,4] fx fy cx cy                               */
    const float *targets;      /* (u, v) of edge e at targets[e*target_stride + {0,1}] */
    int64_t target_stride;     /* in floats: 3 for the caller's targets_3d[..., :2] view */
    const float *weights;      /* [E,2]                                               */
    float *poses_out;          /* [n_buf,7]  (may alias poses only if structure_only) */
    float *patches_out;        /* [p_tot,3]                                           */
    float bounds[4];           /* x0 y0 x1 y1, strict                                 */
    float lmbda, ep, alpha;    /* ba.py:217 (lm = 1e-4 inside block_solve is fixed)   */
    int32_t loss;              /* BT_LOSS_*                                           */
    int32_t structure_only;    /* ba.py:316                                           */
    int64_t mono_stride;       /* in floats between consecutive patches of mono_disp: the caller's prior is the
                                  strided view patches_local[:, :, mid, 2:] (batrack.py:866); 0 or 1 = contiguous */
    const float *lmbda_per_track; /* NULL, or device floats, one per distinct track of the FULL edge list in ascending patch order:
                                  the reference also accepts a lmbda TENSOR shaped like C (ba.py:299-300); `lmbda` is then ignored.
                                  A sharded plan (own_lo / own_hi) reads the entries of its own tracks from the same full array */
} bt_ba_args;

/* Clears the accumulators ([S | y] and the per-pair sums) inside `workspace`.  Call once
 * after allocating a workspace (or pass zero-filled memory), and again only if a
 * bt_ba_reduce was not followed by its bt_ba_solve_update: every completed step leaves the
 * accumulators clear for the next one (the kernels that consume them reset them). */
int bt_ba_workspace_init(const bt_plan *plan, void *workspace, void *stream);

/* One BA_rgbd_droid call, all phases, enqueued on `stream`. */
int bt_ba_step(const bt_plan *plan, const bt_ba_args *args, void *workspace, void *stream);

/* bt_ba_step with a (start, stop) HIP event pair around every kernel, recorded on
 * `stream` by the launch itself; synchronises, then ms[k] = duration of kernel k (SIX floats):
 * 0 (unused, 0), 1 the Jacobian kernel (residual + Jacobian + assembly + Schur), 2 k_pair_finalize,
 * 3 k_solve, 4 k_update (k_etile_upd), 5 the walk over the edges that back-substitutes the depths where
 * that is a kernel of its own (k_edge2u / k_stream: plans of many tiles; elsewhere it is part of 4);
 * 0 for a kernel the call did not launch.  Measurement only. */
int bt_ba_step_timed(const bt_plan *plan, const bt_ba_args *args, void *workspace, void *stream, float *ms);

/* The same call split at the multi-GPU exchange point:
 *   bt_ba_reduce        residuals, Jacobians, block assembly, Schur complement of
 *                       THIS rank's tracks -> partial reduced system [S | y]
 *   (all-reduce the bt_ba_system() buffer across ranks: sum, float64)
 *   bt_ba_solve_update  damped Cholesky solve, back-substitution, retraction  */
int bt_ba_reduce(const bt_plan *plan, const bt_ba_args *args, void *workspace, void *stream);
int bt_ba_solve_update(const bt_plan *plan, const bt_ba_args *args, void *workspace, void *stream);

/* Device pointer to the reduced system inside `workspace`: (6n)^2 doubles of S
 * (row-major, lower triangle populated) followed by 6n doubles of y. */
double *bt_ba_system(const bt_plan *plan, void *workspace, int64_t *count);

/* Compact exchange form of the same system: only the blocks of S that can be non-zero (the plan's
 * sparsity pattern incl. fill, 36 doubles each, in the solver's block order) followed by y - about 8x
 * fewer bytes than bt_ba_system() on a banded graph (140 KB instead of 1.15 MB at 64 keyframes), so
 * the all-reduce between bt_ba_reduce and bt_ba_solve_update is latency- instead of size-bound:
 *   bt_ba_reduce -> bt_ba_pack -> all-reduce bt_ba_packed() -> bt_ba_unpack -> bt_ba_solve_update
 * No-ops for structure-only steps.  Every rank has the same pattern (the plan is built from the full
 * edge list), so the packed buffers add element-wise. */
int bt_ba_pack(const bt_plan *plan, const bt_ba_args *args, void *workspace, void *stream);
int bt_ba_unpack(const bt_plan *plan, const bt_ba_args *args, void *workspace, void *stream);
double *bt_ba_packed(const bt_plan *plan, void *workspace, int64_t *count);
/* The step in TWO calls around a collective (RCCL over xGMI: all-reduce bt_ba_packed(), sum, float64):
 *   bt_ba_reduce_pack           = bt_ba_reduce + bt_ba_pack
 *   bt_ba_unpack_solve_update   = bt_ba_unpack + bt_ba_solve_update  */
int bt_ba_reduce_pack(const bt_plan *plan, const bt_ba_args *args, void *workspace, void *stream);
int bt_ba_unpack_solve_update(const bt_plan *plan, const bt_ba_args *args, void *workspace, void *stream);

/* ... or with NO collective library: a one-shot exchange over peer-mapped buffers (the message is
 * latency-bound and xGMI is a full mesh).  Every rank allocates an exchange buffer (bt_xchg_alloc: uncached device memory
 * plus its 64-byte hipIpc handle), the ranks swap handles once (any transport) and map each other's buffers (bt_xchg_open).
 * Per step, on the compute stream:
 *   bt_ba_reduce_push          reduce, then write the packed partial system into slot `rank` of EVERY rank's buffer
 *                              (bufs[world], bufs[rank] = the rank's own) and raise this rank's flag there
 *   bt_ba_pull_solve_update    wait for all `world` flags of this epoch in the own buffer, sum the slots in rank order
 *                              (bitwise the same system on every rank), solve, back-substitute, retract
 * `epoch` is 1, 2, 3, ... and must advance by one per step on every rank (two parities of slots: a rank may run one
 * step ahead of a peer, not two — guaranteed by the flags themselves).  A peer that never arrives makes the pull give up
 * after a bounded wait and sets BT_XCHG_TIMEOUT in the second status word (bt_ba_xchg_status); it never hangs the GPU.
 * world <= 16.  Structure-only steps exchange nothing: both calls then behave like bt_ba_reduce / bt_ba_solve_update. */
#define BT_XCHG_TIMEOUT 1
size_t bt_xchg_bytes(const bt_plan *plan, int world);
int bt_xchg_alloc(size_t bytes, void **buf, unsigned char handle[64]);
int bt_xchg_open(const unsigned char handle[64], void **peer);
int bt_xchg_close(void *peer);
int bt_xchg_free(void *buf);
int bt_ba_reduce_push(const bt_plan *plan, const bt_ba_args *args, void *workspace, void *const *bufs, int world, int rank,
                      int64_t epoch, void *stream);
int bt_ba_pull_solve_update(const bt_plan *plan, const bt_ba_args *args, void *workspace, void *own_buf, int world,
                            int64_t epoch, void *stream);
/* Synchronous read-back of the exchange status word (0, or BT_XCHG_TIMEOUT once a pull gave up). */
int bt_ba_xchg_status(const bt_plan *plan, void *workspace, void *stream, int32_t *status);

/* Device pointer to dX [n,6] floats inside `workspace`. */
float *bt_ba_dx(const bt_plan *plan, void *workspace);
/* Synchronous read-back of the solver status word (BT_SOLVE_*). */
int bt_ba_status(const bt_plan *plan, void *workspace, void *stream, int32_t *status);

/* Which Jacobian kernel the steps of this (uploaded) plan launch: 0 = k_tile (one tile per workgroup), 1 = k_stream (two
 * waves per tile, tiles streamed), 2 = k_edge2 (edge-major, one wave per tile, two edges per lane), 3 = k_etile (pair-major lanes, one tile of 16
 * tracks per workgroup: sliding-window graphs) — chosen from the plan's size and shape; -1 for a host-only
 * plan.  For tests and tooling. */
int bt_plan_jacobian_kernel(const bt_plan *plan);
/* Precision of the per-edge maths (reprojection, Jacobians, robust weights, the products they enter, E) of this plan's
 * steps: 8 = float64 on the float32 inputs — every plan that takes k_tile and whose tiles' E fits LDS as double (up to
 * 32 free cameras per tile: every window of the real pipeline, which has 15) — 4 = float32 like the reference's
 * own run (plans with more than 64 tracks seen by 33 .. 64 free cameras each; BT_FORCE prec=f32), 6 = mixed: float64 reprojection and residual,
 * float32 Jacobians and products (k_stream / k_edge2, graphs of >= 2048 tiles; update within 1e-5 like 8, S and y 1e-7).  Sums across edges, the reduced system and its factorisation are float64 either way. */
int bt_plan_edge_precision(const bt_plan *plan);
/* The wave-per-tile kernels for graphs of >= 2048 tiles (k_stream, k_edge2: about three times the float64 tile kernels'
 * throughput there).  They evaluate an edge in MIXED precision — reprojection and residual in float64 on the float32 inputs,
 * Jacobians and their products in float32 — which keeps the pose / depth update within the 1e-5 of the reference's float64
 * run this library promises (measured 1e-6 .. 5e-6 on the benchmark graphs; S and y themselves 1e-7 instead of the tile
 * kernels' 1e-13).  On by default; enable = 0 lays out the plans created afterwards for the float64 tile kernels whatever
 * their size (a plan keeps the layout it was built with), enable < 0 only queries.  Returns the previous setting.
 * BT_WPT_KERNELS=0 in the environment is the same switch for a whole process. */
int bt_config_wave_per_tile_kernels(int enable);
/* 1 if the passes over the edges of this plan ran on the device (bt_plan_create with device index tensors and 4096 edges or
 * more, any layout, whole or sharded: per-track figures, a radix sort and the edge-sized tables by kernels, the host laid out
 * tracks, pairs, tiles and the reduced system from the per-track figures), 0 if the host analysed the edges (host arrays,
 * BT_PLAN_DEVICE=0, a target frame more than 63 away from its track's source frame, a rank without tracks), also for a shifted
 * copy (its tables are its source's).  For tests and tooling. */
int bt_plan_built_on_device(const bt_plan *plan);

/* Library/ABI version and the gfx target the kernels were compiled for. */
int bt_version(void);
const char *bt_target_arch(void);

#ifdef __cplusplus
}
#endif
#endif /* BATRACK_BA_H */
