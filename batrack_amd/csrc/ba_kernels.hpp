// ba_kernels.hpp — argument block and launchers shared by ba_kernels.hip / ba_api.cpp.
#pragma once
#include <hip/hip_runtime_api.h>

#include "ba_plan.hpp"

namespace bt {

// bt_ba_args plus the workspace regions, passed by value to every kernel.
struct StepArgs {
    const float *poses, *patches, *mono, *intr, *targets, *weights;
    int tstride, mstride;         // floats between consecutive edges' targets / patches' depth priors
    float *poses_out, *patches_out;
    float b0, b1, b2, b3, lmbda, ep, alpha;
    const float *lmbda_trk;       // per-track lmbda (ba.py:299-300) or nullptr
    int loss;
    double *S, *y, *pairacc;      // S, y, pairacc are contiguous (cleared together)
    double *priv;                 // private copies of y and the per-pair sums (ba_plan.hpp: kPrivY, kPrivP) or nullptr
    double *packed;               // the non-zero blocks of [S | y] in factor order (multi-GPU exchange buffer)
    float2 *qw;
    float *lfac, *linv, *zvec, *dx, *dx0;
    float *pairgeo;               // [pairs][kPairGeomFloats]: relative pose of every camera pair, left by k_tile for k_pair_finalize
    int *status;
    double *spart;                // [tiles][ntl * 256 + max_rows16]: per-tile Schur products of k_etile when the plan's sp_ok (else unused)
    double *esave;                // [tiles][max_rows16][1 << et_lgts]: the tiles' E (rows of the tile's cameras x its tracks), k_etile -> k_etile_upd
    int prec;                     // 1: the per-edge maths, E, pairgeo and qw are float64 (k_tile path, the default there); 0: float32
    int dbg;                      // env BT_DEBUG_MODE, 0 in production: 16 / 32 launch the cycle-counting variants of the solver / k_tile
};

int configure_kernels(const PlanDev &pd);
// precision of the per-edge maths for this plan: 1 = float64 (graphs that take k_tile, unless BT_FORCE prec=f32 or the
// tile's E would not fit LDS as double), 0 = float32 (k_stream / k_edge2: graphs of >= 2048 tiles)
int edge_precision(const PlanDev &pd);
// wave-per-tile streaming kernels (ba_stream.hip) for graphs of many tiles; mode 0 = pose+structure, 1 = structure-only,
// 2 = depth back-substitution
bool stream_applies(const PlanDev &pd);
int launch_stream(const PlanDev &pd, const StepArgs &a, int mode, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
// the same for slot-uniform graphs in the edge-major layout (ba_stream3.hip)
bool edge_applies(const PlanDev &pd);
int launch_edge(const PlanDev &pd, const StepArgs &a, int mode, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
// its pose+structure reduce with two edges per lane (ba_edge2.hip)
int launch_edge2(const PlanDev &pd, const StepArgs &a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
// the pair-major tile kernel (ba_etile.hip) for the graphs k_tile would take, whenever the plan has the pair-major tables
// (every tile <= 64 camera pairs) and the tile's E fits LDS: 8 / 4 = as double / only as float, 0 = k_tile takes the plan.
// launch_etile: mode 0 = pose+structure reduce, 1 = the whole structure-only step, 2 = a pose+structure step's last kernel
int etile_precision_bytes(const PlanDev &pd);
int launch_etile(const PlanDev &pd, const StepArgs &a, int mode, int do_poses, int extra_blocks, int zero_blocks, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
// ev != nullptr: a (start, stop) event pair per kernel; *ran gets bit k set for every kernel k that was launched
// fuse_so_poses >= 0 (and `fused` given): a structure-only step on the k_tile path also does the step's update in the same
// launch (fuse_so_poses = 1: copy the poses too) and sets *fused; the caller then skips launch_solve_update
int launch_reduce(const PlanDev &pd, const StepArgs &a, size_t zero_doubles, bool so, hipStream_t st, hipEvent_t *ev = nullptr, unsigned *ran = nullptr,
                  int fuse_so_poses = -1, bool *fused = nullptr);
// dense [S | y] <-> its non-zero blocks in factor order (bt_ba_pack / bt_ba_unpack)
int launch_pack(const PlanDev &pd, const StepArgs &a, bool unpack, hipStream_t st);
// one-shot peer-write exchange of the packed [S | y] (ba_kernels.hip: k_xchg_push / k_xchg_pull)
constexpr int kMaxRanks = 16;
size_t xchg_bytes(const PlanDev &pd, int world);
int launch_xchg_push(const PlanDev &pd, const StepArgs &a, void *const *bufs, int world, int rank, long long epoch, hipStream_t st);
int launch_xchg_pull(const PlanDev &pd, const StepArgs &a, void *own, int world, long long epoch, hipStream_t st);
int launch_solve_update(const PlanDev &pd, const StepArgs &a, bool so, bool copy_poses, hipStream_t st, hipEvent_t *ev = nullptr, unsigned *ran = nullptr);
// tracks seen by more than 64 free cameras sit in no tile (ba_loose.hip): their part of the reduce phase / of the last kernel
int launch_loose_reduce(const PlanDev &pd, const StepArgs &a, bool so, hipStream_t st);
int launch_loose_update(const PlanDev &pd, const StepArgs &a, hipStream_t st);
// the dense solver of plans with more than 255 free poses (ba_dense.hip)
int launch_solve_dense(const PlanDev &pd, const StepArgs &a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);

}  // namespace bt
