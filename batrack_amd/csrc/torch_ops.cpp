// torch_ops.cpp — torch.ops.batrack_hip.*: the operator registration the north star asks for, laid over the C ABI of
// include/batrack_ba.h (nothing is computed here).  The reference itself has no torch.ops registrations:
// its boundary is the Python function BA_rgbd_droid (/root/reference/main/backend/ba.py:217), which
// batrack_amd/backend/ba.py implements on top of these operators.  Every operator takes the plan as an integer handle
// (the bt_plan pointer), checks device / dtype / contiguity, fetches PyTorch's CURRENT HIP stream and calls the C entry
// point: no allocation, no synchronisation.
//   batrack_hip::plan_create(Tensor ii, Tensor jj, Tensor kk, int n_buf, int p_tot, int fixedp, int own_lo, int own_hi) -> int
//   batrack_hip::plan_destroy(int plan) -> ()
//   batrack_hip::plan_info(int plan) -> int[]                 (the fields of bt_plan_info, in order)
//   batrack_hip::ba_step(int plan, Tensor ws, Tensor poses, Tensor patches, Tensor mono, int mono_stride, Tensor intrinsics,
//                        Tensor targets, int target_stride, Tensor weights, Tensor poses_out, Tensor patches_out,
//                        float[] bounds, float lmbda, float ep, float alpha, int loss, bool structure_only, int phase,
//                        Tensor? lmbda_per_track=None) -> int
//       phase 0 = the whole step, 1 = bt_ba_reduce, 2 = bt_ba_pack, 3 = bt_ba_unpack, 4 = bt_ba_solve_update
//   batrack_hip::ba_droid(int plan, Tensor ws, Tensor poses, Tensor patches, Tensor patches_monodisp, Tensor intrinsics,
//                         Tensor targets_2d, Tensor weights, float[] bounds, float lmbda, float ep, float alpha, int loss,
//                         bool structure_only, Tensor? lmbda_per_track=None) -> (Tensor, Tensor)
//       one BA_rgbd_droid call with the caller's tensors as they are (ba.py:217-339: poses [1, N, 7], patches [1, P, 3, 1, 1],
//       the prior and the 2-D targets possibly strided views): shape checks, the views the ABI needs, the two output
//       tensors and the step in one operator call — the Python wrapper's dozen tensor operations cost more host time
//       (28 us a call) than a structure-only step takes on the GPU (9 us)
// Built by batrack_amd/_lib.py:build() into batrack_amd/lib/libbatrack_torch.so (g++, host code only).
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>
#include <vector>

#include "../../include/batrack_ba.h"

namespace {

const float *f32(const at::Tensor &t, const char *what) {
    TORCH_CHECK(t.is_cuda(), "batrack_hip: `", what, "` must be on the GPU (no CPU fallback)");
    TORCH_CHECK(t.scalar_type() == at::kFloat, "batrack_hip: `", what, "` must be float32");
    return t.data_ptr<float>();
}

int64_t plan_create(const at::Tensor &ii, const at::Tensor &jj, const at::Tensor &kk, int64_t n_buf, int64_t p_tot,
                    int64_t fixedp, int64_t own_lo, int64_t own_hi) {
    for (const at::Tensor *t : {&ii, &jj, &kk})
        TORCH_CHECK(t->scalar_type() == at::kLong && t->is_contiguous() && t->numel() == ii.numel(),
                    "batrack_hip::plan_create: indices must be contiguous int64 of one length (batrack.py:100-102)");
    if (ii.is_cuda()) c10::hip::getCurrentHIPStream(ii.device().index()).synchronize();   // the indices must be complete
    bt_plan *plan = nullptr;
    const int rc = bt_plan_create(ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(), kk.data_ptr<int64_t>(), ii.numel(), n_buf, p_tot,
                                  fixedp, 0, own_lo, own_hi, ii.is_cuda() ? 1 : 0, 1, &plan);
    TORCH_CHECK(rc == BT_OK, "batrack_hip::plan_create failed with status ", rc, rc == BT_EUNSUPPORTED ?
                " (unsupported graph: more than 255 free poses, a track seen by more than 64 free cameras, or edges of one "
                "track naming different source frames)" : "");
    return reinterpret_cast<int64_t>(plan);
}

void plan_destroy(int64_t plan) { bt_plan_destroy(reinterpret_cast<bt_plan *>(plan)); }

std::vector<int64_t> plan_info(int64_t plan) {
    bt_plan_info I{};
    TORCH_CHECK(bt_plan_get_info(reinterpret_cast<const bt_plan *>(plan), &I) == BT_OK, "batrack_hip::plan_info: bad handle");
    return {I.E, I.n_buf, I.p_tot, I.fixedp, I.n_all, I.n, I.m, I.pairs, I.tiles, I.slots, I.erows, I.max_tile_cams,
            I.nnz_blocks, I.updates, I.workspace_bytes, I.sorted_input};
}

int64_t ba_step(int64_t plan, const at::Tensor &ws, const at::Tensor &poses, const at::Tensor &patches, const at::Tensor &mono,
                int64_t mono_stride, const at::Tensor &intrinsics, const at::Tensor &targets, int64_t target_stride,
                const at::Tensor &weights, const at::Tensor &poses_out, const at::Tensor &patches_out, c10::ArrayRef<double> bounds,
                double lmbda, double ep, double alpha, int64_t loss, bool structure_only, int64_t phase,
                const c10::optional<at::Tensor> &lmbda_per_track) {
    TORCH_CHECK(bounds.size() == 4, "batrack_hip::ba_step: bounds = [x0, y0, x1, y1]");
    TORCH_CHECK(ws.is_cuda() && ws.is_contiguous(), "batrack_hip::ba_step: the workspace must be a contiguous GPU tensor");
    bt_ba_args a{};
    a.poses = f32(poses, "poses"); a.patches = f32(patches, "patches"); a.mono_disp = f32(mono, "mono_disp");
    a.intrinsics = f32(intrinsics, "intrinsics"); a.targets = f32(targets, "targets"); a.weights = f32(weights, "weights");
    a.target_stride = target_stride; a.mono_stride = mono_stride;
    a.poses_out = const_cast<float *>(f32(poses_out, "poses_out")); a.patches_out = const_cast<float *>(f32(patches_out, "patches_out"));
    for (int i = 0; i < 4; ++i) a.bounds[i] = (float)bounds[i];
    a.lmbda_per_track = lmbda_per_track.has_value() ? f32(*lmbda_per_track, "lmbda_per_track") : nullptr;
    if (lmbda_per_track.has_value()) TORCH_CHECK(lmbda_per_track->is_contiguous(), "batrack_hip::ba_step: lmbda_per_track must be contiguous");
    a.lmbda = (float)lmbda; a.ep = (float)ep; a.alpha = (float)alpha; a.loss = (int32_t)loss; a.structure_only = structure_only ? 1 : 0;
    const bt_plan *p = reinterpret_cast<const bt_plan *>(plan);
    void *st = c10::hip::getCurrentHIPStream(ws.device().index()).stream();
    int rc;
    switch (phase) {
        case 0: rc = bt_ba_step(p, &a, ws.data_ptr(), st); break;
        case 1: rc = bt_ba_reduce(p, &a, ws.data_ptr(), st); break;
        case 2: rc = bt_ba_pack(p, &a, ws.data_ptr(), st); break;
        case 3: rc = bt_ba_unpack(p, &a, ws.data_ptr(), st); break;
        case 4: rc = bt_ba_solve_update(p, &a, ws.data_ptr(), st); break;
        default: rc = BT_EINVAL;
    }
    return rc;
}

std::tuple<at::Tensor, at::Tensor> ba_droid(int64_t plan, const at::Tensor &ws, const at::Tensor &poses, const at::Tensor &patches,
                                             const at::Tensor &monodisp, const at::Tensor &intrinsics, const at::Tensor &targets_2d,
                                             const at::Tensor &weights, c10::ArrayRef<double> bounds, double lmbda, double ep, double alpha,
                                             int64_t loss, bool structure_only, const c10::optional<at::Tensor> &lmbda_per_track) {
    const bt_plan *p = reinterpret_cast<const bt_plan *>(plan);
    bt_plan_info I{};
    TORCH_CHECK(bt_plan_get_info(p, &I) == BT_OK, "batrack_hip::ba_droid: bad plan handle");
    TORCH_CHECK(poses.dim() == 3 && poses.size(0) == 1 && poses.size(2) == 7 && poses.size(1) == I.n_buf,
                "poses must wrap a [1, N, 7] tensor of the plan's N pose slots (batch b = 1, ba.py:218)");
    TORCH_CHECK(patches.dim() >= 3 && patches.size(0) == 1 && patches.size(2) == 3 && patches.numel() == 3 * I.p_tot,
                "patches must be [1, P_tot, 3, 1, 1] with the plan's P_tot (patch size 1, batrack.py:45)");
    const int64_t n_buf = I.n_buf, p_tot = I.p_tot, E = I.E;
    const at::Tensor Pc = poses.contiguous(), pat = patches.reshape({p_tot, 3}).contiguous();
    (void)f32(Pc, "poses"); (void)f32(pat, "patches");
    // the caller's prior is a strided view (patches_local[:, :, mid, 2:], batrack.py:866): used in place through mono_stride —
    // only a genuine stride >= 1 (an expanded tensor, stride 0, or a negative stride is materialised)
    at::Tensor mono = monodisp;
    TORCH_CHECK(mono.numel() == p_tot, "patches_monodisp does not match the patch buffer");
    int64_t mstride = 1;
    if (!mono.is_contiguous() && p_tot > 1) {
        int64_t d = -1, nd = 0;
        for (int64_t k = 0; k < mono.dim(); ++k) if (mono.size(k) == p_tot) { d = k; ++nd; }
        if (nd == 1 && mono.stride(d) >= 1) { mstride = mono.stride(d); mono = mono.as_strided({p_tot}, {mstride}, mono.storage_offset()); }
        else mono = mono.reshape({-1}).contiguous();
    } else mono = mono.reshape({-1});
    const at::Tensor intr = intrinsics.reshape({-1, 4}).contiguous();
    TORCH_CHECK(intr.size(0) == n_buf, "intrinsics do not match the pose buffer");
    TORCH_CHECK(targets_2d.size(-1) == 2 && targets_2d.numel() == 2 * E, "targets_2d must be [1, E, 2]");
    at::Tensor tg = targets_2d.is_contiguous() ? targets_2d.reshape({E, 2}) : targets_2d.select(0, 0);     // the caller's view has strides (3, 1): used in place
    if (tg.dim() != 2 || tg.stride(1) != 1) tg = targets_2d.reshape({E, 2}).contiguous();
    const at::Tensor w = weights.reshape({E, 2}).contiguous();
    at::Tensor patches_out = at::empty_like(pat), poses_out = structure_only ? Pc : at::empty_like(Pc);
    bt_ba_args a{};
    a.poses = Pc.data_ptr<float>(); a.patches = pat.data_ptr<float>(); a.mono_disp = f32(mono, "patches_monodisp");
    a.intrinsics = f32(intr, "intrinsics"); a.targets = f32(tg, "targets_2d"); a.weights = f32(w, "weights");
    a.target_stride = tg.stride(0); a.mono_stride = mstride;
    a.poses_out = poses_out.data_ptr<float>(); a.patches_out = patches_out.data_ptr<float>();
    TORCH_CHECK(bounds.size() == 4, "bounds = [x0, y0, x1, y1]");
    for (int i = 0; i < 4; ++i) a.bounds[i] = (float)bounds[i];
    if (lmbda_per_track.has_value()) {
        TORCH_CHECK(lmbda_per_track->is_contiguous() && lmbda_per_track->numel() == I.m, "a lmbda tensor must hold one value per distinct track (ba.py:299-300)");
        a.lmbda_per_track = f32(*lmbda_per_track, "lmbda");
    }
    a.lmbda = (float)lmbda; a.ep = (float)ep; a.alpha = (float)alpha; a.loss = (int32_t)loss; a.structure_only = structure_only ? 1 : 0;
    TORCH_CHECK(ws.is_cuda() && ws.is_contiguous(), "batrack_hip::ba_droid: the workspace must be a contiguous GPU tensor");
    const int rc = bt_ba_step(p, &a, ws.data_ptr(), c10::hip::getCurrentHIPStream(ws.device().index()).stream());
    TORCH_CHECK(rc == BT_OK, "batrack_hip::ba_droid: bt_ba_step failed with status ", rc);
    return {poses_out, patches_out.view({1, p_tot, 3, 1, 1})};
}

}  // namespace

TORCH_LIBRARY(batrack_hip, m) {
    m.def("plan_create(Tensor ii, Tensor jj, Tensor kk, int n_buf, int p_tot, int fixedp, int own_lo, int own_hi) -> int", &plan_create);
    m.def("plan_destroy(int plan) -> ()", &plan_destroy);
    m.def("plan_info(int plan) -> int[]", &plan_info);
    m.def("ba_step(int plan, Tensor ws, Tensor poses, Tensor patches, Tensor mono, int mono_stride, Tensor intrinsics, Tensor targets, "
          "int target_stride, Tensor weights, Tensor(a!) poses_out, Tensor(b!) patches_out, float[] bounds, float lmbda, float ep, "
          "float alpha, int loss, bool structure_only, int phase, Tensor? lmbda_per_track=None) -> int", &ba_step);
    m.def("ba_droid(int plan, Tensor ws, Tensor poses, Tensor patches, Tensor patches_monodisp, Tensor intrinsics, Tensor targets_2d, "
          "Tensor weights, float[] bounds, float lmbda, float ep, float alpha, int loss, bool structure_only, "
          "Tensor? lmbda_per_track=None) -> (Tensor, Tensor)", &ba_droid);
}
