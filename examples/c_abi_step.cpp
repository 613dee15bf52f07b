// examples/c_abi_step.cpp — the C ABI (include/batrack_ba.h) used from a plain HIP host program: no Python, no torch.
// Reads a problem written by tests/test_gpu_c_example.py (raw little-endian arrays), runs one pose+structure step through
// bt_plan_create / bt_ba_workspace_init / bt_ba_step, writes the updated poses and patches back.
//   hipcc --offload-arch=gfx950 -Iinclude examples/c_abi_step.cpp -Lbatrack_amd/lib -lbatrack_ba -Wl,-rpath,$PWD/batrack_amd/lib -o c_abi_step
//   ./c_abi_step problem.bin result.bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "batrack_ba.h"

#define HIP_OK(x) do { if ((x) != hipSuccess) { std::fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 2; } } while (0)
#define BT_CHECK(x) do { const int rc_ = (x); if (rc_ != BT_OK) { std::fprintf(stderr, "%s -> %d\n", #x, rc_); return 3; } } while (0)

template <typename T>
static bool read_vec(std::FILE *f, std::vector<T> &v, size_t n) { v.resize(n); return std::fread(v.data(), sizeof(T), n, f) == n; }
template <typename T>
static T *to_device(const std::vector<T> &v) {
    T *d = nullptr;
    if (hipMalloc(&d, v.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
    if (hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main(int argc, char **argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: %s problem.bin result.bin\n", argv[0]); return 1; }
    std::FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    int64_t hdr[4];                                            // E, n_buf, p_tot, fixedp
    float bounds[4];
    if (std::fread(hdr, sizeof(int64_t), 4, f) != 4 || std::fread(bounds, sizeof(float), 4, f) != 4) return 1;
    const int64_t E = hdr[0], N = hdr[1], P = hdr[2], fixedp = hdr[3];
    std::vector<int64_t> ii, jj, kk;
    std::vector<float> poses, patches, mono, intr, targets3, weights;
    if (!read_vec(f, ii, E) || !read_vec(f, jj, E) || !read_vec(f, kk, E) || !read_vec(f, poses, 7 * N) || !read_vec(f, patches, 3 * P) ||
        !read_vec(f, mono, P) || !read_vec(f, intr, 4 * N) || !read_vec(f, targets3, 3 * E) || !read_vec(f, weights, 2 * E)) return 1;
    std::fclose(f);

    std::printf("batrack ABI version %d for %s\n", bt_version(), bt_target_arch());
    bt_plan *plan = nullptr;
    BT_CHECK(bt_plan_create(ii.data(), jj.data(), kk.data(), E, N, P, fixedp, 0, 0, 0, /*on_device=*/0, /*upload=*/1, &plan));
    bt_plan_info info;
    BT_CHECK(bt_plan_get_info(plan, &info));
    std::printf("plan: %lld edges, %lld tracks, %lld free poses, %lld tiles, Jacobian kernel %d, workspace %lld bytes\n", (long long)info.E,
                (long long)info.m, (long long)info.n, (long long)info.tiles, bt_plan_jacobian_kernel(plan), (long long)info.workspace_bytes);

    void *ws = nullptr;
    HIP_OK(hipMalloc(&ws, bt_plan_workspace_bytes(plan)));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    BT_CHECK(bt_ba_workspace_init(plan, ws, st));

    bt_ba_args a = {};
    a.poses = to_device(poses); a.patches = to_device(patches); a.mono_disp = to_device(mono); a.intrinsics = to_device(intr);
    a.targets = to_device(targets3); a.target_stride = 3; a.weights = to_device(weights);
    float *poses_out = nullptr, *patches_out = nullptr;
    HIP_OK(hipMalloc(&poses_out, poses.size() * sizeof(float)));
    HIP_OK(hipMalloc(&patches_out, patches.size() * sizeof(float)));
    a.poses_out = poses_out; a.patches_out = patches_out;
    for (int c = 0; c < 4; ++c) a.bounds[c] = bounds[c];
    a.lmbda = 1e-4f; a.ep = 10.0f; a.alpha = 0.05f; a.loss = BT_LOSS_HUBER; a.structure_only = 0;
    a.mono_stride = 1; a.lmbda_per_track = nullptr;
    if (!a.poses || !a.patches || !a.mono_disp || !a.intrinsics || !a.targets || !a.weights) return 2;
    BT_CHECK(bt_ba_step(plan, &a, ws, st));
    int32_t status = -1;
    BT_CHECK(bt_ba_status(plan, ws, st, &status));             // synchronises `st`
    std::printf("solver status %d\n", status);

    std::vector<float> po(poses.size()), xo(patches.size());
    HIP_OK(hipMemcpy(po.data(), poses_out, po.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(xo.data(), patches_out, xo.size() * sizeof(float), hipMemcpyDeviceToHost));
    f = std::fopen(argv[2], "wb");
    if (!f) return 1;
    std::fwrite(&status, sizeof(status), 1, f);
    std::fwrite(po.data(), sizeof(float), po.size(), f);
    std::fwrite(xo.data(), sizeof(float), xo.size(), f);
    std::fclose(f);
    bt_plan_destroy(plan);
    bt_plan_pool_trim();
    return 0;
}
