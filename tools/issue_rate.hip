// VALU issue rate of one SIMD of gfx950 (MI355X), the number DESIGN.md §6's diagnosis of k_edge rests on:
// independent v_fma_f32 / v_fma_f64 / v_pk_fma_f32 streams (16 accumulators, no dependency stall), timed with
// s_memtime inside the wave, with 1, 2 and 4 waves per SIMD (workgroups of 256 / 512 / 1024 threads on one CU).
//   hipcc --offload-arch=gfx950 -O3 tools/issue_rate.hip -o /tmp/issue_rate && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float float2_t __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(1024) void k_issue(float *out, long long *cyc, int iters) {
    float a[16];
    double d[16];
    float2_t p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 1e-9f + i; d[i] = a[i]; p[i] = float2_t{a[i], a[i] + 1.0f}; }
    const float b = 1.000001f, c = 0.5f;
    const double bd = 1.000001, cd = 0.5;
    const float2_t bp = {b, b}, cp = {c, c};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(bd), "v"(cd));
            if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(bp), "v"(cp));
            if (KIND == 3) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        }
    }
    const long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + (float)d[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float *dout; long long *dcyc;
    CK(hipMalloc(&dout, 1024)); CK(hipMalloc(&dcyc, 64));
    const int iters = 4096;
    const char *names[4] = {"v_fma_f32", "v_fma_f64", "v_pk_fma_f32", "v_add_f32_dpp"};
    for (int kind = 0; kind < 4; ++kind)
        for (int cfg = 0; cfg < 4; ++cfg) {
            const int threads = cfg == 0 ? 64 : cfg == 1 ? 256 : cfg == 2 ? 512 : 1024, grid = 1;
            long long hc = 0;
            for (int rep = 0; rep < 3; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(k_issue<0>, dim3(grid), dim3(threads), 0, 0, dout, dcyc, iters);
                if (kind == 1) hipLaunchKernelGGL(k_issue<1>, dim3(grid), dim3(threads), 0, 0, dout, dcyc, iters);
                if (kind == 2) hipLaunchKernelGGL(k_issue<2>, dim3(grid), dim3(threads), 0, 0, dout, dcyc, iters);
                if (kind == 3) hipLaunchKernelGGL(k_issue<3>, dim3(grid), dim3(threads), 0, 0, dout, dcyc, iters);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(&hc, dcyc, 8, hipMemcpyDeviceToHost));
            }
            const double per = (double)hc / ((double)iters * 16.0);
            const int wps = threads <= 256 ? 1 : threads / 256;
            printf("%-14s %4d threads (%d wave%s per SIMD): %.2f cycles per wave64 instruction as seen by one wave -> %.2f cycles of SIMD time per instruction\n",
                   names[kind], threads, wps, wps > 1 ? "s" : "", per, per / wps);
        }
    return 0;
}
