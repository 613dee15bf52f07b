// Calibration microbenchmarks for MI355X (gfx950): launch floor, effective clock of a
// single-workgroup kernel, s_barrier / LDS round-trip cost, f64 global atomic throughput.
// hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o /tmp/microbench && /tmp/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty(int *p) { if (p == (int *)1) *p = 0; }

__global__ void k_clock(float *out, long long *cyc, int iters) {
    float a = threadIdx.x * 1e-9f + 1.0f, b = 1.000001f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) a = fmaf(a, b, 0.5f);           // dependent chain
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = a; cyc[0] = t1 - t0; cyc[1] = wall_clock64(); }
}

__global__ void k_barrier(float *out, long long *cyc, int iters) {
    __shared__ float s[1024];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    long long t0 = clock64();
    float acc = 0;
    for (int i = 0; i < iters; ++i) {
        acc += s[(threadIdx.x + i) & 1023];
        __syncthreads();
        s[threadIdx.x] = acc;
        __syncthreads();
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = acc; cyc[0] = t1 - t0; }
}

__global__ void k_lds_chain(float *out, long long *cyc, int iters) {
    __shared__ int s[256];
    s[threadIdx.x] = (threadIdx.x * 7 + 1) & 255;
    __syncthreads();
    int idx = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) idx = s[idx];                   // dependent LDS reads
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = idx; cyc[0] = t1 - t0; }
}

__global__ void k_f64_chain(double *out, long long *cyc, int iters) {
    double a = threadIdx.x * 1e-9 + 1.0, b = 1.000001;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) a = fma(a, b, 0.5);
    long long t1 = clock64();
    double r = a;
    long long t2 = clock64();
    for (int i = 0; i < iters / 16; ++i) r = rsqrt(r + 2.0);
    long long t3 = clock64();
    if (threadIdx.x == 0) { out[0] = a + r; cyc[0] = t1 - t0; cyc[1] = t3 - t2; }
}

// reciprocal square root variants for the 6x6 pivots: accuracy against 1/sqrt in double, and the
// cycles of a dependent chain
__device__ __forceinline__ double rs_seed32(double x) { const double y = (double)__builtin_amdgcn_rsqf((float)x); return y * (1.5 - 0.5 * x * y * y); }
__device__ __forceinline__ double rs_hw64(double x) { return __builtin_amdgcn_rsq(x); }
__device__ __forceinline__ double rs_hw64n(double x) { const double y = __builtin_amdgcn_rsq(x); return y * (1.5 - 0.5 * x * y * y); }
__global__ void k_rsq(double *out, long long *cyc, int iters) {
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = threadIdx.x; i < 200000; i += blockDim.x) {
        const double x = exp2((double)(i % 97) - 40.0) * (1.0 + (double)i * 4.7e-6);
        const double ref = 1.0 / sqrt(x);
        e0 = fmax(e0, fabs(rs_seed32(x) - ref) / ref);
        e1 = fmax(e1, fabs(rs_hw64(x) - ref) / ref);
        e2 = fmax(e2, fabs(rs_hw64n(x) - ref) / ref);
    }
    for (int o = 32; o; o >>= 1) { e0 = fmax(e0, __shfl_xor(e0, o)); e1 = fmax(e1, __shfl_xor(e1, o)); e2 = fmax(e2, __shfl_xor(e2, o)); }
    double r = 2.0 + threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) r = rs_seed32(r) + 2.0;
    long long t1 = clock64();
    for (int i = 0; i < iters; ++i) r = rs_hw64(r) + 2.0;
    long long t2 = clock64();
    for (int i = 0; i < iters; ++i) r = rs_hw64n(r) + 2.0;
    long long t3 = clock64();
    if (threadIdx.x == 0) { out[0] = e0; out[1] = e1; out[2] = e2; out[3] = r; cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
}

__global__ void k_atomic(double *dst, int n_addr, int per_thread) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < per_thread; ++i) {
        const int a = (gid * 37 + i * 101) % n_addr;
        atomicAdd(&dst[a], 1.0);
    }
}
__global__ void k_atomic32(float *dst, int n_addr, int per_thread) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < per_thread; ++i) {
        const int a = (gid * 37 + i * 101) % n_addr;
        atomicAdd(&dst[a], 1.0f);
    }
}
__global__ void k_store(double *dst, int n) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < n) dst[gid] = gid;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    float *dout; long long *dcyc; double *dd;
    CK(hipMalloc(&dout, 1024)); CK(hipMalloc(&dcyc, 64)); CK(hipMalloc(&dd, 64 << 20));
    long long hc[4];
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    // launch floor
    for (int grid : {1, 256, 2048}) {
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, 0, (int *)nullptr);
        CK(hipDeviceSynchronize());
        double t0 = now();
        const int N = 2000;
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, 0, (int *)nullptr);
        CK(hipDeviceSynchronize());
        printf("empty kernel grid=%d: %.2f us per launch (back-to-back, one stream)\n", grid, (now() - t0) / N * 1e6);
    }
    {
        const int iters = 20000;
        hipLaunchKernelGGL(k_rsq, dim3(1), dim3(64), 0, 0, dd, dcyc, iters);
        CK(hipDeviceSynchronize());
        double he[4];
        CK(hipMemcpy(he, dd, sizeof(he), hipMemcpyDeviceToHost)); CK(hipMemcpy(hc, dcyc, sizeof(hc), hipMemcpyDeviceToHost));
        printf("rsqrt(double): f32 seed + Newton: max rel err %.2e, %.1f cycles/op (+1 add) | v_rsq_f64: %.2e, %.1f | v_rsq_f64 + Newton: %.2e, %.1f\n",
               he[0], (double)hc[0] / iters, he[1], (double)hc[1] / iters, he[2], (double)hc[2] / iters);
    }
    // effective clock of a single wave / single WG
    for (int rep = 0; rep < 3; ++rep) {
        const int iters = 200000;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, dout, dcyc, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hc, dcyc, 16, hipMemcpyDeviceToHost));
        printf("fp32 dependent fma chain: %.2f cycles/iter (clock64), %.3f ms wall -> %.0f MHz effective, %.2f ns/iter\n",
               (double)hc[0] / iters, ms, hc[0] / (ms * 1e3), ms * 1e6 / iters);
    }
    {
        const int iters = 100000;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_f64_chain, dim3(1), dim3(64), 0, 0, (double *)dout, dcyc, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hc, dcyc, 16, hipMemcpyDeviceToHost));
        printf("fp64 dependent fma chain: %.2f cycles/iter ; fp64 rsqrt chain: %.2f cycles/iter ; wall %.3f ms\n",
               (double)hc[0] / iters, (double)hc[1] / (iters / 16), ms);
    }
    for (int threads : {64, 256, 512, 1024}) {
        const int iters = 20000;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_barrier, dim3(1), dim3(threads), 0, 0, dout, dcyc, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hc, dcyc, 8, hipMemcpyDeviceToHost));
        printf("barrier loop %4d threads: %.1f cycles per (lds read + 2 barriers + lds write) iteration, %.1f ns\n",
               threads, (double)hc[0] / iters, ms * 1e6 / iters);
    }
    {
        const int iters = 100000;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_lds_chain, dim3(1), dim3(64), 0, 0, dout, dcyc, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hc, dcyc, 8, hipMemcpyDeviceToHost));
        printf("dependent LDS read chain: %.1f cycles per read, %.1f ns\n", (double)hc[0] / iters, ms * 1e6 / iters);
    }
    // atomics: 256 blocks x 256 threads x per_thread atomics onto n_addr doubles
    for (int n_addr : {1 << 10, 1 << 14, 1 << 17, 1 << 20}) {
        const int per = 8, blocks = 256;
        hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, 0, dd, n_addr, per);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, 0, dd, n_addr, per);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double n = (double)blocks * 256 * per;
        printf("f64 atomicAdd: %d addrs, %.0fk atomics in %.2f us -> %.1f G atomics/s\n", n_addr, n / 1e3, ms * 1e3, n / (ms * 1e6));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_atomic32, dim3(blocks), dim3(256), 0, 0, (float *)dd, n_addr, per);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("f32 atomicAdd: %d addrs, %.0fk atomics in %.2f us -> %.1f G atomics/s\n", n_addr, n / 1e3, ms * 1e3, n / (ms * 1e6));
    }
    for (int n : {1 << 17, 1 << 20}) {
        hipLaunchKernelGGL(k_store, dim3((n + 255) / 256), dim3(256), 0, 0, dd, n);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_store, dim3((n + 255) / 256), dim3(256), 0, 0, dd, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("plain store of %d doubles: %.2f us (event pair around one launch)\n", n, ms * 1e3);
    }
    return 0;
}
