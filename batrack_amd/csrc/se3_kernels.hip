// se3_kernels.hip — element-wise SE(3) operations for gfx950, one thread per group element,
// grid-stride, AoS rows of 7 / 6 / 4 / 3 scalars as the caller holds them.
// Formulas: /root/reference/main/backend/lietorch/include/se3.h:36-142, so3.h:31-190, common.h:7.
#include <hip/hip_runtime.h>

#include "../../include/batrack_ba.h"
#include "../../include/batrack_se3.h"

namespace bt {

template <typename T> struct Quat { T x, y, z, w; };
template <typename T> struct Pose { T t[3]; Quat<T> q; };

template <typename T> __device__ __forceinline__ T tsqrt(T x);
template <> __device__ __forceinline__ float tsqrt(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double tsqrt(double x) { return sqrt(x); }

template <typename T>
__device__ __forceinline__ Quat<T> qunit(Quat<T> q) {
    const T n = (T)1 / tsqrt(q.x*q.x + q.y*q.y + q.z*q.z + q.w*q.w);
    return {q.x*n, q.y*n, q.z*n, q.w*n};
}
template <typename T>
__device__ __forceinline__ Quat<T> qmul(Quat<T> a, Quat<T> b) {
    return { a.w*b.x + a.x*b.w + a.y*b.z - a.z*b.y,
             a.w*b.y - a.x*b.z + a.y*b.w + a.z*b.x,
             a.w*b.z + a.x*b.y - a.y*b.x + a.z*b.w,
             a.w*b.w - a.x*b.x - a.y*b.y - a.z*b.z };
}
template <typename T>
__device__ __forceinline__ void qrot(Quat<T> q, const T *p, T *o) {       // so3.h:55-60
    T ux = q.y*p[2] - q.z*p[1], uy = q.z*p[0] - q.x*p[2], uz = q.x*p[1] - q.y*p[0];
    ux += ux; uy += uy; uz += uz;
    o[0] = p[0] + q.w*ux + (q.y*uz - q.z*uy);
    o[1] = p[1] + q.w*uy + (q.z*ux - q.x*uz);
    o[2] = p[2] + q.w*uz + (q.x*uy - q.y*ux);
}
template <typename T>
__device__ __forceinline__ Pose<T> load_pose(const T *d) {
    Pose<T> g;
    g.t[0] = d[0]; g.t[1] = d[1]; g.t[2] = d[2];
    g.q = qunit(Quat<T>{d[3], d[4], d[5], d[6]});
    return g;
}
template <typename T>
__device__ __forceinline__ void store_pose(T *d, const Pose<T> &g) {
    d[0] = g.t[0]; d[1] = g.t[1]; d[2] = g.t[2]; d[3] = g.q.x; d[4] = g.q.y; d[5] = g.q.z; d[6] = g.q.w;
}
template <typename T> __device__ __forceinline__ Quat<T> qconj(Quat<T> q) { return {-q.x, -q.y, -q.z, q.w}; }

#define BT_GRID_STRIDE(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

template <typename T>
__global__ void k_se3_exp(const T *xi, T *X, int64_t B) {                  // so3.h:153-190, se3.h:134-142
    BT_GRID_STRIDE(i, B) {
        const T *a = xi + 6 * i;
        const T tau[3] = {a[0], a[1], a[2]}, phi[3] = {a[3], a[4], a[5]};
        const T th2 = phi[0]*phi[0] + phi[1]*phi[1] + phi[2]*phi[2], th = tsqrt(th2);
        T imag, real, c1, c2;
        if (th < (T)1e-6) {
            const T th4 = th2 * th2;
            imag = (T)0.5 - th2 / (T)48 + th4 / (T)3840;
            real = (T)1 - th2 / (T)8 + th4 / (T)384;
            c1 = (T)0.5 - th2 / (T)24;
            c2 = (T)(1.0 / 6.0) - th2 / (T)120;
        } else {
            const T s = (T)sin(0.5 * (double)th), c = (T)cos(0.5 * (double)th);
            imag = s / th; real = c;
            c1 = (T)2 * s * s / th2;                                          // (1 - cos th) / th^2 without cancellation
            c2 = th < (T)0.25 ? (T)(1.0 / 6.0) - th2 / (T)120 + th2 * th2 / (T)5040 - th2 * th2 * th2 / (T)362880
                              : (th - (T)sin((double)th)) / (th2 * th);
        }
        Pose<T> g;
        g.q = qunit(Quat<T>{imag * phi[0], imag * phi[1], imag * phi[2], real});
        const T p1[3] = {phi[1]*tau[2] - phi[2]*tau[1], phi[2]*tau[0] - phi[0]*tau[2], phi[0]*tau[1] - phi[1]*tau[0]};
        const T p2[3] = {phi[1]*p1[2] - phi[2]*p1[1], phi[2]*p1[0] - phi[0]*p1[2], phi[0]*p1[1] - phi[1]*p1[0]};
        for (int c = 0; c < 3; ++c) g.t[c] = tau[c] + c1 * p1[c] + c2 * p2[c];
        store_pose(X + 7 * i, g);
    }
}

template <typename T>
__global__ void k_se3_log(const T *X, T *xi, int64_t B) {                  // so3.h:115-151, se3.h:124-132
    BT_GRID_STRIDE(i, B) {
        const Pose<T> g = load_pose(X + 7 * i);
        const T n2 = g.q.x*g.q.x + g.q.y*g.q.y + g.q.z*g.q.z, w = g.q.w;
        T k;
        if (n2 < (T)1e-12) {
            k = (T)2 / w - (T)(2.0 / 3.0) * n2 / (w * w * w);
        } else {
            const T n = tsqrt(n2);
            if ((w < 0 ? -w : w) < (T)1e-6) k = (w > 0 ? (T)3.14159265358979323846 : -(T)3.14159265358979323846) / n;
            else k = (T)2 * (T)atan((double)(n / w)) / n;
        }
        const T phi[3] = {k * g.q.x, k * g.q.y, k * g.q.z};
        const T th2 = phi[0]*phi[0] + phi[1]*phi[1] + phi[2]*phi[2], th = tsqrt(th2), half = (T)0.5 * th;
        const T c2 = th < (T)1e-6 ? (T)(1.0 / 12.0)
                                   : ((T)1 - th * (T)cos((double)half) / ((T)2 * (T)sin((double)half))) / th2;
        const T p1[3] = {phi[1]*g.t[2] - phi[2]*g.t[1], phi[2]*g.t[0] - phi[0]*g.t[2], phi[0]*g.t[1] - phi[1]*g.t[0]};
        const T p2[3] = {phi[1]*p1[2] - phi[2]*p1[1], phi[2]*p1[0] - phi[0]*p1[2], phi[0]*p1[1] - phi[1]*p1[0]};
        T *o = xi + 6 * i;
        for (int c = 0; c < 3; ++c) { o[c] = g.t[c] - (T)0.5 * p1[c] + c2 * p2[c]; o[3 + c] = phi[c]; }
    }
}

template <typename T>
__global__ void k_se3_inv(const T *X, T *Y, int64_t B) {                   // se3.h:36-38
    BT_GRID_STRIDE(i, B) {
        const Pose<T> g = load_pose(X + 7 * i);
        Pose<T> r;
        r.q = qunit(qconj(g.q));
        T tt[3]; qrot(r.q, g.t, tt);
        r.t[0] = -tt[0]; r.t[1] = -tt[1]; r.t[2] = -tt[2];
        store_pose(Y + 7 * i, r);
    }
}

template <typename T>
__global__ void k_se3_mul(const T *X, const T *Y, T *Z, int64_t B) {       // se3.h:45-47
    BT_GRID_STRIDE(i, B) {
        const Pose<T> a = load_pose(X + 7 * i), b = load_pose(Y + 7 * i);
        Pose<T> r;
        r.q = qunit(qmul(a.q, b.q));
        T tt[3]; qrot(a.q, b.t, tt);
        for (int c = 0; c < 3; ++c) r.t[c] = a.t[c] + tt[c];
        store_pose(Z + 7 * i, r);
    }
}

template <typename T, int DIM>
__global__ void k_se3_act(const T *X, const T *p, T *q, int64_t B) {       // se3.h:49-56
    BT_GRID_STRIDE(i, B) {
        const Pose<T> g = load_pose(X + 7 * i);
        const T *pi = p + DIM * i;
        T r[3]; qrot(g.q, pi, r);
        const T h = DIM == 4 ? pi[3] : (T)1;
        T *o = q + DIM * i;
        for (int c = 0; c < 3; ++c) o[c] = r[c] + g.t[c] * h;
        if (DIM == 4) o[3] = pi[3];
    }
}

template <typename T, bool TRANSPOSE>
__global__ void k_se3_adj(const T *X, const T *a, T *b, int64_t B) {       // se3.h:58-86
    BT_GRID_STRIDE(i, B) {
        const Pose<T> g = load_pose(X + 7 * i);
        const T *ai = a + 6 * i;
        T *o = b + 6 * i;
        if (TRANSPOSE) {           // Ad^T a = [R^T a_tau ; R^T (a_tau x t + a_phi)]
            const Quat<T> qi = qconj(g.q);
            const T c[3] = {ai[1]*g.t[2] - ai[2]*g.t[1] + ai[3], ai[2]*g.t[0] - ai[0]*g.t[2] + ai[4], ai[0]*g.t[1] - ai[1]*g.t[0] + ai[5]};
            qrot(qi, ai, o); qrot(qi, c, o + 3);
        } else {                   // Ad a = [R a_tau + t x (R a_phi) ; R a_phi]
            T rt[3], rp[3];
            qrot(g.q, ai, rt); qrot(g.q, ai + 3, rp);
            o[0] = rt[0] + g.t[1]*rp[2] - g.t[2]*rp[1];
            o[1] = rt[1] + g.t[2]*rp[0] - g.t[0]*rp[2];
            o[2] = rt[2] + g.t[0]*rp[1] - g.t[1]*rp[0];
            o[3] = rp[0]; o[4] = rp[1]; o[5] = rp[2];
        }
    }
}

template <typename T>
__global__ void k_se3_matrix(const T *X, T *M, int64_t B) {                // se3.h:69-78
    BT_GRID_STRIDE(i, B) {
        const Pose<T> g = load_pose(X + 7 * i);
        const T x = g.q.x, y = g.q.y, z = g.q.z, w = g.q.w;
        T *o = M + 16 * i;
        o[0] = 1 - 2*(y*y + z*z); o[1] = 2*(x*y - z*w);     o[2] = 2*(x*z + y*w);      o[3] = g.t[0];
        o[4] = 2*(x*y + z*w);     o[5] = 1 - 2*(x*x + z*z); o[6] = 2*(y*z - x*w);      o[7] = g.t[1];
        o[8] = 2*(x*z - y*w);     o[9] = 2*(y*z + x*w);     o[10] = 1 - 2*(x*x + y*y); o[11] = g.t[2];
        o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 1;
    }
}

static inline dim3 grid_for(int64_t B) {
    int64_t nb = (B + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    return dim3((unsigned)nb);
}

}  // namespace bt

using namespace bt;

#define BT_SE3_CHECK(...)                                             \
    const void *ptrs_[] = {__VA_ARGS__};                              \
    for (const void *p_ : ptrs_) if (!p_ && B > 0) return BT_EINVAL;  \
    if (B < 0 || (dtype != 0 && dtype != 1)) return BT_EINVAL;        \
    if (B == 0) return BT_OK;                                         \
    hipStream_t st = static_cast<hipStream_t>(stream);

#define BT_SE3_DISPATCH(KERN, ...)                                                                      \
    if (dtype == 0) hipLaunchKernelGGL((KERN<float>), grid_for(B), dim3(256), 0, st, __VA_ARGS__);      \
    else            hipLaunchKernelGGL((KERN<double>), grid_for(B), dim3(256), 0, st, __VA_ARGS__);     \
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;

#define F(p) static_cast<const float *>(p)
#define D(p) static_cast<const double *>(p)

extern "C" {

int bt_se3_exp(const void *xi, void *X, int64_t B, int dtype, void *stream) {
    BT_SE3_CHECK(xi, X)
    if (dtype == 0) hipLaunchKernelGGL(k_se3_exp<float>, grid_for(B), dim3(256), 0, st, F(xi), static_cast<float *>(X), B);
    else            hipLaunchKernelGGL(k_se3_exp<double>, grid_for(B), dim3(256), 0, st, D(xi), static_cast<double *>(X), B);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}
int bt_se3_log(const void *X, void *xi, int64_t B, int dtype, void *stream) {
    BT_SE3_CHECK(X, xi)
    if (dtype == 0) hipLaunchKernelGGL(k_se3_log<float>, grid_for(B), dim3(256), 0, st, F(X), static_cast<float *>(xi), B);
    else            hipLaunchKernelGGL(k_se3_log<double>, grid_for(B), dim3(256), 0, st, D(X), static_cast<double *>(xi), B);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}
int bt_se3_inv(const void *X, void *Y, int64_t B, int dtype, void *stream) {
    BT_SE3_CHECK(X, Y)
    if (dtype == 0) hipLaunchKernelGGL(k_se3_inv<float>, grid_for(B), dim3(256), 0, st, F(X), static_cast<float *>(Y), B);
    else            hipLaunchKernelGGL(k_se3_inv<double>, grid_for(B), dim3(256), 0, st, D(X), static_cast<double *>(Y), B);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}
int bt_se3_mul(const void *X, const void *Y, void *Z, int64_t B, int dtype, void *stream) {
    BT_SE3_CHECK(X, Y, Z)
    if (dtype == 0) hipLaunchKernelGGL(k_se3_mul<float>, grid_for(B), dim3(256), 0, st, F(X), F(Y), static_cast<float *>(Z), B);
    else            hipLaunchKernelGGL(k_se3_mul<double>, grid_for(B), dim3(256), 0, st, D(X), D(Y), static_cast<double *>(Z), B);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}
int bt_se3_act(const void *X, const void *p, void *q, int64_t B, int dtype, void *stream) {
    BT_SE3_CHECK(X, p, q)
    if (dtype == 0) hipLaunchKernelGGL((k_se3_act<float, 3>), grid_for(B), dim3(256), 0, st, F(X), F(p), static_cast<float *>(q), B);
    else            hipLaunchKernelGGL((k_se3_act<double, 3>), grid_for(B), dim3(256), 0, st, D(X), D(p), static_cast<double *>(q), B);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}
int bt_se3_act4(const void *X, const void *p, void *q, int64_t B, int dtype, void *stream) {
    BT_SE3_CHECK(X, p, q)
    if (dtype == 0) hipLaunchKernelGGL((k_se3_act<float, 4>), grid_for(B), dim3(256), 0, st, F(X), F(p), static_cast<float *>(q), B);
    else            hipLaunchKernelGGL((k_se3_act<double, 4>), grid_for(B), dim3(256), 0, st, D(X), D(p), static_cast<double *>(q), B);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}
int bt_se3_adj(const void *X, const void *a, void *b, int64_t B, int dtype, void *stream) {
    BT_SE3_CHECK(X, a, b)
    if (dtype == 0) hipLaunchKernelGGL((k_se3_adj<float, false>), grid_for(B), dim3(256), 0, st, F(X), F(a), static_cast<float *>(b), B);
    else            hipLaunchKernelGGL((k_se3_adj<double, false>), grid_for(B), dim3(256), 0, st, D(X), D(a), static_cast<double *>(b), B);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}
int bt_se3_adjT(const void *X, const void *a, void *b, int64_t B, int dtype, void *stream) {
    BT_SE3_CHECK(X, a, b)
    if (dtype == 0) hipLaunchKernelGGL((k_se3_adj<float, true>), grid_for(B), dim3(256), 0, st, F(X), F(a), static_cast<float *>(b), B);
    else            hipLaunchKernelGGL((k_se3_adj<double, true>), grid_for(B), dim3(256), 0, st, D(X), D(a), static_cast<double *>(b), B);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}
int bt_se3_matrix(const void *X, void *M, int64_t B, int dtype, void *stream) {
    BT_SE3_CHECK(X, M)
    if (dtype == 0) hipLaunchKernelGGL(k_se3_matrix<float>, grid_for(B), dim3(256), 0, st, F(X), static_cast<float *>(M), B);
    else            hipLaunchKernelGGL(k_se3_matrix<double>, grid_for(B), dim3(256), 0, st, D(X), static_cast<double *>(M), B);
    return hipGetLastError() == hipSuccess ? BT_OK : BT_EHIP;
}

}  // extern "C"
