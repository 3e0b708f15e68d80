// BatchNorm(+ReLU) over the active-row matrix, and the OutputLayer row gather.
// HBM-bound passes: coalesced streaming reads, fp64 per-thread accumulation,
// fixed-order two-stage column reduction (deterministic), one write.
// Replaces scn.BatchNormReLU / scn.BatchNormLeakyReLU(leak 0) and scn.OutputLayer
// (reference uresnet/models/uresnet_sparse.py:22-24).
#include "urn_common.h"

#define BN_MAXBLK 256

extern "C" int64_t urn_bn_scratch_bytes(int c) { return (int64_t)BN_MAXBLK * 2 * c * 8 + 256; }

// mode 0: (sum x, sum x^2); mode 1: (sum g, sum g*xhat) with g = dy * (relu ? y>0 : 1)
template <int MODE>
__global__ __launch_bounds__(256) void k_bn_partial(const float *__restrict__ x, const float *__restrict__ y,
                                                    const float *__restrict__ dy,
                                                    const float *__restrict__ mean,
                                                    const float *__restrict__ invstd, long n, int c, int relu,
                                                    long rows_per_block, double *__restrict__ part)
{
    __shared__ double s0[256], s1[256];
    const int t = threadIdx.x;
    const int c0 = blockIdx.y * 256;
    const int cw = min(256, c - c0);
    const int R = 256 / cw;
    const int col = c0 + t % cw, rsub = t / cw;
    const bool active = t < R * cw;
    const long row_begin = (long)blockIdx.x * rows_per_block;
    const long row_end = min(n, row_begin + rows_per_block);
    double a0 = 0.0, a1 = 0.0;
    if (active) {
        float mu = 0.f, is = 0.f;
        if (MODE == 1) { mu = mean[col]; is = invstd[col]; }
        for (long row = row_begin + rsub; row < row_end; row += R) {
            long e = row * c + col;
            if (MODE == 0) {
                double v = (double)x[e];
                a0 += v;
                a1 += v * v;
            } else {
                float g = dy[e];
                if (relu && !(y[e] > 0.f)) g = 0.f;
                double xh = ((double)x[e] - (double)mu) * (double)is;
                a0 += (double)g;
                a1 += (double)g * xh;
            }
        }
    }
    s0[t] = a0; s1[t] = a1;
    __syncthreads();
    if (t < cw) {
        double r0 = 0.0, r1 = 0.0;
        for (int k = 0; k < R; ++k) { r0 += s0[k * cw + t]; r1 += s1[k * cw + t]; }
        part[((long)blockIdx.x * 2 + 0) * c + col] = r0;
        part[((long)blockIdx.x * 2 + 1) * c + col] = r1;
    }
}

__global__ void k_bn_finalize_fwd(const double *__restrict__ part, int nblk, long n, int c, double eps,
                                  float *__restrict__ mean, float *__restrict__ invstd,
                                  float *running_mean, float *running_var, double momentum)
{
    int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= c) return;
    double s = 0.0, ss = 0.0;
    for (int b = 0; b < nblk; ++b) { s += part[((long)b * 2) * c + col]; ss += part[((long)b * 2 + 1) * c + col]; }
    double m = n > 0 ? s / (double)n : 0.0;
    double v = n > 0 ? ss / (double)n - m * m : 0.0;
    if (v < 0.0) v = 0.0;
    mean[col] = (float)m;
    invstd[col] = (float)(1.0 / sqrt(v + eps));
    if (running_mean) running_mean[col] = (float)(momentum * running_mean[col] + (1.0 - momentum) * m);
    if (running_var) running_var[col] = (float)(momentum * running_var[col] + (1.0 - momentum) * v);
}

__global__ void k_bn_apply_fwd(const float *__restrict__ x, long total, int c, const float *__restrict__ gamma,
                               const float *__restrict__ beta, const float *__restrict__ mean,
                               const float *__restrict__ invstd, int relu, float *__restrict__ y)
{
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= total) return;  // c % 4 == 0 so a float4 never straddles rows
    int col = (int)(i % c);
    f32x4 v = *(const f32x4 *)(x + i), o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float a = gamma[col + k] * invstd[col + k];
        float u = fmaf(v[k] - mean[col + k], a, beta[col + k]);
        o[k] = (relu && u < 0.f) ? 0.f : u;
    }
    *(f32x4 *)(y + i) = o;
}

__global__ void k_bn_apply_fwd_scalar(const float *__restrict__ x, long total, int c,
                                      const float *__restrict__ gamma, const float *__restrict__ beta,
                                      const float *__restrict__ mean, const float *__restrict__ invstd,
                                      int relu, float *__restrict__ y)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int col = (int)(i % c);
    float u = fmaf(x[i] - mean[col], gamma[col] * invstd[col], beta[col]);
    y[i] = (relu && u < 0.f) ? 0.f : u;
}

static int bn_grid(int64_t n, long *rows_per_block)
{
    int nblk = (int)((n + 255) / 256);
    if (nblk > BN_MAXBLK) nblk = BN_MAXBLK;
    if (nblk < 1) nblk = 1;
    *rows_per_block = (long)((n + nblk - 1) / nblk);
    return nblk;
}

extern "C" int urn_bn_relu_fwd(const float *x, int64_t n, int c, const float *gamma, const float *beta,
                               double eps, int relu, float *y, float *mean, float *invstd,
                               float *running_mean, float *running_var, double momentum, void *scratch,
                               void *stream)
{
    URN_CHECK_ARG(c > 0 && n >= 0 && gamma && beta && mean && invstd && scratch, "bad argument");
    URN_CHECK_ARG(n == 0 || (x && y), "null pointer");
    hipStream_t st = (hipStream_t)stream;
    double *part = (double *)scratch;
    long rpb;
    int nblk = bn_grid(n, &rpb);
    hipLaunchKernelGGL(k_bn_partial<0>, dim3(nblk, urn_cdiv(c, 256)), dim3(256), 0, st, x, (const float *)nullptr,
                       (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (long)n, c, 0, rpb,
                       part);
    hipLaunchKernelGGL(k_bn_finalize_fwd, dim3(urn_cdiv(c, 64)), dim3(64), 0, st, part, nblk, (long)n, c, eps, mean,
                       invstd, running_mean, running_var, momentum);
    long total = (long)n * c;
    if (total > 0) {
        if (c % 4 == 0)
            hipLaunchKernelGGL(k_bn_apply_fwd, dim3(urn_cdiv(total / 4, 256)), dim3(256), 0, st, x, total, c, gamma,
                               beta, mean, invstd, relu, y);
        else
            hipLaunchKernelGGL(k_bn_apply_fwd_scalar, dim3(urn_cdiv(total, 256)), dim3(256), 0, st, x, total, c,
                               gamma, beta, mean, invstd, relu, y);
    }
    URN_LAUNCH_CHECK();
    return URN_OK;
}

extern "C" int urn_bn_relu_apply(const float *x, int64_t n, int c, const float *gamma, const float *beta,
                                 const float *mean, const float *invstd, int relu, float *y, void *stream)
{
    URN_CHECK_ARG(c > 0 && n >= 0 && gamma && beta && mean && invstd, "bad argument");
    long total = (long)n * c;
    if (total == 0) return URN_OK;
    URN_CHECK_ARG(x && y, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (c % 4 == 0)
        hipLaunchKernelGGL(k_bn_apply_fwd, dim3(urn_cdiv(total / 4, 256)), dim3(256), 0, st, x, total, c, gamma, beta,
                           mean, invstd, relu, y);
    else
        hipLaunchKernelGGL(k_bn_apply_fwd_scalar, dim3(urn_cdiv(total, 256)), dim3(256), 0, st, x, total, c, gamma,
                           beta, mean, invstd, relu, y);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

__global__ void k_bn_finalize_bwd(const double *__restrict__ part, int nblk, long n, int c,
                                  float *__restrict__ dgamma, float *__restrict__ dbeta,
                                  float *__restrict__ coef /* [2][c]: sb/n, sg/n */)
{
    int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= c) return;
    double sb = 0.0, sg = 0.0;
    for (int b = 0; b < nblk; ++b) { sb += part[((long)b * 2) * c + col]; sg += part[((long)b * 2 + 1) * c + col]; }
    dbeta[col] = (float)sb;
    dgamma[col] = (float)sg;
    double invn = n > 0 ? 1.0 / (double)n : 0.0;
    coef[col] = (float)(sb * invn);
    coef[c + col] = (float)(sg * invn);
}

__global__ void k_bn_apply_bwd(const float *__restrict__ x, const float *__restrict__ y,
                               const float *__restrict__ dy, long total, int c,
                               const float *__restrict__ gamma, const float *__restrict__ mean,
                               const float *__restrict__ invstd, const float *__restrict__ coef, int relu,
                               float *__restrict__ dx)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int col = (int)(i % c);
    float g = dy[i];
    if (relu && !(y[i] > 0.f)) g = 0.f;
    float is = invstd[col];
    float xh = (x[i] - mean[col]) * is;
    dx[i] = gamma[col] * is * (g - coef[col] - xh * coef[c + col]);
}

extern "C" int urn_bn_relu_bwd(const float *x, const float *y, const float *dy, int64_t n, int c,
                               const float *gamma, const float *mean, const float *invstd, int relu, float *dx,
                               float *dgamma, float *dbeta, void *scratch, void *stream)
{
    URN_CHECK_ARG(c > 0 && n >= 0 && gamma && mean && invstd && dgamma && dbeta && scratch, "bad argument");
    URN_CHECK_ARG(n == 0 || (x && y && dy && dx), "null pointer");
    hipStream_t st = (hipStream_t)stream;
    double *part = (double *)scratch;
    // coefficients live after the partials (scratch has 256 spare bytes only for alignment, so
    // carve them from the partial area's tail: nblk <= BN_MAXBLK-1 keeps one slab free)
    long rpb;
    int nblk = bn_grid(n, &rpb);
    if (nblk == BN_MAXBLK) { nblk = BN_MAXBLK - 1; rpb = (long)((n + nblk - 1) / nblk); }
    float *coef = (float *)(part + (long)(BN_MAXBLK - 1) * 2 * c);
    hipLaunchKernelGGL(k_bn_partial<1>, dim3(nblk, urn_cdiv(c, 256)), dim3(256), 0, st, x, y, dy, mean, invstd,
                       (long)n, c, relu, rpb, part);
    hipLaunchKernelGGL(k_bn_finalize_bwd, dim3(urn_cdiv(c, 64)), dim3(64), 0, st, part, nblk, (long)n, c, dgamma,
                       dbeta, coef);
    long total = (long)n * c;
    if (total > 0)
        hipLaunchKernelGGL(k_bn_apply_bwd, dim3(urn_cdiv(total, 256)), dim3(256), 0, st, x, y, dy, total, c, gamma,
                           mean, invstd, coef, relu, dx);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

// ------------------------------------------------------------------- OutputLayer --
__global__ void k_rows_gather(const float *__restrict__ x, const int *__restrict__ idx, long n, int c,
                              float *__restrict__ y)
{
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * c) return;
    long i = t / c;
    int k = (int)(t - i * c);
    y[t] = x[(long)idx[i] * c + k];
}

__global__ void k_rows_scatter_add(const float *__restrict__ dy, const int *__restrict__ idx, long n, int c,
                                   float *__restrict__ dx)
{
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * c) return;
    long i = t / c;
    int k = (int)(t - i * c);
    atomicAdd(&dx[(long)idx[i] * c + k], dy[t]);
}

extern "C" int urn_rows_gather(const float *x, const int32_t *idx, int64_t n, int c, float *y, void *stream)
{
    if (n <= 0) return URN_OK;
    URN_CHECK_ARG(x && idx && y && c > 0, "bad argument");
    hipLaunchKernelGGL(k_rows_gather, dim3(urn_cdiv(n * c, 256)), dim3(256), 0, (hipStream_t)stream, x, idx, (long)n,
                       c, y);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

extern "C" int urn_rows_scatter_add(const float *dy, const int32_t *idx, int64_t n, int c, float *dx, void *stream)
{
    if (n <= 0) return URN_OK;
    URN_CHECK_ARG(dy && idx && dx && c > 0, "bad argument");
    hipLaunchKernelGGL(k_rows_scatter_add, dim3(urn_cdiv(n * c, 256)), dim3(256), 0, (hipStream_t)stream, dy, idx,
                       (long)n, c, dx);
    URN_LAUNCH_CHECK();
    return URN_OK;
}
