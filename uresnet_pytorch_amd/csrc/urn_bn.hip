// BatchNorm(+ReLU) over the active-row matrix, and the OutputLayer row gather.
// HBM-bound passes: 16-byte coalesced streaming reads with several rows in flight per
// lane, fp64 per-thread accumulation, fixed-order two-stage column reduction
// (deterministic), one 16-byte write per lane.
// Replaces scn.BatchNormReLU / scn.BatchNormLeakyReLU(leak 0) and scn.OutputLayer
// (reference uresnet/models/uresnet_sparse.py:22-24).
#include "urn_common.h"

#define BN_MAXBLK 1024

extern "C" int64_t urn_bn_scratch_bytes(int c) { return (int64_t)(BN_MAXBLK + 1) * 2 * c * 8 + 256; }

// mode 0: (sum x, sum x^2); mode 1: (sum g, sum g*xhat) with g = dy * (relu ? y>0 : 1)
// VEC = 4: each lane owns 4 consecutive columns (c % 4 == 0, c <= 1024); VEC = 1: scalar columns (c <= 256)
template <int MODE, int VEC>
__global__ __launch_bounds__(256) void k_bn_partial(const float *__restrict__ x, const float *__restrict__ y,
                                                    const float *__restrict__ dy,
                                                    const float *__restrict__ mean,
                                                    const float *__restrict__ invstd, long n, int c, int relu,
                                                    long rows_per_block, double *__restrict__ part)
{
    __shared__ double s0[256 * VEC], s1[256 * VEC];
    const int t = threadIdx.x;
    const int cw = c / VEC;          // lanes per row
    const int R = 256 / cw;          // rows per pass
    const int cg = t % cw, rsub = t / cw;
    const bool active = t < R * cw;
    const long row_begin = (long)blockIdx.x * rows_per_block;
    const long row_end = min(n, row_begin + rows_per_block);
    double a0[VEC], a1[VEC];
    float mu[VEC], is[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        a0[k] = 0.0; a1[k] = 0.0;
        mu[k] = 0.f; is[k] = 0.f;
        if (MODE == 1 && active) { mu[k] = mean[cg * VEC + k]; is[k] = invstd[cg * VEC + k]; }
    }
    if (active) {
        constexpr int U = 4;  // rows in flight per lane
        for (long row = row_begin + rsub; row < row_end; row += (long)U * R) {
            float vx[U][VEC], vy[U][VEC], vd[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                long rr = row + (long)u * R;
                bool ok = rr < row_end;
                long e = rr * c + cg * VEC;
#pragma unroll
                for (int k = 0; k < VEC; ++k) { vx[u][k] = 0.f; vy[u][k] = 0.f; vd[u][k] = 0.f; }
                if (ok) {
                    if (VEC == 4) {
                        f32x4 v = *(const f32x4 *)(x + e);
#pragma unroll
                        for (int k = 0; k < VEC; ++k) vx[u][k] = v[k];
                        if (MODE == 1) {
                            f32x4 w = *(const f32x4 *)(dy + e);
                            f32x4 z = *(const f32x4 *)(y + e);
#pragma unroll
                            for (int k = 0; k < VEC; ++k) { vd[u][k] = w[k]; vy[u][k] = z[k]; }
                        }
                    } else {
                        vx[u][0] = x[e];
                        if (MODE == 1) { vd[u][0] = dy[e]; vy[u][0] = y[e]; }
                    }
                }
                if (MODE == 1 && !ok) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) vx[u][k] = mu[k];  // xhat = 0, g = 0
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    if (MODE == 0) {
                        double v = (double)vx[u][k];
                        a0[k] += v;
                        a1[k] += v * v;
                    } else {
                        float g = vd[u][k];
                        if (relu && !(vy[u][k] > 0.f)) g = 0.f;
                        double xh = ((double)vx[u][k] - (double)mu[k]) * (double)is[k];
                        a0[k] += (double)g;
                        a1[k] += (double)g * xh;
                    }
                }
        }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) { s0[k * 256 + t] = a0[k]; s1[k * 256 + t] = a1[k]; }
    __syncthreads();
    if (t < cw) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            double r0 = 0.0, r1 = 0.0;
            for (int j = 0; j < R; ++j) { r0 += s0[k * 256 + j * cw + t]; r1 += s1[k * 256 + j * cw + t]; }
            part[((long)blockIdx.x * 2 + 0) * c + t * VEC + k] = r0;
            part[((long)blockIdx.x * 2 + 1) * c + t * VEC + k] = r1;
        }
    }
}

// Second stage: 16 columns per block, 16 lanes per column sweep the block partials in a fixed order.
// mode 0 -> mean/invstd (+ running stats); mode 1 -> dgamma/dbeta and the two backward coefficients.
template <int MODE>
__global__ __launch_bounds__(1024) void k_bn_finalize(const double *__restrict__ part, int nblk, long n, int c,
                                                     double eps, float *__restrict__ o0, float *__restrict__ o1,
                                                     float *r0p, float *r1p, double momentum)
{
    __shared__ double s0[1024], s1[1024];
    const int t = threadIdx.x, cl = t & 15, bg = t >> 4, G = blockDim.x >> 4;
    const int col = blockIdx.x * 16 + cl;
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    if (col < c) {
        int b = bg;
        for (; b + G < nblk; b += 2 * G) {   // two partials in flight per lane
            a0 += part[((long)b * 2) * c + col];
            a1 += part[((long)b * 2 + 1) * c + col];
            b0 += part[((long)(b + G) * 2) * c + col];
            b1 += part[((long)(b + G) * 2 + 1) * c + col];
        }
        if (b < nblk) { a0 += part[((long)b * 2) * c + col]; a1 += part[((long)b * 2 + 1) * c + col]; }
    }
    s0[t] = a0 + b0; s1[t] = a1 + b1;
    __syncthreads();
    if (t < 16 && col < c) {
        double v0 = 0.0, v1 = 0.0;
        for (int j = 0; j < G; ++j) { v0 += s0[j * 16 + t]; v1 += s1[j * 16 + t]; }
        if (MODE == 0) {
            double m = n > 0 ? v0 / (double)n : 0.0;
            double v = n > 0 ? v1 / (double)n - m * m : 0.0;
            if (v < 0.0) v = 0.0;
            o0[col] = (float)m;
            o1[col] = (float)(1.0 / sqrt(v + eps));
            if (r0p) r0p[col] = (float)(momentum * r0p[col] + (1.0 - momentum) * m);
            if (r1p) r1p[col] = (float)(momentum * r1p[col] + (1.0 - momentum) * v);
        } else {
            double invn = n > 0 ? 1.0 / (double)n : 0.0;
            o1[col] += (float)v0;           // dbeta  (accumulated, like every parameter gradient)
            o0[col] += (float)v1;           // dgamma
            r0p[col] = (float)(v0 * invn);  // coef[0][col]
            r1p[col] = (float)(v1 * invn);  // coef[1][col]
        }
    }
}

template <int VEC>
__global__ void k_bn_apply_fwd(const float *__restrict__ x, long total, int c, const float *__restrict__ gamma,
                               const float *__restrict__ beta, const float *__restrict__ mean,
                               const float *__restrict__ invstd, int relu, float *__restrict__ y)
{
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (i >= total) return;  // c % VEC == 0 so a vector never straddles rows
    int col = (int)(i % c);
    float v[VEC], o[VEC];
    if (VEC == 4) {
        f32x4 t = *(const f32x4 *)(x + i);
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] = t[k];
    } else {
        v[0] = x[i];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        float a = gamma[col + k] * invstd[col + k];
        float u = fmaf(v[k] - mean[col + k], a, beta[col + k]);
        o[k] = (relu && u < 0.f) ? 0.f : u;
    }
    if (VEC == 4) {
        f32x4 t;
#pragma unroll
        for (int k = 0; k < VEC; ++k) t[k] = o[k];
        *(f32x4 *)(y + i) = t;
    } else {
        y[i] = o[0];
    }
}

template <int VEC>
__global__ void k_bn_apply_bwd(const float *__restrict__ x, const float *__restrict__ y,
                               const float *__restrict__ dy, long total, int c,
                               const float *__restrict__ gamma, const float *__restrict__ mean,
                               const float *__restrict__ invstd, const float *__restrict__ coef, int relu,
                               float *__restrict__ dx)
{
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (i >= total) return;
    int col = (int)(i % c);
    float vx[VEC], vy[VEC], vd[VEC], o[VEC];
    if (VEC == 4) {
        f32x4 a = *(const f32x4 *)(x + i), b = *(const f32x4 *)(y + i), d = *(const f32x4 *)(dy + i);
#pragma unroll
        for (int k = 0; k < VEC; ++k) { vx[k] = a[k]; vy[k] = b[k]; vd[k] = d[k]; }
    } else {
        vx[0] = x[i]; vy[0] = y[i]; vd[0] = dy[i];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        float g = vd[k];
        if (relu && !(vy[k] > 0.f)) g = 0.f;
        float is = invstd[col + k];
        float xh = (vx[k] - mean[col + k]) * is;
        o[k] = gamma[col + k] * is * (g - coef[col + k] - xh * coef[c + col + k]);
    }
    if (VEC == 4) {
        f32x4 t;
#pragma unroll
        for (int k = 0; k < VEC; ++k) t[k] = o[k];
        *(f32x4 *)(dx + i) = t;
    } else {
        dx[i] = o[0];
    }
}

// rows handled by one block of the partial pass, and the number of blocks
static int bn_grid(int64_t n, int c, int vec, long *rows_per_block)
{
    int R = 256 / (c / vec);
    if (R < 1) R = 1;
    long rpb = (long)R * 8;                       // at least 8 rows per lane
    long need = (n + BN_MAXBLK - 1) / BN_MAXBLK;  // cap the block count
    if (rpb < need) rpb = ((need + R - 1) / R) * R;
    int nblk = (int)((n + rpb - 1) / rpb);
    if (nblk < 1) nblk = 1;
    *rows_per_block = rpb;
    return nblk;
}

static inline int bn_vec(int c) { return (c % 4 == 0 && c / 4 <= 256) ? 4 : 1; }

#define URN_BN_SHAPE_OK(c) (bn_vec(c) == 4 || (c) <= 256)

extern "C" int urn_bn_relu_apply(const float *x, int64_t n, int c, const float *gamma, const float *beta,
                                 const float *mean, const float *invstd, int relu, float *y, void *stream)
{
    URN_CHECK_ARG(c > 0 && n >= 0 && gamma && beta && mean && invstd, "bad argument");
    long total = (long)n * c;
    if (total == 0) return URN_OK;
    URN_CHECK_ARG(x && y, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (c % 4 == 0)
        hipLaunchKernelGGL(k_bn_apply_fwd<4>, dim3(urn_cdiv(total / 4, 256)), dim3(256), 0, st, x, total, c, gamma,
                           beta, mean, invstd, relu, y);
    else
        hipLaunchKernelGGL(k_bn_apply_fwd<1>, dim3(urn_cdiv(total, 256)), dim3(256), 0, st, x, total, c, gamma, beta,
                           mean, invstd, relu, y);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

extern "C" int urn_bn_relu_fwd(const float *x, int64_t n, int c, const float *gamma, const float *beta,
                               double eps, int relu, float *y, float *mean, float *invstd,
                               float *running_mean, float *running_var, double momentum, void *scratch,
                               void *stream)
{
    URN_CHECK_ARG(c > 0 && n >= 0 && gamma && beta && mean && invstd && scratch, "bad argument");
    URN_CHECK_ARG(n == 0 || (x && y), "null pointer");
    if (!URN_BN_SHAPE_OK(c)) { urn_set_error("urn_bn_relu_fwd: unsupported channel count %d", c); return URN_EUNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    double *part = (double *)scratch;
    const int vec = bn_vec(c);
    long rpb;
    int nblk = bn_grid(n, c, vec, &rpb);
    if (vec == 4)
        hipLaunchKernelGGL((k_bn_partial<0, 4>), dim3(nblk), dim3(256), 0, st, x, (const float *)nullptr,
                           (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (long)n, c, 0, rpb,
                           part);
    else
        hipLaunchKernelGGL((k_bn_partial<0, 1>), dim3(nblk), dim3(256), 0, st, x, (const float *)nullptr,
                           (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (long)n, c, 0, rpb,
                           part);
    hipLaunchKernelGGL(k_bn_finalize<0>, dim3(urn_cdiv(c, 16)), dim3(nblk > 128 ? 1024 : 256), 0, st, part, nblk, (long)n, c, eps, mean,
                       invstd, running_mean, running_var, momentum);
    URN_LAUNCH_CHECK();
    return urn_bn_relu_apply(x, n, c, gamma, beta, mean, invstd, relu, y, stream);
}

extern "C" int urn_bn_relu_bwd(const float *x, const float *y, const float *dy, int64_t n, int c,
                               const float *gamma, const float *mean, const float *invstd, int relu, float *dx,
                               float *dgamma, float *dbeta, void *scratch, void *stream)
{
    URN_CHECK_ARG(c > 0 && n >= 0 && gamma && mean && invstd && dgamma && dbeta && scratch, "bad argument");
    URN_CHECK_ARG(n == 0 || (x && y && dy && dx), "null pointer");
    if (!URN_BN_SHAPE_OK(c)) { urn_set_error("urn_bn_relu_bwd: unsupported channel count %d", c); return URN_EUNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    double *part = (double *)scratch;
    float *coef = (float *)(part + (long)BN_MAXBLK * 2 * c);  // the slab after the partials
    const int vec = bn_vec(c);
    long rpb;
    int nblk = bn_grid(n, c, vec, &rpb);
    if (vec == 4)
        hipLaunchKernelGGL((k_bn_partial<1, 4>), dim3(nblk), dim3(256), 0, st, x, y, dy, mean, invstd, (long)n, c, relu,
                           rpb, part);
    else
        hipLaunchKernelGGL((k_bn_partial<1, 1>), dim3(nblk), dim3(256), 0, st, x, y, dy, mean, invstd, (long)n, c, relu,
                           rpb, part);
    hipLaunchKernelGGL(k_bn_finalize<1>, dim3(urn_cdiv(c, 16)), dim3(nblk > 128 ? 1024 : 256), 0, st, part, nblk, (long)n, c, 0.0, dgamma,
                       dbeta, coef, coef + c, 0.0);
    long total = (long)n * c;
    if (total > 0) {
        if (c % 4 == 0)
            hipLaunchKernelGGL(k_bn_apply_bwd<4>, dim3(urn_cdiv(total / 4, 256)), dim3(256), 0, st, x, y, dy, total, c,
                               gamma, mean, invstd, coef, relu, dx);
        else
            hipLaunchKernelGGL(k_bn_apply_bwd<1>, dim3(urn_cdiv(total, 256)), dim3(256), 0, st, x, y, dy, total, c,
                               gamma, mean, invstd, coef, relu, dx);
    }
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

// ------------------------------------------------------------ fused BatchNorm pieces --
// (see include/uresnet_hip.h: the partial sums come from the gather-conv epilogues)
__global__ __launch_bounds__(1024) void k_bn_finalize_fwd_f(const double *__restrict__ part, int nblk, long n, int c,
                                                           int ld, double eps, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, float *__restrict__ mean,
                                                           float *__restrict__ invstd, float *__restrict__ scale,
                                                           float *__restrict__ shift, float *rm, float *rv,
                                                           double momentum)
{
    __shared__ double s0[1024], s1[1024];
    const int t = threadIdx.x, cl = t & 15, bg = t >> 4, G = blockDim.x >> 4;
    const int col = blockIdx.x * 16 + cl;
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    if (col < c) {
        int b = bg;
        for (; b + G < nblk; b += 2 * G) {   // two partials in flight per lane
            a0 += part[((long)b * 2) * ld + col];
            a1 += part[((long)b * 2 + 1) * ld + col];
            b0 += part[((long)(b + G) * 2) * ld + col];
            b1 += part[((long)(b + G) * 2 + 1) * ld + col];
        }
        if (b < nblk) { a0 += part[((long)b * 2) * ld + col]; a1 += part[((long)b * 2 + 1) * ld + col]; }
    }
    s0[t] = a0 + b0; s1[t] = a1 + b1;
    __syncthreads();
    if (t < 16 && col < c) {
        double v0 = 0.0, v1 = 0.0;
        for (int j = 0; j < G; ++j) { v0 += s0[j * 16 + t]; v1 += s1[j * 16 + t]; }
        const double m = n > 0 ? v0 / (double)n : 0.0;
        double v = n > 0 ? v1 / (double)n - m * m : 0.0;
        if (v < 0.0) v = 0.0;
        const double is = 1.0 / sqrt(v + eps);
        mean[col] = (float)m;
        invstd[col] = (float)is;
        const float sc = gamma[col] * (float)is;
        scale[col] = sc;
        shift[col] = fmaf(-(float)m, sc, beta[col]);
        if (rm) rm[col] = (float)(momentum * rm[col] + (1.0 - momentum) * m);
        if (rv) rv[col] = (float)(momentum * rv[col] + (1.0 - momentum) * v);
    }
}

__global__ void k_bn_bwd_apply(const float *__restrict__ x, const float *__restrict__ g,
                               const float *__restrict__ extra, long total, int c,
                               const float *__restrict__ gamma, const float *__restrict__ mean,
                               const float *__restrict__ invstd, const float *__restrict__ c0,
                               const float *__restrict__ c1, float *__restrict__ dx)
{
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= total) return;  // c % 4 == 0
    const int col = (int)(i % c);
    const f32x4 xv = *(const f32x4 *)(x + i), gv = *(const f32x4 *)(g + i);
    f32x4 ev = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (extra) ev = *(const f32x4 *)(extra + i);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float is = invstd[col + k];
        const float xh = (xv[k] - mean[col + k]) * is;
        o[k] = gamma[col + k] * is * (gv[k] - c0[col + k] - xh * c1[col + k]) + ev[k];
    }
    *(f32x4 *)(dx + i) = o;
}

extern "C" int urn_bn_stats_partial(const float *x, int64_t n, int c, double *part, int *n_part, void *stream)
{
    URN_CHECK_ARG(c > 0 && n >= 0 && part && n_part && (n == 0 || x), "bad argument");
    if (!URN_BN_SHAPE_OK(c)) { urn_set_error("urn_bn_stats_partial: unsupported channel count %d", c); return URN_EUNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    const int vec = bn_vec(c);
    long rpb;
    int nblk = bn_grid(n, c, vec, &rpb);
    if (vec == 4)
        hipLaunchKernelGGL((k_bn_partial<0, 4>), dim3(nblk), dim3(256), 0, st, x, (const float *)nullptr, (const float *)nullptr,
                           (const float *)nullptr, (const float *)nullptr, (long)n, c, 0, rpb, part);
    else
        hipLaunchKernelGGL((k_bn_partial<0, 1>), dim3(nblk), dim3(256), 0, st, x, (const float *)nullptr, (const float *)nullptr,
                           (const float *)nullptr, (const float *)nullptr, (long)n, c, 0, rpb, part);
    *n_part = nblk;
    URN_LAUNCH_CHECK();
    return URN_OK;
}

extern "C" int urn_bn_finalize_fwd(const double *part, int n_part, int64_t n, int c, int part_ld, double eps,
                                   const float *gamma, const float *beta, float *mean, float *invstd, float *scale,
                                   float *shift, float *running_mean, float *running_var, double momentum,
                                   void *stream)
{
    URN_CHECK_ARG(part && n_part >= 0 && c > 0 && part_ld >= c && gamma && beta && mean && invstd && scale && shift, "bad argument");
    hipLaunchKernelGGL(k_bn_finalize_fwd_f, dim3(urn_cdiv(c, 16)), dim3(n_part > 128 ? 1024 : 256), 0, (hipStream_t)stream, part, n_part, (long)n, c,
                       part_ld, eps, gamma, beta, mean, invstd, scale, shift, running_mean, running_var, momentum);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

extern "C" int urn_bn_finalize_bwd(const double *part, int n_part, int64_t n, int c, float *dgamma, float *dbeta,
                                   float *coef0, float *coef1, void *stream)
{
    URN_CHECK_ARG(part && n_part >= 0 && c > 0 && dgamma && dbeta && coef0 && coef1, "bad argument");
    hipLaunchKernelGGL(k_bn_finalize<1>, dim3(urn_cdiv(c, 16)), dim3(n_part > 128 ? 1024 : 256), 0, (hipStream_t)stream, part, n_part, (long)n, c, 0.0,
                       dgamma, dbeta, coef0, coef1, 0.0);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

extern "C" int urn_bn_bwd_apply(const float *x, const float *g, const float *extra, int64_t n, int c,
                                const float *gamma, const float *mean, const float *invstd, const float *coef0,
                                const float *coef1, float *dx, void *stream)
{
    URN_CHECK_ARG(c > 0 && c % 4 == 0 && n >= 0 && gamma && mean && invstd && coef0 && coef1, "bad argument (c must be a multiple of 4)");
    const long total = (long)n * c;
    if (total == 0) return URN_OK;
    URN_CHECK_ARG(x && g && dx, "null pointer");
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(urn_cdiv(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, g, extra, total, c,
                       gamma, mean, invstd, coef0, coef1, dx);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

// bn_bwd_apply with the two coefficients taken from an accumulated slab ([slots][2][c], gather-conv epilogue 2 with
// part_slots): every block first reduces the slots into LDS, then streams its share of the elements.  Block 0
// accumulates dgamma/dbeta.  EPB elements per block so that the slab re-read stays small beside the stream.
#define URN_APPLY_EPB 1024
__global__ __launch_bounds__(256) void k_bn_bwd_apply_sums(const float *__restrict__ x, const float *__restrict__ g,
                                                           const float *__restrict__ extra, long ld_extra, long total, int c,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ mean,
                                                           const float *__restrict__ invstd,
                                                           const double *__restrict__ sums, int slots, double inv_n,
                                                           float *dgamma, float *dbeta, float *__restrict__ dx)
{
    __shared__ float s_c0[512], s_c1[512], s_a[512], s_mu[512], s_is[512];
    constexpr int IT = URN_APPLY_EPB / 1024;
    // the element stream is issued first: it is in flight while the slab is reduced
    const long base = (long)blockIdx.x * URN_APPLY_EPB;
    f32x4 xv[IT], gv[IT], ev[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const long i = base + ((long)it * 256 + threadIdx.x) * 4;
        xv[it] = gv[it] = ev[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (i < total) {   // c % 4 == 0: a vector never straddles rows
            xv[it] = *(const f32x4 *)(x + i); gv[it] = *(const f32x4 *)(g + i);
            if (extra) {
                const long row = i / c;                           // extra may be a column block of a wider matrix
                ev[it] = *(const f32x4 *)(extra + row * ld_extra + (i - row * c));
            }
        }
    }
    for (int e = threadIdx.x; e < c; e += 256) {
        const float is = invstd[e], mu = mean[e], ga = gamma[e];   // requested together with the slab rows
        double v0, v1;
        urn_slab_sum2(sums + e, c, slots, v0, v1);
        s_c0[e] = (float)(v0 * inv_n);
        s_c1[e] = (float)(v1 * inv_n);
        s_is[e] = is; s_mu[e] = mu; s_a[e] = ga * is;
        if (blockIdx.x == 0) { dbeta[e] += (float)v0; dgamma[e] += (float)v1; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const long i = base + ((long)it * 256 + threadIdx.x) * 4;
        if (i >= total) break;
        const int col = (int)(i % c);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xv[it][k] - s_mu[col + k]) * s_is[col + k];
            o[k] = s_a[col + k] * (gv[it][k] - s_c0[col + k] - xh * s_c1[col + k]) + ev[it][k];
        }
        *(f32x4 *)(dx + i) = o;
    }
}

extern "C" int urn_bn_bwd_apply_sums(const float *x, const float *g, const float *extra, int64_t ld_extra, int64_t n, int c,
                                     const float *gamma, const float *mean, const float *invstd, const double *sums,
                                     int slots, float *dgamma, float *dbeta, float *dx, void *stream)
{
    if (ld_extra <= 0) ld_extra = c;
    URN_CHECK_ARG(ld_extra >= c && ld_extra % 4 == 0, "ld_extra smaller than the row or not a multiple of 4");
    URN_CHECK_ARG(c > 0 && c % 4 == 0 && c <= 512 && n >= 0 && gamma && mean && invstd && sums && slots > 0 && dgamma && dbeta,
                  "bad argument (c must be a multiple of 4, at most 512)");
    const long total = (long)n * c;
    if (total == 0) return URN_OK;
    URN_CHECK_ARG(x && g && dx, "null pointer");
    hipLaunchKernelGGL(k_bn_bwd_apply_sums, dim3(urn_cdiv(total, URN_APPLY_EPB)), dim3(256), 0, (hipStream_t)stream, x, g, extra,
                       (long)ld_extra, total, c, gamma, mean, invstd, sums, slots, 1.0 / (double)n, dgamma, dbeta, dx);
    URN_LAUNCH_CHECK();
    return URN_OK;
}
