// Row passes of the dense U-ResNet around its convolutions (reference uresnet/models/uresnet_dense.py:72-83: every
// convolution is followed by a batch-statistics BatchNorm; a ResNetModule ends in relu(shortcut + residual), :82):
//
//   forward   out = [relu]( raw * scale + shift  [+ res | + res * res_scale + res_shift] )
//             ONE pass for BatchNorm-apply + residual add + ReLU; the shortcut branch is either the module's input (identity)
//             or the raw output of the 1x1 shortcut convolution with its own BatchNorm folded in.  The statistics behind
//             scale / shift come from the producing convolution's epilogue (urn_dense_conv: stats), not from a pass of
//             their own.
//   backward  g = d_out * [out > 0];  reduce: (sum g, sum g * xhat) per channel for the main BatchNorm and, with a shortcut
//             BatchNorm, (sum g, sum g * xhat_s) for it -- one pass over d_out, out, raw (, res_raw);
//             apply:  d_raw = gamma * invstd * (g - c0 - xhat * c1),  d_res = the same for the shortcut BatchNorm, or g itself
//             for an identity shortcut -- one pass, two outputs.
//
// Row matrices (n, c) fp32, dense rows; c % 4 == 0 and 256 % (c / 4) == 0 (a thread keeps its four channels while it strides
// over the rows, so the per-channel constants live in registers).  Partial sums are fp64, combined per workgroup through LDS
// and added with fp64 atomics into slot (workgroup % slots) of a [slots][2][c] slab (the layout urn_bn_finalize_* read).
#include "urn_common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float *p) { return *(const f32x4 *)p; }

__global__ __launch_bounds__(256) void k_dbn_fwd(const float *__restrict__ raw, const float *__restrict__ scale,
                                                 const float *__restrict__ shift, const float *__restrict__ res,
                                                 const float *__restrict__ res_scale, const float *__restrict__ res_shift, int relu,
                                                 float *__restrict__ out, long total4, int c4)
{
    long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    const int k4 = (int)(e % c4);
    const f32x4 a = ld4(scale + 4 * k4), b = ld4(shift + 4 * k4);
    f32x4 ra = (f32x4){1.f, 1.f, 1.f, 1.f}, rb = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (res_scale) { ra = ld4(res_scale + 4 * k4); rb = ld4(res_shift + 4 * k4); }
    for (; e < total4; e += 4 * stride) {
        f32x4 v[4], r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long eu = e + u * stride < total4 ? e + u * stride : e;
            v[u] = ld4(raw + 4 * eu);
            if (res) r[u] = ld4(res + 4 * eu);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (e + u * stride >= total4) break;
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t = fmaf(v[u][k], a[k], b[k]);
                if (res) t += fmaf(r[u][k], ra[k], rb[k]);
                o[k] = relu ? fmaxf(t, 0.f) : t;
            }
            *(f32x4 *)(out + 4 * (e + u * stride)) = o;
        }
    }
}

// sums[slot][0][c] += sum g, sums[slot][1][c] += sum g * xhat; res_sums likewise with xhat of the shortcut BatchNorm
__global__ __launch_bounds__(256) void k_dbn_bwd_reduce(const float *__restrict__ dout, const float *__restrict__ out,
                                                        const float *__restrict__ raw, const float *__restrict__ mean,
                                                        const float *__restrict__ invstd, const float *__restrict__ res_raw,
                                                        const float *__restrict__ res_mean, const float *__restrict__ res_invstd,
                                                        long total4, int c4, double *sums, double *res_sums, int slots)
{
    __shared__ double s_red[12 * 256];
    long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    const int k4 = (int)(e % c4), c = 4 * c4;
    const f32x4 mu = ld4(mean + 4 * k4), is = ld4(invstd + 4 * k4);
    f32x4 rmu = (f32x4){0.f, 0.f, 0.f, 0.f}, ris = rmu;
    if (res_raw) { rmu = ld4(res_mean + 4 * k4); ris = ld4(res_invstd + 4 * k4); }
    double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    for (; e < total4; e += 2 * stride) {
        f32x4 g[2], o[2], x[2], xr[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long eu = e + u * stride < total4 ? e + u * stride : e;
            g[u] = ld4(dout + 4 * eu);
            if (out) o[u] = ld4(out + 4 * eu);
            x[u] = ld4(raw + 4 * eu);
            if (res_raw) xr[u] = ld4(res_raw + 4 * eu);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (e + u * stride >= total4) break;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gv = (out && !(o[u][k] > 0.f)) ? 0.f : g[u][k];
                s0[k] += (double)gv;
                s1[k] += (double)gv * (((double)x[u][k] - (double)mu[k]) * (double)is[k]);
                if (res_raw) s2[k] += (double)gv * (((double)xr[u][k] - (double)rmu[k]) * (double)ris[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        s_red[(0 + k) * 256 + threadIdx.x] = s0[k];
        s_red[(4 + k) * 256 + threadIdx.x] = s1[k];
        s_red[(8 + k) * 256 + threadIdx.x] = s2[k];
    }
    __syncthreads();
    if ((int)threadIdx.x < c4) {
        double r[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) r[k] = 0.0;
        for (int j = threadIdx.x; j < 256; j += c4)
#pragma unroll
            for (int k = 0; k < 12; ++k) r[k] += s_red[k * 256 + j];
        const long slot = blockIdx.x % (unsigned)slots;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col = 4 * threadIdx.x + k;
            unsafeAtomicAdd(&sums[(slot * 2 + 0) * c + col], r[k]);
            unsafeAtomicAdd(&sums[(slot * 2 + 1) * c + col], r[4 + k]);
            if (res_sums) {
                unsafeAtomicAdd(&res_sums[(slot * 2 + 0) * c + col], r[k]);
                unsafeAtomicAdd(&res_sums[(slot * 2 + 1) * c + col], r[8 + k]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_dbn_bwd_apply(const float *__restrict__ dout, const float *__restrict__ out,
                                                       const float *__restrict__ raw, const float *__restrict__ gamma,
                                                       const float *__restrict__ mean, const float *__restrict__ invstd,
                                                       const float *__restrict__ c0, const float *__restrict__ c1,
                                                       const float *__restrict__ res_raw, const float *__restrict__ res_gamma,
                                                       const float *__restrict__ res_mean, const float *__restrict__ res_invstd,
                                                       const float *__restrict__ rc0, const float *__restrict__ rc1,
                                                       float *__restrict__ d_raw, float *__restrict__ d_res, long total4, int c4)
{
    long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    const int k4 = (int)(e % c4);
    const f32x4 mu = ld4(mean + 4 * k4), is = ld4(invstd + 4 * k4), ga = ld4(gamma + 4 * k4), a0 = ld4(c0 + 4 * k4), a1 = ld4(c1 + 4 * k4);
    f32x4 rmu = mu, ris = is, rga = ga, b0 = a0, b1 = a1;
    if (res_raw) { rmu = ld4(res_mean + 4 * k4); ris = ld4(res_invstd + 4 * k4); rga = ld4(res_gamma + 4 * k4); b0 = ld4(rc0 + 4 * k4); b1 = ld4(rc1 + 4 * k4); }
    for (; e < total4; e += 2 * stride) {
        f32x4 g[2], o[2], x[2], xr[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long eu = e + u * stride < total4 ? e + u * stride : e;
            g[u] = ld4(dout + 4 * eu);
            if (out) o[u] = ld4(out + 4 * eu);
            x[u] = ld4(raw + 4 * eu);
            if (res_raw) xr[u] = ld4(res_raw + 4 * eu);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (e + u * stride >= total4) break;
            f32x4 dr, ds;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gv = (out && !(o[u][k] > 0.f)) ? 0.f : g[u][k];
                const float xh = (x[u][k] - mu[k]) * is[k];
                dr[k] = ga[k] * is[k] * (gv - a0[k] - xh * a1[k]);
                if (res_raw) {
                    const float xs = (xr[u][k] - rmu[k]) * ris[k];
                    ds[k] = rga[k] * ris[k] * (gv - b0[k] - xs * b1[k]);
                } else {
                    ds[k] = gv;
                }
            }
            *(f32x4 *)(d_raw + 4 * (e + u * stride)) = dr;
            if (d_res) *(f32x4 *)(d_res + 4 * (e + u * stride)) = ds;
        }
    }
}

// one 16-lane group per (BatchNorm, channel): the lanes sweep the slots, a fixed shuffle tree adds them
__global__ __launch_bounds__(256) void k_dbn_bwd_finalize(const double *__restrict__ sums, int nb, int slots, double inv_n, int c,
                                                          float *__restrict__ out)
{
    const int e = (blockIdx.x * 256 + threadIdx.x) >> 4, l = threadIdx.x & 15;
    const bool ok = e < nb * c;
    const int b = ok ? e / c : 0, col = ok ? e - b * c : 0;
    const double *p = sums + (long)b * slots * 2 * c + col;
    double v0 = 0.0, v1 = 0.0;
    for (int s = l; s < slots; s += 16) { v0 += p[(long)(2 * s) * c]; v1 += p[(long)(2 * s + 1) * c]; }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) { v0 += __shfl_xor(v0, m); v1 += __shfl_xor(v1, m); }
    if (!ok || l != 0) return;
    float *o = out + (long)b * 4 * c + col;
    o[0] = (float)v1;              // dgamma = sum g * xhat
    o[c] = (float)v0;              // dbeta  = sum g
    o[2 * c] = (float)(v0 * inv_n);
    o[3 * c] = (float)(v1 * inv_n);
}

// DenseSegmentationLoss of one event (reference uresnet_dense.py:246-258) in one pass over the voxels: per voxel the
// log-sum-exp (kept for the backward), the cross-entropy at the label, the arg-max; sums over the event of
// ce * weight * mask, mask (= data > 1e-6: the non-zero count) and mask * [argmax == label], fp64, one atomic triple per
// workgroup.  acc[0..2] must be zero on entry.
__global__ __launch_bounds__(256) void k_dce_fwd(const float *__restrict__ logits, long ld, const float *__restrict__ label,
                                                 const float *__restrict__ data, const float *__restrict__ weight, long n, int nc,
                                                 float *__restrict__ row_lse, double *acc)
{
    __shared__ double s_red[3 * 256];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long)gridDim.x * 256) {
        const float *p = logits + v * ld;
        float mx = p[0];
        int am = 0;
        for (int c = 1; c < nc; ++c) { const float x = p[c]; if (x > mx) { mx = x; am = c; } }   // first maximum, like torch.argmax
        float se = 0.f;
        for (int c = 0; c < nc; ++c) se += __expf(p[c] - mx);
        const float lse = mx + __logf(se);
        row_lse[v] = lse;
        const int lab = (int)label[v];
        if (data[v] > 0.000001f) {
            const float ce = lse - p[lab];
            a0 += (double)(weight ? ce * weight[v] : ce);
            a1 += 1.0;
            a2 += am == lab ? 1.0 : 0.0;
        }
    }
    s_red[threadIdx.x] = a0; s_red[256 + threadIdx.x] = a1; s_red[512 + threadIdx.x] = a2;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) {
            s_red[threadIdx.x] += s_red[threadIdx.x + m];
            s_red[256 + threadIdx.x] += s_red[256 + threadIdx.x + m];
            s_red[512 + threadIdx.x] += s_red[512 + threadIdx.x + m];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { unsafeAtomicAdd(&acc[0], s_red[0]); unsafeAtomicAdd(&acc[1], s_red[256]); unsafeAtomicAdd(&acc[2], s_red[512]); }
}

__global__ void k_dce_finalize(const double *acc, float *out)
{
    out[0] = (float)(acc[0] / acc[1]);     // 0 / 0 = nan for an event without non-zero voxels, like the reference's division
    out[1] = (float)(acc[2] / acc[1]);
}

// dlogits[v][c] = grad * weight * mask / nnz * (softmax(logits[v])[c] - [c == label])
__global__ __launch_bounds__(256) void k_dce_bwd(const float *__restrict__ logits, long ld, const float *__restrict__ label,
                                                 const float *__restrict__ data, const float *__restrict__ weight,
                                                 const float *__restrict__ row_lse, const double *__restrict__ acc,
                                                 const float *__restrict__ grad, long n, int nc, float *__restrict__ dl)
{
    const float gs = (float)((double)grad[0] / acc[1]);
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long)gridDim.x * 256) {
        const float *p = logits + v * ld;
        float *o = dl + v * nc;
        if (!(data[v] > 0.000001f)) { for (int c = 0; c < nc; ++c) o[c] = 0.f; continue; }
        const float sc = weight ? gs * weight[v] : gs;
        const float lse = row_lse[v];
        const int lab = (int)label[v];
        for (int c = 0; c < nc; ++c) o[c] = sc * (__expf(p[c] - lse) - (c == lab ? 1.f : 0.f));
    }
}

int ew_blocks(long total4)
{
    long b = urn_cdiv(total4, 256L * 8);      // >= 8 row groups per thread where the tensor has them
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

bool ew_shape_ok(int64_t n, int c) { return n >= 0 && c > 0 && c % 4 == 0 && 256 % (c / 4) == 0; }

}   // namespace

extern "C" int urn_dense_bn_act_fwd(const float *raw, const float *scale, const float *shift, const float *res,
                                    const float *res_scale, const float *res_shift, int relu, float *out, int64_t n, int c,
                                    void *stream)
{
    URN_CHECK_ARG(ew_shape_ok(n, c), "c must be a multiple of 4 with 256 % (c / 4) == 0");
    if (n == 0) return URN_OK;
    URN_CHECK_ARG(raw && scale && shift && out && (!res_scale || (res && res_shift)), "null pointer");
    const long total4 = (long)n * (c / 4);
    hipLaunchKernelGGL(k_dbn_fwd, dim3(ew_blocks(total4)), dim3(256), 0, (hipStream_t)stream, raw, scale, shift, res, res_scale,
                       res_shift, relu, out, total4, c / 4);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

extern "C" int urn_dense_bn_act_bwd_reduce(const float *d_out, const float *out, const float *raw, const float *mean,
                                           const float *invstd, const float *res_raw, const float *res_mean,
                                           const float *res_invstd, int64_t n, int c, double *sums, double *res_sums, int slots,
                                           void *stream)
{
    URN_CHECK_ARG(ew_shape_ok(n, c) && slots > 0, "c must be a multiple of 4 with 256 % (c / 4) == 0; slots > 0");
    if (n == 0) return URN_OK;
    URN_CHECK_ARG(d_out && raw && mean && invstd && sums && (!res_raw || (res_mean && res_invstd && res_sums)), "null pointer");
    const long total4 = (long)n * (c / 4);
    hipLaunchKernelGGL(k_dbn_bwd_reduce, dim3(ew_blocks(total4)), dim3(256), 0, (hipStream_t)stream, d_out, out, raw, mean, invstd,
                       res_raw, res_mean, res_invstd, total4, c / 4, sums, res_raw ? res_sums : nullptr, slots);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

extern "C" int urn_dense_bn_bwd_finalize(const double *sums, int nb, int slots, int64_t n, int c, float *out, void *stream)
{
    URN_CHECK_ARG(sums && out && (nb == 1 || nb == 2) && slots > 0 && c > 0 && n >= 0, "bad argument");
    hipLaunchKernelGGL(k_dbn_bwd_finalize, dim3(urn_cdiv((long)nb * c * 16, 256)), dim3(256), 0, (hipStream_t)stream, sums, nb, slots,
                       n > 0 ? 1.0 / (double)n : 0.0, c, out);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

extern "C" int urn_dense_bn_act_bwd_apply(const float *d_out, const float *out, const float *raw, const float *gamma,
                                          const float *mean, const float *invstd, const float *coef0, const float *coef1,
                                          const float *res_raw, const float *res_gamma, const float *res_mean,
                                          const float *res_invstd, const float *res_coef0, const float *res_coef1, float *d_raw,
                                          float *d_res, int64_t n, int c, void *stream)
{
    URN_CHECK_ARG(ew_shape_ok(n, c), "c must be a multiple of 4 with 256 % (c / 4) == 0");
    if (n == 0) return URN_OK;
    URN_CHECK_ARG(d_out && raw && gamma && mean && invstd && coef0 && coef1 && d_raw, "null pointer");
    URN_CHECK_ARG(!res_raw || (res_gamma && res_mean && res_invstd && res_coef0 && res_coef1 && d_res), "null pointer (shortcut BatchNorm)");
    const long total4 = (long)n * (c / 4);
    hipLaunchKernelGGL(k_dbn_bwd_apply, dim3(ew_blocks(total4)), dim3(256), 0, (hipStream_t)stream, d_out, out, raw, gamma, mean,
                       invstd, coef0, coef1, res_raw, res_gamma, res_mean, res_invstd, res_coef0, res_coef1, d_raw, d_res, total4,
                       c / 4);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

extern "C" int urn_dense_ce_fwd(const float *logits, int64_t ld, const float *label, const float *data, const float *weight,
                                int64_t n, int nc, float *row_lse, double *acc, float *out, void *stream)
{
    URN_CHECK_ARG(logits && label && data && row_lse && acc && out && n >= 0 && nc >= 1 && ld >= nc, "bad argument");
    long b = urn_cdiv((long)n, 256L * 4);
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    hipLaunchKernelGGL(k_dce_fwd, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, label, data, weight, (long)n, nc,
                       row_lse, acc);
    hipLaunchKernelGGL(k_dce_finalize, dim3(1), dim3(1), 0, (hipStream_t)stream, (const double *)acc, out);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

extern "C" int urn_dense_ce_bwd(const float *logits, int64_t ld, const float *label, const float *data, const float *weight,
                                const float *row_lse, const double *acc, const float *grad_out, int64_t n, int nc, float *dlogits,
                                void *stream)
{
    URN_CHECK_ARG(logits && label && data && row_lse && acc && grad_out && dlogits && n >= 0 && nc >= 1 && ld >= nc, "bad argument");
    if (n == 0) return URN_OK;
    long b = urn_cdiv((long)n, 256L * 4);
    if (b > 2048) b = 2048;
    hipLaunchKernelGGL(k_dce_bwd, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, label, data, weight, row_lse, acc,
                       grad_out, (long)n, nc, dlogits);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

// ---- batched weight layouts --------------------------------------------------------------------------------------------
// The dense convolution kernels read their weights as [tap][cout_p][cin_p] (forward) and [tap][cin_p][cout_p] (input
// gradient), channel counts zero-padded to multiples of 16; torch keeps nn.Conv's as (cout, cin, taps) and
// nn.ConvTranspose's as (cin, cout, taps).  One launch writes both layouts of EVERY convolution of the model (the Python
// route made them with a permute + contiguous copy per convolution and pass: ~120 launches and 1.2 ms per cfg2 step).
#define URN_WL_MAX 64
struct WLDesc { const float *src; float *fwd, *bwd; int d0, d1, taps, transposed, cin, cout, cin_p, cout_p; };
struct WLDescs { int n; WLDesc d[URN_WL_MAX]; };

// Workgroup = a 16 x 16 tile of (cout, cin) channel pairs with all their taps, staged through LDS so that the source rows
// (taps contiguous, then the second channel index) and BOTH destinations (cin contiguous / cout contiguous) move in 64-byte
// pieces.  The one-element-per-thread form wrote one of the two layouts with a stride of a whole channel row: 0.73 ms per
// cfg2 step for 600 MB.
#define WL_T 16
#define WL_MAXTAPS 27
__global__ __launch_bounds__(256) void k_dense_weight_layouts(WLDescs t)
{
    __shared__ float s_w[WL_T][WL_T][WL_MAXTAPS + 1];      // [co][ci][tap]
    const WLDesc d = t.d[blockIdx.y];
    const int tiles_ci = d.cin_p / WL_T, tiles_co = d.cout_p / WL_T;
    const int taps = d.taps;
    for (int tile = blockIdx.x; tile < tiles_ci * tiles_co; tile += gridDim.x) {      // workgroup-uniform
        const int co0 = (tile / tiles_ci) * WL_T, ci0 = (tile % tiles_ci) * WL_T;
        // load: the 16 x 16 pairs' taps; a pair's taps are contiguous, consecutive pairs along the source's second index too
        for (int e = threadIdx.x; e < WL_T * WL_T * taps; e += 256) {
            const int tt = e % taps, p = e / taps;
            // p runs along the source's contiguous channel index first: Conv (cout, cin, taps) -> ci; ConvTranspose (cin, cout, taps) -> co
            const int a = p / WL_T, b = p % WL_T;
            const int co = d.transposed ? co0 + b : co0 + a, ci = d.transposed ? ci0 + a : ci0 + b;
            float v = 0.f;
            if (co < d.cout && ci < d.cin)
                v = d.src[(d.transposed ? ((long)ci * d.cout + co) : ((long)co * d.cin + ci)) * taps + tt];
            s_w[co - co0][ci - ci0][tt] = v;
        }
        __syncthreads();
        // store: forward [tap][cout_p][cin_p] (ci contiguous), input gradient [tap][cin_p][cout_p] (co contiguous)
        for (int e = threadIdx.x; e < WL_T * WL_T * taps; e += 256) {
            const int x = e % WL_T, y = (e / WL_T) % WL_T, tt = e / (WL_T * WL_T);
            d.fwd[((long)tt * d.cout_p + co0 + y) * d.cin_p + ci0 + x] = s_w[y][x][tt];
            d.bwd[((long)tt * d.cin_p + ci0 + y) * d.cout_p + co0 + x] = s_w[x][y][tt];
        }
        __syncthreads();
    }
}

// descs: n records of 11 int64 each: src, fwd, bwd (device pointers), taps, transposed, cin, cout, cin_p, cout_p, 0, 0
extern "C" int urn_dense_weight_layouts(int n, const int64_t *descs, void *stream)
{
    URN_CHECK_ARG(n >= 0 && (n == 0 || descs), "null pointer");
    for (int base = 0; base < n; base += URN_WL_MAX) {
        WLDescs t;
        t.n = n - base < URN_WL_MAX ? n - base : URN_WL_MAX;
        long big = 0;
        for (int i = 0; i < t.n; ++i) {
            const int64_t *r = descs + (int64_t)(base + i) * 11;
            WLDesc &d = t.d[i];
            d.src = (const float *)(uintptr_t)r[0]; d.fwd = (float *)(uintptr_t)r[1]; d.bwd = (float *)(uintptr_t)r[2];
            d.taps = (int)r[3]; d.transposed = (int)r[4]; d.cin = (int)r[5]; d.cout = (int)r[6]; d.cin_p = (int)r[7]; d.cout_p = (int)r[8];
            d.d0 = d.d1 = 0;
            URN_CHECK_ARG(d.src && d.fwd && d.bwd && d.taps > 0 && d.taps <= WL_MAXTAPS && d.cin > 0 && d.cout > 0 && d.cin_p >= d.cin &&
                          d.cout_p >= d.cout && d.cin_p % 16 == 0 && d.cout_p % 16 == 0, "bad record (taps <= 27, padded channel counts multiples of 16)");
            const long tiles = (long)(d.cin_p / WL_T) * (d.cout_p / WL_T);
            if (tiles > big) big = tiles;
        }
        unsigned gx = (unsigned)(big > 64 ? 64 : (big < 1 ? 1 : big));     // up to 64 workgroups per convolution walk its tiles
        hipLaunchKernelGGL(k_dense_weight_layouts, dim3(gx, t.n), dim3(256), 0, (hipStream_t)stream, t);
    }
    URN_LAUNCH_CHECK();
    return URN_OK;
}
