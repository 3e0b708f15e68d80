// Network head and loss on the device: scn.OutputLayer + torch.nn.Linear
// (reference uresnet/models/uresnet_sparse.py:24-25,36) as one gather+GEMV kernel, and the per-event
// mean cross-entropy / accuracy of SegmentationLoss (reference uresnet_sparse.py:46-82) without a
// host synchronisation per event.  HBM-bound streaming kernels; N x 16 x 5 is far too small for MFMA.
#include "urn_common.h"

#define HEAD_MAXC 32     // classes
#define HEAD_MAXM 256    // trunk width
#define CE_MAXEV 1024    // batch ids handled by the loss

// logits[i,:] = x[row2site[i],:] @ W^T + b          W is (nc, m) like torch.nn.Linear.weight
__global__ __launch_bounds__(256) void k_head_fwd(const float *__restrict__ x, const int *__restrict__ row2site, long n,
                                                  int m, int nc, const float *__restrict__ W, const float *__restrict__ b,
                                                  float *__restrict__ logits)
{
    __shared__ float s_w[HEAD_MAXC * HEAD_MAXM / 8];   // nc*m <= 1024 floats
    __shared__ float s_b[HEAD_MAXC];
    for (int e = threadIdx.x; e < nc * m; e += 256) s_w[e] = W[e];
    if (threadIdx.x < nc) s_b[threadIdx.x] = b ? b[threadIdx.x] : 0.f;
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *xi = x + (row2site ? (long)row2site[i] : i) * m;
    float acc[HEAD_MAXC];
#pragma unroll
    for (int c = 0; c < HEAD_MAXC; ++c) acc[c] = 0.f;
    for (int k = 0; k < m; k += 4) {
        const f32x4 v = *(const f32x4 *)(xi + k);
#pragma unroll
        for (int c = 0; c < HEAD_MAXC; ++c)
            if (c < nc) {
                const float *w = s_w + c * m + k;
                acc[c] = fmaf(v[0], w[0], fmaf(v[1], w[1], fmaf(v[2], w[2], fmaf(v[3], w[3], acc[c]))));
            }
    }
#pragma unroll
    for (int c = 0; c < HEAD_MAXC; ++c)
        if (c < nc) logits[i * nc + c] = acc[c] + s_b[c];
}

// dx[row2site[i],:] += dl[i,:] @ W ; dW += dl^T @ x_rows ; db += sum dl        (all accumulated; caller zeroes)
__global__ __launch_bounds__(256) void k_head_bwd(const float *__restrict__ dl, const float *__restrict__ x,
                                                  const int *__restrict__ row2site, long n, int m, int nc,
                                                  const float *__restrict__ W, float *__restrict__ dx,
                                                  float *__restrict__ dW, float *__restrict__ db)
{
    __shared__ float s_w[HEAD_MAXC * HEAD_MAXM / 8];
    __shared__ float s_dl[256][HEAD_MAXC / 4 + 1];      // nc <= 8 fast path rows of 9 floats
    __shared__ float s_x[256][17];                       // m == 16 fast path
    for (int e = threadIdx.x; e < nc * m; e += 256) s_w[e] = W[e];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    float d[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) d[c] = 0.f;
    long site = -1;
    if (i < n) {
        site = row2site ? (long)row2site[i] : i;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < nc) d[c] = dl[i * nc + c];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) s_dl[threadIdx.x][c] = d[c];
    __syncthreads();
    if (i < n && m == 16) {
        // m == 16 (the network's width): the row and its gradient move as four 16-byte pieces (one float per instruction
        // touched 64 cache lines for 64 floats: 32 such instructions per wave made this kernel 19 us at 50k rows)
        const float *xi = x + site * 16;
        f32x4 xv[4];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) xv[k4] = *(const f32x4 *)(xi + 4 * k4);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            f32x4 gv;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float g = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c < nc) g = fmaf(d[c], s_w[c * 16 + 4 * k4 + kk], g);
                gv[kk] = g;
                s_x[threadIdx.x][4 * k4 + kk] = xv[k4][kk];
            }
            if (row2site) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) atomicAdd(&dx[site * 16 + 4 * k4 + kk], gv[kk]);   // several input rows may share a site
            } else {
                *(f32x4 *)(dx + site * 16 + 4 * k4) = gv;                                          // identity map: plain store
            }
        }
    } else if (i < n) {
        for (int k = 0; k < m; ++k) {
            float g = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < nc) g = fmaf(d[c], s_w[c * m + k], g);
            if (row2site) atomicAdd(&dx[site * m + k], g);   // several input rows may share a site
            else dx[site * m + k] = g;                         // identity map: plain store
        }
    } else if (m == 16) {
        for (int k = 0; k < 16; ++k) s_x[threadIdx.x][k] = 0.f;
    }
    __syncthreads();
    // block-level dW / db: thread e = (c, k) sums over the block's 256 rows
    for (int e = threadIdx.x; e < nc * m; e += 256) {
        const int c = e / m, k = e - c * m;
        float acc = 0.f;
        if (m == 16) {
            for (int r = 0; r < 256; ++r) acc = fmaf(s_dl[r][c], s_x[r][k], acc);
        } else {
            const long base = (long)blockIdx.x * 256;
            for (int r = 0; r < 256 && base + r < n; ++r) acc = fmaf(s_dl[r][c], x[(row2site ? (long)row2site[base + r] : base + r) * m + k], acc);
        }
        atomicAdd(&dW[e], acc);
    }
    if (threadIdx.x < nc) {
        float acc = 0.f;
        for (int r = 0; r < 256; ++r) acc += s_dl[r][threadIdx.x];
        atomicAdd(&db[threadIdx.x], acc);
    }
}

extern "C" int urn_head_fwd(const float *x, const int32_t *row2site, int64_t n, int m, int nc, const float *W,
                            const float *b, float *logits, void *stream)
{
    if (n <= 0) return URN_OK;
    URN_CHECK_ARG(x && W && logits, "null pointer");
    URN_CHECK_ARG(m > 0 && m % 4 == 0 && nc > 0 && nc <= HEAD_MAXC && nc * m <= HEAD_MAXC * HEAD_MAXM / 8, "unsupported head shape");
    hipLaunchKernelGGL(k_head_fwd, dim3(urn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, row2site, (long)n, m, nc, W, b,
                       logits);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

extern "C" int urn_head_bwd(const float *dlogits, const float *x, const int32_t *row2site, int64_t n, int m, int nc,
                            const float *W, float *dx, float *dW, float *db, void *stream)
{
    if (n <= 0) return URN_OK;
    URN_CHECK_ARG(dlogits && x && W && dx && dW && db, "null pointer");
    URN_CHECK_ARG(m > 0 && nc > 0 && nc <= 8 && nc * m <= HEAD_MAXC * HEAD_MAXM / 8, "unsupported head shape (nc <= 8)");
    hipLaunchKernelGGL(k_head_bwd, dim3(urn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dlogits, x, row2site, (long)n, m, nc,
                       W, dx, dW, db);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

// ---- the network's tail as two kernels (executor, urn_net_set_head) ---------------------------------------------
// Forward: last BatchNormReLU (its statistics finalized here from the accumulated slab of the producing convolution, or
// given) + OutputLayer (site -> input row) + Linear in ONE pass: logits[i,:] = relu(x[row2site[i],:] * scale + shift) @ W^T + b.
// The normalised (n0, m) tensor and the (N, m) row matrix are never written (four launches and two passes less).
#define TAIL_MAXM 64
__global__ __launch_bounds__(256) void k_tail_fwd(const float *__restrict__ x, const int *__restrict__ row2site, long n, int m, int nc,
                                                  const float *__restrict__ W, const float *__restrict__ b,
                                                  const double *__restrict__ sums, int slots, long n_sites, double eps,
                                                  const float *__restrict__ gamma, const float *__restrict__ beta,
                                                  float *mean, float *invstd, float *scale, float *shift, float *rm, float *rv,
                                                  double momentum, float *__restrict__ logits)
{
    __shared__ float s_w[HEAD_MAXC * HEAD_MAXM / 8];
    __shared__ float s_b[HEAD_MAXC];
    __shared__ __attribute__((aligned(16))) float s_sc[TAIL_MAXM], s_sh[TAIL_MAXM];
    for (int e = threadIdx.x; e < nc * m; e += 256) s_w[e] = W[e];
    if (threadIdx.x < nc) s_b[threadIdx.x] = b ? b[threadIdx.x] : 0.f;
    if (threadIdx.x < m) {
        const int e = threadIdx.x;
        if (sums) {   // same arithmetic as k_bn_finalize_fwd_f / the folding convolutions' prologue
            const double inv_n = n_sites > 0 ? 1.0 / (double)n_sites : 0.0;
            const float gam = gamma[e], bet = beta[e];
            double v0, v1;
            urn_slab_sum2(sums + e, m, slots, v0, v1);
            const double mu = v0 * inv_n;
            double var = v1 * inv_n - mu * mu;
            if (var < 0.0) var = 0.0;
            const double is = rsqrt(var + eps);
            const float sc = gam * (float)is;
            const float sh = fmaf(-(float)mu, sc, bet);
            s_sc[e] = sc; s_sh[e] = sh;
            if (blockIdx.x == 0) {
                mean[e] = (float)mu; invstd[e] = (float)is; scale[e] = sc; shift[e] = sh;
                if (rm) rm[e] = (float)(momentum * rm[e] + (1.0 - momentum) * mu);
                if (rv) rv[e] = (float)(momentum * rv[e] + (1.0 - momentum) * var);
            }
        } else {
            s_sc[e] = scale[e]; s_sh[e] = shift[e];
        }
    }
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *xi = x + (long)row2site[i] * m;
    float acc[HEAD_MAXC];
#pragma unroll
    for (int c = 0; c < HEAD_MAXC; ++c) acc[c] = 0.f;
    for (int k = 0; k < m; k += 4) {
        f32x4 v = *(const f32x4 *)(xi + k);
        const f32x4 sc = *(const f32x4 *)(s_sc + k), sh = *(const f32x4 *)(s_sh + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], sc[j], sh[j]), 0.f);
#pragma unroll
        for (int c = 0; c < HEAD_MAXC; ++c)
            if (c < nc) {
                const float *w = s_w + c * m + k;
                acc[c] = fmaf(v[0], w[0], fmaf(v[1], w[1], fmaf(v[2], w[2], fmaf(v[3], w[3], acc[c]))));
            }
    }
#pragma unroll
    for (int c = 0; c < HEAD_MAXC; ++c)
        if (c < nc) logits[i * nc + c] = acc[c] + s_b[c];
}

// Backward of the same: per input row  y = relu(x * scale + shift) (recomputed),  g = (dl @ W) masked by y > 0, added onto
// the row's site;  dW += dl^T y, db += sum dl;  the BatchNorm-backward column sums  sum g, sum g * xhat  over the ROWS (= over
// the sites, by linearity) into the accumulated slab `part` ([slots][2][m] doubles) that urn_bn_bwd_apply_sums then reads.
// A thread owns FOUR channels of a row (m / 4 threads per row, 16-byte accesses) and keeps its share of dW, db and of the column
// sums in registers over all the rows its workgroup walks; the rows of a wave are combined with shuffles, the waves through
// LDS, the workgroup adds 2 m doubles and nc (m + 1) floats with atomics.  (The first version -- a thread per row, serial
// sums over LDS columns, one float atomic per (row, channel) onto the site -- took 35 us for 50k rows x 16 channels.)
// one2one: every site has exactly one row (n == number of sites): g is STORED, and gsite need not be zeroed.
#define TAILB_THREADS 256
__global__ __launch_bounds__(TAILB_THREADS) void k_tail_bwd(const float *__restrict__ dl, const float *__restrict__ x,
                                                            const int *__restrict__ row2site, long n, int m, int nc,
                                                            const float *__restrict__ W, const float *__restrict__ scale,
                                                            const float *__restrict__ shift, const float *__restrict__ mean,
                                                            const float *__restrict__ invstd, float *__restrict__ gsite,
                                                            float *__restrict__ dW, float *__restrict__ db, double *__restrict__ part,
                                                            int slots, int one2one)
{
    __shared__ float s_dw[TAILB_THREADS / 64][8 * 32];
    __shared__ float s_db[TAILB_THREADS / 64][8];
    __shared__ double s_pt[TAILB_THREADS / 64][2 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tpr = m >> 2;                       // threads per row: 4 (m = 16) or 8 (m = 32); a power of two (host)
    const int k = 4 * (tid & (tpr - 1));          // first channel of this thread
    const int rpp = TAILB_THREADS / tpr;          // rows per pass of the workgroup
    const f32x4 sc = *(const f32x4 *)(scale + k), sh = *(const f32x4 *)(shift + k);
    const f32x4 mu = *(const f32x4 *)(mean + k), is = *(const f32x4 *)(invstd + k);
    f32x4 w[8], dwa[8];
    float dba[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        w[c] = c < nc ? *(const f32x4 *)(W + c * m + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
        dwa[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dba[c] = 0.f;
    }
    double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
    for (long row = (long)blockIdx.x * rpp + tid / tpr; row < n; row += (long)gridDim.x * rpp) {
        float d[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) d[c] = c < nc ? dl[row * nc + c] : 0.f;
        const long site = (long)row2site[row];
        const f32x4 xv = *(const f32x4 *)(x + site * m + k);
        f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float y = fmaxf(fmaf(xv[j], sc[j], sh[j]), 0.f);
            float gj = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) gj = fmaf(d[c], w[c][j], gj);
            if (!(y > 0.f)) gj = 0.f;
            const float xh = (xv[j] - mu[j]) * is[j];
            s0[j] += (double)gj;
            s1[j] += (double)(gj * xh);
#pragma unroll
            for (int c = 0; c < 8; ++c) dwa[c][j] = fmaf(d[c], y, dwa[c][j]);
            g[j] = gj;
        }
        if (one2one) {
            *(f32x4 *)(gsite + site * m + k) = g;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (g[j] != 0.f) atomicAdd(&gsite[site * m + k + j], g[j]);
        }
        if (k == 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c) dba[c] += d[c];
        }
    }
    // rows of the wave: lanes with the same channels are tpr apart
    for (int off = tpr; off < 64; off <<= 1) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c >= nc) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) dwa[c][j] += __shfl_xor(dwa[c][j], off);
            dba[c] += __shfl_xor(dba[c], off);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { s0[j] += __shfl_xor(s0[j], off); s1[j] += __shfl_xor(s1[j], off); }
    }
    if (lane < tpr) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < nc) {
#pragma unroll
                for (int j = 0; j < 4; ++j) s_dw[wave][c * m + k + j] = dwa[c][j];
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) { s_pt[wave][k + j] = s0[j]; s_pt[wave][m + k + j] = s1[j]; }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c) s_db[wave][c] = dba[c];
        }
    }
    __syncthreads();
    constexpr int NW = TAILB_THREADS / 64;
    for (int e = tid; e < nc * m; e += TAILB_THREADS) {
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < NW; ++v) acc += s_dw[v][e];
        atomicAdd(&dW[e], acc);
    }
    if (tid < nc) {
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < NW; ++v) acc += s_db[v][tid];
        atomicAdd(&db[tid], acc);
    }
    if (tid < 2 * m) {
        double acc = 0.0;
#pragma unroll
        for (int v = 0; v < NW; ++v) acc += s_pt[v][tid];
        const long slot = blockIdx.x % (unsigned)slots;
        unsafeAtomicAdd(&part[slot * 2 * m + tid], acc);
    }
}

extern "C" int urn_tail_fwd(const float *x, const int32_t *row2site, int64_t n, int m, int nc, const float *W, const float *b,
                            const double *sums, int slots, int64_t n_sites, double eps, const float *gamma, const float *beta,
                            float *mean, float *invstd, float *scale, float *shift, float *running_mean, float *running_var,
                            double momentum, float *logits, void *stream)
{
    if (n <= 0) return URN_OK;
    URN_CHECK_ARG(x && row2site && W && logits && scale && shift, "null pointer");
    URN_CHECK_ARG(m > 0 && m % 4 == 0 && m <= TAIL_MAXM && nc > 0 && nc <= HEAD_MAXC && nc * m <= HEAD_MAXC * HEAD_MAXM / 8, "unsupported head shape");
    URN_CHECK_ARG(!sums || (slots > 0 && n_sites > 0 && gamma && beta && mean && invstd), "incomplete statistics");
    hipLaunchKernelGGL(k_tail_fwd, dim3(urn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, row2site, (long)n, m, nc, W, b, sums, slots,
                       (long)n_sites, eps, gamma, beta, mean, invstd, scale, shift, running_mean, running_var, momentum, logits);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

extern "C" int urn_tail_bwd(const float *dlogits, const float *x, const int32_t *row2site, int64_t n, int m, int nc, const float *W,
                            const float *scale, const float *shift, const float *mean, const float *invstd, float *gsite,
                            int64_t n_sites, float *dW, float *db, double *part, int slots, void *stream)
{
    if (n <= 0) return URN_OK;
    URN_CHECK_ARG(dlogits && x && row2site && W && scale && shift && mean && invstd && gsite && dW && db && part && slots > 0, "null pointer");
    URN_CHECK_ARG((m == 16 || m == 32) && nc > 0 && nc <= 8, "unsupported head shape (m 16 or 32, nc <= 8)");
    const int rpp = TAILB_THREADS / (m / 4);
    const long passes = urn_cdiv(n, rpp);
    // (every workgroup ends with nc * m float atomics onto the SAME addresses: 782 workgroups were 20 of the kernel's 26 us)
    const long per_wg = urn_cdiv(passes, 192);      // at most 192 workgroups, each walking the same number of passes
    hipLaunchKernelGGL(k_tail_bwd, dim3((unsigned)urn_cdiv(passes, per_wg)), dim3(TAILB_THREADS), 0, (hipStream_t)stream, dlogits, x, row2site,
                       (long)n, m, nc, W, scale, shift, mean, invstd, gsite, dW, db, part, slots, n_sites == n ? 1 : 0);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

// ---- per-event mean cross-entropy ------------------------------------------------------------
// pass 1: per row log-softmax CE (optionally weighted) and argmax hit; per-event sums with atomics:
//         ev[e] = (sum w*ce, row count, hits)   (double, float-exact counts)
__global__ __launch_bounds__(256) void k_ce_fwd(const float *__restrict__ logits, const float *__restrict__ label,
                                                const float *__restrict__ bid, int bid_stride,
                                                const float *__restrict__ weight, long n, int nc,
                                                float *__restrict__ row_lse, double *__restrict__ ev)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const bool ok = i < n;
    double ce = 0.0, one = 0.0, hit = 0.0;
    int e = -1;
    if (ok) {
        const float *l = logits + i * nc;
        float mx = l[0];
        int am = 0;
        for (int c = 1; c < nc; ++c)
            if (l[c] > mx) { mx = l[c]; am = c; }
        float s = 0.f;
        for (int c = 0; c < nc; ++c) s += expf(l[c] - mx);
        const float lse = mx + logf(s);
        row_lse[i] = lse;
        const int lab = (int)label[i];
        e = (int)bid[i * bid_stride];
        const float w = weight ? weight[i] : 1.f;
        ce = (double)((lse - l[lab]) * w);
        one = 1.0;
        hit = am == lab ? 1.0 : 0.0;
    }
    // rows of one event are contiguous in practice: when the whole wave belongs to one event, reduce in
    // the wave and issue 3 atomics instead of 192 (all rows would otherwise hit the same three addresses)
    const int e0 = __shfl(e, 0);
    const bool uniform = __all(!ok || e == e0) && e0 >= 0;
    __shared__ int s_e[4];
    __shared__ double s_v[4][3];
    const int wave = threadIdx.x >> 6;
    if (uniform) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { ce += __shfl_xor(ce, d); one += __shfl_xor(one, d); hit += __shfl_xor(hit, d); }
    }
    if ((threadIdx.x & 63) == 0) {
        s_e[wave] = uniform ? e0 : -1;
        s_v[wave][0] = ce; s_v[wave][1] = one; s_v[wave][2] = hit;
    }
    __syncthreads();
    const bool block_uniform = s_e[0] >= 0 && s_e[1] == s_e[0] && s_e[2] == s_e[0] && s_e[3] == s_e[0];
    if (block_uniform) {   // one event in the whole block: 3 atomics per block
        if (threadIdx.x < 3)
            atomicAdd(&ev[3 * s_e[0] + threadIdx.x], ((s_v[0][threadIdx.x] + s_v[1][threadIdx.x]) + s_v[2][threadIdx.x]) + s_v[3][threadIdx.x]);
    } else if (uniform) {
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&ev[3 * e0 + 0], ce);
            atomicAdd(&ev[3 * e0 + 1], one);
            atomicAdd(&ev[3 * e0 + 2], hit);
        }
    } else if (ok) {
        atomicAdd(&ev[3 * e + 0], ce);
        atomicAdd(&ev[3 * e + 1], one);
        atomicAdd(&ev[3 * e + 2], hit);
    }
}

// pass 2: loss = sum_e mean_e, acc = sum_e hits_e/count_e  (out[0], out[1]); one small block
__global__ void k_ce_reduce(const double *__restrict__ ev, int nev, float *__restrict__ out)
{
    __shared__ double s0[256], s1[256];
    double a = 0.0, b = 0.0;
    for (int e = threadIdx.x; e < nev; e += 256) {
        const double cnt = ev[3 * e + 1];
        if (cnt > 0.0) { a += ev[3 * e + 0] / cnt; b += ev[3 * e + 2] / cnt; }
    }
    s0[threadIdx.x] = a; s1[threadIdx.x] = b;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) { s0[threadIdx.x] += s0[threadIdx.x + d]; s1[threadIdx.x] += s1[threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = (float)s0[0]; out[1] = (float)s1[0]; }
}

// backward: dlogits[i,c] = gout * w_i / count_e * (softmax_ic - [c == label_i])
__global__ __launch_bounds__(256) void k_ce_bwd(const float *__restrict__ logits, const float *__restrict__ label,
                                                const float *__restrict__ bid, int bid_stride,
                                                const float *__restrict__ weight, const float *__restrict__ row_lse,
                                                const double *__restrict__ ev, const float *__restrict__ gout, long n,
                                                int nc, float *__restrict__ dlogits)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int lab = (int)label[i];
    const int e = (int)bid[i * bid_stride];
    const float w = weight ? weight[i] : 1.f;
    const float scale = gout[0] * w / (float)ev[3 * e + 1];
    const float lse = row_lse[i];
    for (int c = 0; c < nc; ++c) {
        const float p = expf(logits[i * nc + c] - lse);
        dlogits[i * nc + c] = scale * (p - (c == lab ? 1.f : 0.f));
    }
}

extern "C" int64_t urn_ce_scratch_bytes(void) { return (int64_t)CE_MAXEV * 3 * 8; }

extern "C" int urn_ce_fwd(const float *logits, const float *label, const float *batch_id, int batch_id_stride,
                          const float *weight, int64_t n, int nc, float *row_lse, double *ev, float *out, void *stream)
{
    URN_CHECK_ARG(ev && out && n >= 0 && nc > 0 && nc <= HEAD_MAXC, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(ev, 0, (size_t)urn_ce_scratch_bytes(), st) != hipSuccess) { urn_set_error("urn_ce_fwd: memset failed"); return URN_EHIP; }
    if (n > 0) {
        URN_CHECK_ARG(logits && label && batch_id && row_lse && batch_id_stride > 0, "null pointer");
        hipLaunchKernelGGL(k_ce_fwd, dim3(urn_cdiv(n, 256)), dim3(256), 0, st, logits, label, batch_id, batch_id_stride, weight,
                           (long)n, nc, row_lse, ev);
    }
    hipLaunchKernelGGL(k_ce_reduce, dim3(1), dim3(256), 0, st, ev, CE_MAXEV, out);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

extern "C" int urn_ce_bwd(const float *logits, const float *label, const float *batch_id, int batch_id_stride,
                          const float *weight, const float *row_lse, const double *ev, const float *grad_out, int64_t n,
                          int nc, float *dlogits, void *stream)
{
    if (n <= 0) return URN_OK;
    URN_CHECK_ARG(logits && label && batch_id && row_lse && ev && grad_out && dlogits, "null pointer");
    hipLaunchKernelGGL(k_ce_bwd, dim3(urn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, logits, label, batch_id,
                       batch_id_stride, weight, row_lse, ev, grad_out, (long)n, nc, dlogits);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

// ------------------------------------------------------------------------------------ Adam on flat buffers --
// torch.optim.Adam (reference uresnet/trainval.py:37: Adam(self._net.parameters(), lr)) over ONE contiguous
// segment of parameters: the executor keeps parameters, gradients and both moments flat, so a step is one
// 16-byte-per-lane streaming pass instead of a multi-tensor launch over ~190 small tensors.
__global__ __launch_bounds__(256) void k_adam_flat(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, long n, float b1, float b2, float omb1,
                                                   float omb2, float eps, float wd, float step_size,
                                                   float inv_bc2_sqrt)
{
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n && ((((uintptr_t)(p + i)) | ((uintptr_t)(g + i)) | ((uintptr_t)(m + i)) | ((uintptr_t)(v + i))) & 15) == 0) {
        f32x4 pv = *(f32x4 *)(p + i), mv = *(f32x4 *)(m + i), vv = *(f32x4 *)(v + i);
        const f32x4 gv = *(const f32x4 *)(g + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gr = fmaf(wd, pv[k], gv[k]);
            mv[k] = fmaf(b1, mv[k], omb1 * gr);          // lerp(m, g, 1 - b1)
            vv[k] = fmaf(b2, vv[k], omb2 * gr * gr);
            pv[k] -= step_size * mv[k] / (sqrtf(vv[k]) * inv_bc2_sqrt + eps);
        }
        *(f32x4 *)(p + i) = pv; *(f32x4 *)(m + i) = mv; *(f32x4 *)(v + i) = vv;
    } else {
        for (long j = i; j < min(n, i + 4); ++j) {
            const float gr = fmaf(wd, p[j], g[j]);
            const float mm = fmaf(b1, m[j], omb1 * gr);
            const float vv = fmaf(b2, v[j], omb2 * gr * gr);
            m[j] = mm; v[j] = vv;
            p[j] -= step_size * mm / (sqrtf(vv) * inv_bc2_sqrt + eps);
        }
    }
}

extern "C" int urn_adam_flat(float *p, const float *g, float *m, float *v, int64_t n, double lr, double beta1, double beta2,
                             double eps, double weight_decay, int64_t step, void *stream)
{
    if (n <= 0) return URN_OK;
    URN_CHECK_ARG(p && g && m && v && step >= 1 && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1, "bad argument");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(k_adam_flat, dim3(urn_cdiv((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n,
                       (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)));
    URN_LAUNCH_CHECK();
    return URN_OK;
}
