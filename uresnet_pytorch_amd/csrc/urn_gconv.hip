// Gather convolution on gfx950: the arithmetic behind scn.SubmanifoldConvolution,
// scn.Convolution(k2,s2) and scn.Deconvolution(k2,s2) forward, input-gradient and
// weight-gradient (reference call sites uresnet/models/uresnet_sparse.py:21-22).
//
//   y[j,:] = sum_{o<K} x[tbl[t(o)*ld + j], :] @ W[o]
//
// Forward / input-gradient kernel (output stationary, no atomics, deterministic):
//   one wave owns MB*16 output rows x NB*16 output columns; for each filter offset
//   it ballots whether any of its rows has an active neighbour and skips the offset
//   otherwise; A fragments are gathered straight from HBM/L2 in MFMA operand layout
//   (one 16-byte load per lane = 4 k-values of one gathered row), B fragments come
//   from the pre-transposed weights (K, cout, cin) which stay L2-resident;
//   v_mfma_f32_16x16x4_f32 accumulates in fp32 (exact f32 fma chain).
// Weight-gradient kernel: rows are the reduction dimension, so the valid (in,out)
//   pairs of an offset are compacted with wave64 ballot/popcount prefix sums, staged
//   through LDS in batches of 32 pairs and contracted with the same MFMA.
#include "urn_common.h"
#include "urn_prof.h"

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------- weight gradient --
// grid (row chunks, K, output tiles); block 256 = 4 waves.
// Output tile: up to 4 ci-blocks x 5 co-blocks of 16x16, block b -> wave b%4, slot b/4.
#define DW_KT 32      // pairs per MFMA batch
#define DW_MAXI 4     // ci blocks per tile  (64 channels)
#define DW_MAXN 5     // co blocks per tile  (80 channels)
#define DW_SLOTS 5    // ceil(4*5/4)
#define DW_LIST 1024

__global__ __launch_bounds__(256) void k_gconv_dw(const float *__restrict__ x, const float *__restrict__ xf_scale,
                                                  const float *__restrict__ xf_shift, const float *__restrict__ dy,
                                                  const int *__restrict__ tbl, long ld, long n_out, int cin,
                                                  int cout, long chunk, int n_ci_tiles,
                                                  float *__restrict__ dw)
{
    __shared__ int s_in[DW_LIST], s_out[DW_LIST];
    __shared__ int s_wcnt[4];
    __shared__ __attribute__((aligned(16))) float s_a[2][DW_KT][DW_MAXI * 16 + 16];
    __shared__ __attribute__((aligned(16))) float s_b[2][DW_KT][DW_MAXN * 16 + 32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int o = blockIdx.y;
    const int ci_tile = blockIdx.z % n_ci_tiles, co_tile = blockIdx.z / n_ci_tiles;
    const int ci0 = ci_tile * DW_MAXI * 16, co0 = co_tile * DW_MAXN * 16;
    const int ci_w = min(DW_MAXI * 16, cin - ci0), co_w = min(DW_MAXN * 16, cout - co0);
    const int mi_n = ci_w / 16, ni_n = co_w / 16, nblk = mi_n * ni_n;

    f32x4 acc[DW_SLOTS];
#pragma unroll
    for (int s = 0; s < DW_SLOTS; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const long row_begin = (long)blockIdx.x * chunk;
    const long row_end = min(n_out, row_begin + chunk);
    int cnt = 0;  // pairs currently in the list (block-uniform)

    for (long row0 = row_begin; row0 < row_end || cnt > 0; row0 += 256) {
        const bool last = row0 >= row_end;
        if (!last) {
            // compact the valid (in,out) pairs of 256 candidate rows, order preserved
            long row = row0 + tid;
            int idx = (row < row_end) ? tbl[(long)o * ld + row] : -1;
            unsigned long long bal = __ballot(idx >= 0);
            if (lane == 0) s_wcnt[wave] = __popcll(bal);
            __syncthreads();
            int base = cnt, tot = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                int v = s_wcnt[w];
                if (w < wave) base += v;
                tot += v;
            }
            if (idx >= 0) {
                int p = base + __popcll(bal & ((1ull << lane) - 1ull));
                s_in[p] = idx;
                s_out[p] = (int)row;
            }
            cnt += tot;
            __syncthreads();
        }
        // consume full batches (and the final partial one).  Software pipeline: the rows of batch b+1 are in flight
        // (registers) while the MFMAs of batch b run out of LDS; loads are unconditional (clamped indices) because a
        // branch around a load makes hipcc wait for every load separately; the barrier is LDS-only (no vmcnt drain).
        int done = 0;
        const int nbatch = last ? (cnt + DW_KT - 1) / DW_KT : cnt / DW_KT;
        const int a_tot = DW_KT * (ci_w / 4), b_tot = DW_KT * (co_w / 4);
        f32x4 ra[2], rb[3];
        auto prefetch = [&](int b) {
            const int base = b * DW_KT, nb = min(DW_KT, cnt - base);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int e = min(j * 256 + tid, a_tot - 1);
                const int rr = e / (ci_w / 4), c4 = e - rr * (ci_w / 4);
                ra[j] = *(const f32x4 *)(x + (long)s_in[base + min(rr, nb - 1)] * cin + ci0 + 4 * c4);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int e = min(j * 256 + tid, b_tot - 1);
                const int rr = e / (co_w / 4), c4 = e - rr * (co_w / 4);
                rb[j] = *(const f32x4 *)(dy + (long)s_out[base + min(rr, nb - 1)] * cout + co0 + 4 * c4);
            }
        };
        auto park = [&](int b, int buf) {
            const int nb = min(DW_KT, cnt - b * DW_KT);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int e = j * 256 + tid;
                if (e < a_tot) {
                    const int rr = e / (ci_w / 4), c4 = e - rr * (ci_w / 4);
                    f32x4 v = ra[j];
                    if (xf_scale) {  // the conv's input was relu(x*scale + shift): recompute it on the fly
                        const f32x4 sc = *(const f32x4 *)(xf_scale + ci0 + 4 * c4);
                        const f32x4 sh = *(const f32x4 *)(xf_shift + ci0 + 4 * c4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaf(v[k], sc[k], sh[k]), 0.f);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = rr < nb ? v[k] : 0.f;   // zero-fill the tail of a partial batch
                    *(f32x4 *)&s_a[buf][rr][4 * c4] = v;
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int e = j * 256 + tid;
                if (e < b_tot) {
                    const int rr = e / (co_w / 4), c4 = e - rr * (co_w / 4);
                    f32x4 v = rb[j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = rr < nb ? v[k] : 0.f;
                    *(f32x4 *)&s_b[buf][rr][4 * c4] = v;
                }
            }
        };
        if (nbatch > 0) prefetch(0);
        for (int b = 0; b < nbatch; ++b) {
            const int buf = b & 1;
            park(b, buf);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (b + 1 < nbatch) prefetch(b + 1);
#pragma unroll
            for (int s = 0; s < DW_SLOTS; ++s) {
                int blk = wave + 4 * s;
                if (blk < nblk) {  // wave-uniform
                    int mi = blk / ni_n, ni = blk - mi * ni_n;
#pragma unroll
                    for (int ks = 0; ks < DW_KT / 4; ++ks) {
                        float av = s_a[buf][4 * ks + q][mi * 16 + m];
                        float bv = s_b[buf][4 * ks + q][ni * 16 + m];
                        acc[s] = MFMA16(av, bv, acc[s]);
                    }
                }
            }
            done += min(DW_KT, cnt - b * DW_KT);
        }
        __syncthreads();
        // move the remainder (< DW_KT pairs) to the front of the list
        int rem = cnt - done;
        int vi = 0, vo = 0;
        if (tid < rem) { vi = s_in[done + tid]; vo = s_out[done + tid]; }
        __syncthreads();
        if (tid < rem) { s_in[tid] = vi; s_out[tid] = vo; }
        cnt = rem;
        __syncthreads();
        if (last) break;
    }
    // accumulate this block's partial into dw[o][ci][co]; C layout: col = lane&15, row = q*4+i
#pragma unroll
    for (int s = 0; s < DW_SLOTS; ++s) {
        int b = wave + 4 * s;
        if (b < nblk) {
            int mi = b / ni_n, ni = b - mi * ni_n;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int ci = ci0 + mi * 16 + q * 4 + i, co = co0 + ni * 16 + m;
                atomicAdd(&dw[((long)o * cin + ci) * cout + co], acc[s][i]);
            }
        }
    }
}

// VALU weight gradient for channel counts that are not multiples of 16 (the 1-channel stem).
// grid (chunks, K); 256 threads = (cin*cout) pairs x R row lanes; LDS reduction over the row lanes,
// one atomic per (pair, block).
__global__ __launch_bounds__(256) void k_gconv_dw_small(const float *__restrict__ x, const float *__restrict__ dy,
                                                        const int *__restrict__ tbl, long ld, long n_out, int cin, int cout,
                                                        long chunk, float *__restrict__ dw)
{
    __shared__ float s_acc[256];
    const int o = blockIdx.y, t = threadIdx.x;
    const int pairs = cin * cout;
    const long row_begin = (long)blockIdx.x * chunk, row_end = min(n_out, row_begin + chunk);
    if (pairs <= 256) {
        const int R = 256 / pairs, pair = t % pairs, rl = t / pairs;
        const int ci = pair / cout, co = pair - ci * cout;
        float acc = 0.f;
        if (rl < R)
            for (long j = row_begin + rl; j < row_end; j += R) {
                const int i = tbl[(long)o * ld + j];
                if (i >= 0) acc = fmaf(x[(long)i * cin + ci], dy[j * cout + co], acc);
            }
        s_acc[t] = acc;
        __syncthreads();
        if (t < pairs) {
            float v = 0.f;
            for (int k = 0; k < R; ++k) v += s_acc[k * pairs + t];
            atomicAdd(&dw[((long)o * cin + ci) * cout + co], v);
        }
    } else {
        for (int e = t; e < pairs; e += 256) {
            const int ci = e / cout, co = e - ci * cout;
            float acc = 0.f;
            for (long j = row_begin; j < row_end; ++j) {
                const int i = tbl[(long)o * ld + j];
                if (i >= 0) acc = fmaf(x[(long)i * cin + ci], dy[j * cout + co], acc);
            }
            atomicAdd(&dw[((long)o * cin + ci) * cout + co], acc);
        }
    }
}

int g_dw_blocks = 2048;   // target number of workgroups of the weight-gradient kernel (urn_set_option "dw_blocks")

extern "C" int urn_gconv_bwd_dw(const float *x, const float *dy, const int32_t *tbl, int64_t ld, int K,
                                int64_t n_out, int cin, int cout, float *dw, void *stream)
{
    return urn_gconv_bwd_dw_ex(x, nullptr, nullptr, dy, tbl, ld, K, n_out, cin, cout, dw, stream);
}

extern "C" int urn_gconv_bwd_dw_ex(const float *x, const float *xf_scale, const float *xf_shift, const float *dy,
                                   const int32_t *tbl, int64_t ld, int K, int64_t n_out, int cin, int cout,
                                   float *dw, void *stream)
{
    if (n_out <= 0) return URN_OK;
    URN_CHECK_ARG(x && dy && tbl && dw, "null pointer");
    URN_CHECK_ARG(K > 0 && cin > 0 && cout > 0 && ld >= n_out, "bad shape");
    URN_CHECK_ARG((xf_scale == nullptr) == (xf_shift == nullptr), "scale and shift go together");
    hipStream_t st = (hipStream_t)stream;
    if ((cin % 16) || (cout % 16)) {
        if (xf_scale) { urn_set_error("urn_gconv_bwd_dw_ex: input transform needs channel counts that are multiples of 16"); return URN_EUNSUPPORTED; }
        int chunks = (int)((n_out + 255) / 256);
        if (chunks > 128) chunks = 128;
        long chunk = (n_out + chunks - 1) / chunks;
        hipLaunchKernelGGL(k_gconv_dw_small, dim3(chunks, K), dim3(256), 0, st, x, dy, tbl, (long)ld, (long)n_out,
                           cin, cout, chunk, dw);
        URN_LAUNCH_CHECK();
        return URN_OK;
    }
    const int n_ci_tiles = urn_cdiv(cin, DW_MAXI * 16), n_co_tiles = urn_cdiv(cout, DW_MAXN * 16);
    // aim for ~2048 blocks, chunks of at least 256 rows (multiple of 256)
    int chunks = g_dw_blocks / (K * n_ci_tiles * n_co_tiles);
    if (chunks < 1) chunks = 1;
    long chunk = (n_out + chunks - 1) / chunks;
    if (chunk < 256) chunk = 256;
    chunk = ((chunk + 255) / 256) * 256;
    chunks = (int)((n_out + chunk - 1) / chunk);
    const bool prof = urn_prof_on();
    if (prof) urn_prof_begin(URN_PROF_DW, st);
    hipLaunchKernelGGL(k_gconv_dw, dim3(chunks, K, n_ci_tiles * n_co_tiles), dim3(256), 0, st, x, xf_scale, xf_shift, dy, tbl, (long)ld,
                       (long)n_out, cin, cout, chunk, n_ci_tiles, dw);
    if (prof) urn_prof_end(st);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

// ----------------------------------------------------------------- weight transpose --
__global__ void k_transpose_w(const float *__restrict__ w, int a, int b, float *__restrict__ wt)
{
    // grid.y = K; wt[o][j][i] = w[o][i][j]
    const long base = (long)blockIdx.y * a * b;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < a * b; e += gridDim.x * blockDim.x) {
        int j = e / a, i = e - j * a;
        wt[base + e] = w[base + (long)i * b + j];
    }
}

extern "C" int urn_transpose_w(const float *w, int K, int a, int b, float *wt, void *stream)
{
    URN_CHECK_ARG(w && wt && K > 0 && a > 0 && b > 0, "bad argument");
    int gx = urn_cdiv((int64_t)a * b, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_transpose_w, dim3(gx, K), dim3(256), 0, (hipStream_t)stream, w, a, b, wt);
    URN_LAUNCH_CHECK();
    return URN_OK;
}
