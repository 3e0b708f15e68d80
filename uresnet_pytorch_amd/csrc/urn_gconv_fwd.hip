// Gather convolution forward / input-gradient on gfx950: the arithmetic behind
// scn.SubmanifoldConvolution, scn.Convolution(k2,s2), scn.Deconvolution(k2,s2) and
// scn.NetworkInNetwork (reference call sites uresnet/models/uresnet_sparse.py:21-22).
//
//   y[j,:] = sum_{o<K} x[tbl[t(o)*ld + j], :] @ W[o]   (+ res[j,:])
//
// Output stationary, no atomics, bit-reproducible.  A workgroup owns one tile of 16
// output rows x NB*16 output columns (SPLIT=4: its four waves share the tile and split
// the ACTIVE filter offsets between them, partial tiles are summed through LDS in wave
// order) or four such tiles (SPLIT=1: one per wave).  Per tile:
//   1. all K table rows of the tile are fetched at once (7 coalesced loads per lane),
//      wave64 ballots turn them into a bit mask of offsets that have any active
//      neighbour; inactive offsets cost nothing;
//   2. for each active offset the wave gathers its A operand straight from HBM/L2 in
//      MFMA layout (16-byte loads: 4 k-values of one gathered row per lane) and the B
//      operand from the pre-transposed, L2-resident weights (K, cout, cin); the k loop is
//      fully unrolled (KS = cin/16 is a template parameter) so every load of an offset
//      is in flight before the first v_mfma_f32_16x16x4_f32 issues.
#include "urn_common.h"
#include "urn_prof.h"

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int KS, int NB, int SPLIT>
__global__ __launch_bounds__(256) void k_gconv_fwd(const float *__restrict__ x, const float *__restrict__ wt,
                                                   const int *__restrict__ tbl, long ld, int K, int flip,
                                                   const int *n_dev, long n_cap, int cout,
                                                   const float *__restrict__ res, float *__restrict__ y)
{
    constexpr int CIN = KS * 16;
    __shared__ int s_idx[4][28 * 16];
    __shared__ f32x4 s_red[SPLIT == 4 ? 4 * NB * 64 : 1];

    const long n_out = n_dev ? (long)*n_dev : n_cap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const long tile = (SPLIT == 4) ? (long)blockIdx.x : (long)blockIdx.x * 4 + wave;
    const long row_base = tile * 16;
    if (SPLIT == 1 && row_base >= n_out) return;  // wave-uniform, no barriers in this variant
    const bool tile_ok = row_base < n_out;         // block-uniform when SPLIT == 4
    const int col_base = blockIdx.y * (NB * 16);

    // 1. the tile's table: lane (r, q) fetches table rows q, q+4, ... for output row r
    unsigned amask = 0u;
    {
        const long row = row_base + r;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int t = 4 * i + q;
            int v = -1;
            if (t < K && row < n_out) v = tbl[(long)t * ld + row];
            s_idx[wave][i * 64 + lane] = v;  // == [t][r]
            const unsigned long long b = __ballot(v >= 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((b >> (16 * j)) & 0xFFFFull) amask |= 1u << (4 * i + j);
        }
    }

    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // 2. active offsets (table rows), split round-robin between the waves when SPLIT == 4
    int cnt = 0;
    unsigned m = amask;
    while (m) {
        const int t = __builtin_ctz(m);
        m &= m - 1u;
        if (SPLIT == 4 && ((cnt++ & 3) != wave)) continue;
        const int o = flip ? (K - 1 - t) : t;  // weight index
        const int idx = s_idx[wave][t * 16 + r];
        const float *xa = x + (long)(idx < 0 ? 0 : idx) * CIN + 4 * q;
        const float *wo = wt + ((long)o * cout + col_base + r) * CIN + 4 * q;
        f32x4 a[KS], b[KS][NB];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            a[ks] = *(const f32x4 *)(xa + ks * 16);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b[ks][nb] = *(const f32x4 *)(wo + (long)nb * 16 * CIN + ks * 16);
        }
        if (idx < 0) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a[ks] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA16(a[ks][tt], b[ks][nb][tt], acc[nb]);
    }

    // 3. (SPLIT == 4) sum the four partial tiles in wave order
    if (SPLIT == 4) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) s_red[(wave * NB + nb) * 64 + lane] = acc[nb];
        __syncthreads();
        if (wave != 0 || !tile_ok) return;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 s = s_red[(0 * NB + nb) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                f32x4 p = s_red[(w * NB + nb) * 64 + lane];
                s[0] += p[0]; s[1] += p[1]; s[2] += p[2]; s[3] += p[3];
            }
            acc[nb] = s;
        }
    }
    // C layout of 16x16x4: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long row = row_base + q * 4 + i;
        if (row >= n_out) continue;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const long off = row * cout + col_base + nb * 16 + r;
            float v = acc[nb][i];
            if (res) v += res[off];
            y[off] = v;
        }
    }
}

// VALU fallback for widths that are not multiples of 16 (the 1-channel stem).
__global__ void k_gconv_small(const float *__restrict__ x, const float *__restrict__ wt,
                              const int *__restrict__ tbl, long ld, int K, int flip, long n_out, int cin,
                              int cout, const float *__restrict__ res, float *__restrict__ y)
{
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_out * cout) return;
    long j = t / cout;
    int c = (int)(t - j * cout);
    float acc = 0.f;
    for (int o = 0; o < K; ++o) {
        int to = flip ? (K - 1 - o) : o;
        int i = tbl[(long)to * ld + j];
        if (i < 0) continue;
        const float *xi = x + (long)i * cin;
        const float *w = wt + ((long)o * cout + c) * cin;
        for (int a = 0; a < cin; ++a) acc = fmaf(xi[a], w[a], acc);
    }
    if (res) acc += res[t];
    y[t] = acc;
}

struct FwdArgs {
    const float *x, *wt;
    const int *tbl;
    long ld;
    int K, flip;
    long n_out;
    int cout;
    const float *res;
    float *y;
    hipStream_t st;
};

template <int KS, int NB>
static void launch_ks_nb(const FwdArgs &a, bool split)
{
    const long tiles = (a.n_out + 15) / 16;
    const int gy = a.cout / (NB * 16);
    if (split)
        hipLaunchKernelGGL((k_gconv_fwd<KS, NB, 4>), dim3((unsigned)tiles, gy), dim3(256), 0, a.st, a.x, a.wt, a.tbl,
                           a.ld, a.K, a.flip, (const int *)nullptr, a.n_out, a.cout, a.res, a.y);
    else
        hipLaunchKernelGGL((k_gconv_fwd<KS, NB, 1>), dim3((unsigned)((tiles + 3) / 4), gy), dim3(256), 0, a.st, a.x,
                           a.wt, a.tbl, a.ld, a.K, a.flip, (const int *)nullptr, a.n_out, a.cout, a.res, a.y);
}

template <int KS>
static void launch_ks(const FwdArgs &a, int nb, bool split)
{
    // operand registers: 4*KS*(1+NB); keep them under the 256-register budget
    if (nb == 1) { launch_ks_nb<KS, 1>(a, split); return; }
    if (nb == 2) { launch_ks_nb<KS, 2>(a, split); return; }
    if constexpr (KS <= 8) {
        if (nb == 3) { launch_ks_nb<KS, 3>(a, split); return; }
        if (nb == 4) { launch_ks_nb<KS, 4>(a, split); return; }
    }
    if constexpr (KS <= 5) {
        if (nb == 5) { launch_ks_nb<KS, 5>(a, split); return; }
    }
}

extern "C" int urn_gconv_fwd(const float *x, const float *wt, const int32_t *tbl, int64_t ld, int K, int flip,
                             int64_t n_out, int cin, int cout, const float *res, float *y, void *stream)
{
    if (n_out <= 0) return URN_OK;
    URN_CHECK_ARG(x && wt && tbl && y, "null pointer");
    URN_CHECK_ARG(K > 0 && cin > 0 && cout > 0 && ld >= n_out, "bad shape");
    URN_CHECK_ARG((const void *)x != (const void *)y, "y aliases x");
    hipStream_t st = (hipStream_t)stream;
    const int ks = cin / 16;
    const bool mfma_ok = (cin % 16 == 0) && (cout % 16 == 0) && K <= 28 &&
                         (ks <= 6 || ks == 8 || ks == 10 || ks == 12 || ks == 14);
    if (!mfma_ok) {
        hipLaunchKernelGGL(k_gconv_small, dim3(urn_cdiv(n_out * cout, 256)), dim3(256), 0, st, x, wt, tbl, (long)ld,
                           K, flip, (long)n_out, cin, cout, res, y);
        URN_LAUNCH_CHECK();
        return URN_OK;
    }
    const int nblk = cout / 16;
    // columns per wave: the largest divisor of cout/16 that keeps the operand registers under budget
    const int nb_max = ks <= 5 ? 5 : (ks <= 8 ? 4 : 2);
    const bool split = K >= 8;
    // ... and that still leaves >= ~4 waves per SIMD (1024 SIMDs) to hide the gather latency
    const long tiles16 = (n_out + 15) / 16;
    int nb = 1;
    for (int d = nb_max; d >= 1; --d)
        if (nblk % d == 0 && (d == 1 || tiles16 * (split ? 4 : 1) * (nblk / d) >= 4096)) { nb = d; break; }
    FwdArgs a{x, wt, tbl, (long)ld, K, flip, (long)n_out, cout, res, y, st};
    const bool prof = urn_prof_on();
    if (prof) urn_prof_begin(URN_PROF_GCONV, st);
    switch (ks) {
    case 1: launch_ks<1>(a, nb, split); break;
    case 2: launch_ks<2>(a, nb, split); break;
    case 3: launch_ks<3>(a, nb, split); break;
    case 4: launch_ks<4>(a, nb, split); break;
    case 5: launch_ks<5>(a, nb, split); break;
    case 6: launch_ks<6>(a, nb, split); break;
    case 8: launch_ks<8>(a, nb, split); break;
    case 10: launch_ks<10>(a, nb > 2 ? 2 : nb, split); break;
    case 12: launch_ks<12>(a, nb > 2 ? 2 : nb, split); break;
    default: launch_ks<14>(a, nb > 2 ? 2 : nb, split); break;
    }
    if (prof) urn_prof_end(st);
    URN_LAUNCH_CHECK();
    return URN_OK;
}
