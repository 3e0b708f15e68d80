// Weight gradient of the gather convolution on the compacted rule lists, two stages, no atomics:
//     dw[t][ci][co] += sum over the rules (i, j) of table row t of  x[i][ci] * dy[j][co]
//
// Stage 1 (k_dw_pairs): one wave per (share s of the tiles, table row t, 16U x 16V block of the (cin, cout) matrix).  It
// walks the tiles of its share; per tile the header of the list says where the blocks of row t start; per block of 16
// rules it loads the 16 input rows and the 16 dy rows (U / V consecutive channels per lane: whole 64..320-byte row
// segments) and runs 4 k-steps of U x V v_mfma_f32_16x16x4_f32 with the RULES as the contraction index.  The sums stay
// in registers over the whole share and are stored once into slab[s][t] -- cin*cout*K*S floats per convolution instead
// of one partial tile per (1024-row chunk, offset) added with fp32 atomics (the earlier kernel: 5.8 MB of atomic
// traffic per launch for 0.22 MB of weights).
// Stage 2 (k_dw_reduce): dw[e] += slab[0][e] + slab[1][e] + ... in that order.
// The order of every floating-point sum is fixed by the lists: the gradient is bitwise reproducible.
#include "urn_common.h"
#include "urn_gconv_int.h"
#include "urn_prof.h"

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct DwpArgs {
    const float *x, *xf_scale, *xf_shift, *dy;
    long ldx, ld_dy;
    const int *pairs;      // NULL = identity table (K == 1)
    int T, K, S;
    long n_out;
    int cin, cout;
    int n_ci;              // blocks of 16U input channels
    float *slab;           // [S][K][cin][cout]
};

template <int N>
__device__ __forceinline__ void load_n(float (&d)[N], const float *p)
{
    if constexpr (N == 4) { const f32x4 t = *(const f32x4 *)p; d[0] = t[0]; d[1] = t[1]; d[2] = t[2]; d[3] = t[3]; }
    else if constexpr (N == 2) { const float2 t = *(const float2 *)p; d[0] = t.x; d[1] = t.y; }
    else if constexpr (N == 1) { d[0] = *p; }
    else if constexpr (N == 3) { d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; }
    else {   // N == 5: 20-byte steps, only 4-byte aligned
        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
        const f32x4u t = *(const f32x4u *)p; d[0] = t[0]; d[1] = t[1]; d[2] = t[2]; d[3] = t[3]; d[4] = p[4];
    }
}

// U, V: 16-channel groups of the input / output side per wave (lane r owns channels U*r .. U*r+U-1 of its block: the row
// of MFMA u is channel U*i + u, so that a lane's U values are contiguous in memory); XF: rows are relu(x*scale+shift)
// PREC: MFMA operand precision (0 fp32: four v_mfma_f32_16x16x4_f32 per block of 16 rules and (u, v); 1 bf16 / 2 fp16: the
// four rules a lane loads ARE its four contraction slots of ONE v_mfma_f32_16x16x16_*: values rounded at use, fp32 accumulate)
template <int U, int V, int XF, int PREC = 0, int GB = (U * V <= 4 ? 4 : (U * V <= 9 ? 2 : 1))>
__global__ __launch_bounds__(64) void k_dw_pairs(DwpArgs g)
{
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    // XCD-aware shares: workgroups are dealt round-robin to the 8 XCDs (linear id % 8 = blockIdx.x % 8: S is a multiple of
    // 8), and every XCD has its own 4 MB L2.  XCD x takes the CONTIGUOUS shares [x S/8, (x+1) S/8): its L2 then holds one
    // eighth of x, dy and the lists and re-uses it for all K table rows, instead of every XCD streaming everything K times.
    const int s = (g.S & 7) == 0 ? (int)(blockIdx.x & 7u) * (g.S >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int t = blockIdx.y;
    const int ib = blockIdx.z % g.n_ci, ob = blockIdx.z / g.n_ci;
    const int ci0 = ib * 16 * U + U * r, co0 = ob * 16 * V + V * r;   // first channel of this lane on either side
    const int T = g.T, K = g.K;
    const long ntiles = (g.n_out + T - 1) / T;
    const long t_lo = ntiles * s / g.S, t_hi = ntiles * (s + 1) / g.S;
    const long words = urn_pairs_words(K, T);
    f32x4 acc[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float sc[U], sh[U];
    if constexpr (XF != 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) { sc[u] = g.xf_scale[ci0 + u]; sh[u] = g.xf_shift[ci0 + u]; }
    }
    // The blocks of table row t in this share, flattened: in chunks of 64 tiles, lane L reads where tile L's blocks of row t
    // start and how many there are (one round trip for 64 tiles), an inclusive scan over the lanes numbers the blocks, and
    // block k of the chunk is found with a ballot.  The loop over the blocks is then software-pipelined: pair words two
    // blocks ahead, rows one block ahead of the MFMAs -- a wave no longer pays header -> words -> rows -> MFMA in series
    // for every tile (measured: 94 -> 2x us at level 0).
    auto fetch_words = [&](long tile, int b, int (&pw)[4]) {
        if (g.pairs) {
            const int *blk_p = g.pairs + tile * words + URN_PAIRS_HDR + urn_pairs_tpad(K, T);
            const int4 w4 = *(const int4 *)(blk_p + (long)b * 16 + 4 * q);
            pw[0] = w4.x; pw[1] = w4.y; pw[2] = w4.z; pw[3] = w4.w;
        } else {
            const long row0 = tile * T;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const long lr = 16 * b + 4 * q + m;
                pw[m] = row0 + lr < g.n_out ? (int)((row0 + lr) | (lr << 24)) : (T << 24);
            }
        }
    };
    auto fetch_rows = [&](long tile, const int (&pw)[4], float (&a)[4][U], float (&d)[4][V]) {
        const long row0 = tile * T;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            load_n<U>(a[m], g.x + (long)(pw[m] & 0xFFFFFF) * g.ldx + ci0);
            const int dst = (unsigned)pw[m] >> 24;
            const long drow = dst < T ? row0 + dst : 0;            // padding rule: any valid row, zeroed when it is multiplied
            load_n<V>(d[m], g.dy + drow * g.ld_dy + co0);
        }
    };
    for (long c_lo = t_lo; c_lo < t_hi; c_lo += 64) {
        const long my_tile = c_lo + lane;
        int b_lo = 0, cnt = 0;
        if (my_tile < t_hi) {
            if (g.pairs) {
                const int *hdr = g.pairs + my_tile * words;
                const unsigned char *start = (const unsigned char *)(hdr + 1);   // start[t]: first block of table row t
                b_lo = start[t];
                cnt = (t + 1 < K ? (int)start[t + 1] : hdr[0]) - b_lo;
            } else {
                const long rows = g.n_out - my_tile * T < (long)T ? g.n_out - my_tile * T : (long)T;
                cnt = (int)((rows + 15) >> 4);
            }
        }
        int incl = cnt;                                             // inclusive scan over the 64 lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        auto locate = [&](int k, long &tile, int &b) {              // block k of the chunk -> (tile, block in the tile)
            const int L = __popcll(__ballot(incl <= k));            // first lane whose inclusive count exceeds k
            const int Lc = L < 63 ? L : 63;
            tile = c_lo + Lc;
            b = __builtin_amdgcn_readlane(b_lo, Lc) + k - (__builtin_amdgcn_readlane(incl, Lc) - __builtin_amdgcn_readlane(cnt, Lc));
        };
        if (total == 0) continue;
        // Two register sets (E, O) alternate as "current block" and "next block": no register copies of values that
        // were just requested (a copy would wait for them).  Per step: the padding bits of the current block's words
        // are extracted, then its word registers receive the words of block k + 2; the rows of block k + 1 are
        // requested (their words were requested one step ago); then the MFMAs of block k run on rows requested one
        // step ago.
        // Narrow layers have 4 MFMAs per block against a memory round trip of a microsecond: a step handles a GROUP of GB
        // blocks (GB x 8 row loads in flight together).
        int pw_e[GB][4], pw_o[GB][4];
        float a_e[GB][4][U], d_e[GB][4][V], a_o[GB][4][U], d_o[GB][4][V];
        long tl_e[GB], tl_o[GB];
        const int ngroups = (total + GB - 1) / GB;
        auto words_of_group = [&](int grp, int (&pw)[GB][4], long (&tl)[GB]) {
            const int gc = grp < ngroups ? grp : ngroups - 1;             // clamped: loads stay unconditional
#pragma unroll
            for (int j = 0; j < GB; ++j) {
                int b2;
                const int k = gc * GB + j;
                locate(k < total ? k : total - 1, tl[j], b2);
                fetch_words(tl[j], b2, pw[j]);                             // past the end: the last block again, masked in step()
            }
        };
        words_of_group(0, pw_e, tl_e);
        words_of_group(1, pw_o, tl_o);
#pragma unroll
        for (int j = 0; j < GB; ++j) fetch_rows(tl_e[j], pw_e[j], a_e[j], d_e[j]);
        auto step = [&](int grp, int (&pw_c)[GB][4], long (&tl_c)[GB], float (&a_c)[GB][4][U], float (&d_c)[GB][4][V],
                        int (&pw_n)[GB][4], long (&tl_n)[GB], float (&a_n)[GB][4][U], float (&d_n)[GB][4][V]) {
            bool pad[GB][4];
#pragma unroll
            for (int j = 0; j < GB; ++j)
#pragma unroll
                for (int m = 0; m < 4; ++m) pad[j][m] = ((unsigned)pw_c[j][m] >> 24) >= (unsigned)T || grp * GB + j >= total;
            words_of_group(grp + 2, pw_c, tl_c);                           // the current set becomes the set of group + 2
#pragma unroll
            for (int j = 0; j < GB; ++j) fetch_rows(tl_n[j], pw_n[j], a_n[j], d_n[j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < GB; ++j) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if constexpr (XF != 0) {
#pragma unroll
                        for (int u = 0; u < U; ++u) a_c[j][m][u] = fmaxf(fmaf(a_c[j][m][u], sc[u], sh[u]), 0.f);
                    }
#pragma unroll
                    for (int v = 0; v < V; ++v) d_c[j][m][v] = pad[j][m] ? 0.f : d_c[j][m][v];
                    if constexpr (PREC == 0) {
#pragma unroll
                        for (int u = 0; u < U; ++u)
#pragma unroll
                            for (int v = 0; v < V; ++v) acc[u][v] = MFMA16(a_c[j][m][u], d_c[j][m][v], acc[u][v]);
                    }
                }
                if constexpr (PREC != 0) {
                    typedef short s16x4 __attribute__((ext_vector_type(4)));
                    s16x4 ah[U], dh[V];
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        ah[u] = __builtin_bit_cast(s16x4, urn_round16x4<PREC>((f32x4){a_c[j][0][u], a_c[j][1][u], a_c[j][2][u], a_c[j][3][u]}));
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        dh[v] = __builtin_bit_cast(s16x4, urn_round16x4<PREC>((f32x4){d_c[j][0][v], d_c[j][1][v], d_c[j][2][v], d_c[j][3][v]}));
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            if constexpr (PREC == 1) acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[u], dh[v], acc[u][v], 0, 0, 0);
                            else {
                                typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
                                acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, ah[u]), __builtin_bit_cast(h16x4, dh[v]), acc[u][v], 0, 0, 0);
                            }
                        }
                }
            }
        };
        for (int grp = 0; grp < ngroups; grp += 2) {
            step(grp, pw_e, tl_e, a_e, d_e, pw_o, tl_o, a_o, d_o);
            if (grp + 1 < ngroups) step(grp + 1, pw_o, tl_o, a_o, d_o, pw_e, tl_e, a_e, d_e);
        }
    }
    // D[row 4q + i][col r] of MFMA (u, v) = dw[ci = 16U ib + U (4q + i) + u][co = 16V ob + V r + v]
    float *out = g.slab + ((long)s * K + t) * g.cin * g.cout;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float *p = out + (long)(ib * 16 * U + U * (4 * q + i) + u) * g.cout + co0;
#pragma unroll
            for (int v = 0; v < V; ++v) p[v] = acc[u][v][i];
        }
}

// dw[e] += slab[0][e] + slab[1][e] + ... (fixed order).  The S loads of a thread are independent: issued in groups of 8
// (a plain loop made them one dependent round trip each: 45 us for S = 64)
__global__ void k_dw_reduce(const float *__restrict__ slab, int S, long n, float *__restrict__ dw)
{
    const long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e >= n) return;
    if (e + 3 < n) {
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 8 <= S; s += 8) {
            f32x4 p[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) p[k] = *(const f32x4 *)(slab + (long)(s + k) * n + e);
#pragma unroll
            for (int k = 0; k < 8; ++k) v += p[k];
        }
        for (; s < S; ++s) v += *(const f32x4 *)(slab + (long)s * n + e);
        *(f32x4 *)(dw + e) += v;
    } else {
        for (long k = e; k < n; ++k) {
            float v = slab[k];
            for (int s = 1; s < S; ++s) v += slab[(long)s * n + k];
            dw[k] += v;
        }
    }
}

int g_dwp_dbg = 0;         // timing only: 1 = skip stage 2, 2 = skip stage 1 (urn_set_option "dwp_dbg")
int g_dwp_waves = 2048;   // stage 1 aims at this many waves (urn_set_option "dwp_waves")
int g_dwp_cap = 2;        // most 16-channel groups per wave on either side of the weight matrix ("dwp_cap": 1, 2, 4/5)
int g_dwp_smax = 16;      // ... with at most this many shares of the tiles (slab size, stage-2 traffic) ("dwp_smax")

// 16-channel groups per wave on the (cin, cout) sides and shares of the tiles.  Wide register blocks (4 x 4 groups = 64 x
// 64 weights per wave: every row is loaded once, 64 MFMAs per 8 loads) when the level is big enough to give the chip
// its waves through the shares; 2 x 2 blocks when it is not.
static void dwp_plan(int64_t n_out, int tile, int K, int cin, int cout, int &U, int &V, int &S)
{
    auto pick = [](int c16, int cap) {
        if (cap >= 5 && c16 % 5 == 0) return 5;
        for (int d : {4, 3, 2}) if (d <= cap && c16 % d == 0) return d;
        return 1;
    };
    const long ntiles = (n_out + tile - 1) / tile;
    const long smax = ntiles < g_dwp_smax ? (ntiles > 0 ? ntiles : 1) : g_dwp_smax;
    const int cap = g_dwp_cap;
    U = pick(cin / 16, cap); V = pick(cout / 16, cap);
    const long per_s = (long)K * (cin / (16 * U)) * (cout / (16 * V));
    long s = (g_dwp_waves + per_s - 1) / per_s;
    S = (int)(s < 1 ? 1 : (s > smax ? smax : s));
    if (S >= 8) S &= ~7;   // multiples of 8: XCD-aware shares (see k_dw_pairs)
}

extern "C" int64_t urn_gconv_dw_pairs_scratch_bytes(int64_t n_out, int tile, int K, int cin, int cout)
{
    if (n_out < 0 || K <= 0 || cin <= 0 || cout <= 0 || cin % 16 || cout % 16 || (tile != 32 && tile != 64 && tile != 128)) return -1;
    return (int64_t)g_dwp_smax * K * cin * cout * 4 + 256;
}

extern int g_opt_precision;   // urn_set_option("gconv_precision"): operand precision of the gather convolutions and their weight gradients
template <int U, int V>
static void launch_dwp(const DwpArgs &a, dim3 grid, hipStream_t st)
{
    const int prec = g_opt_precision;
#define URN_DWP_L(XFv) do { \
        if (prec == 1) hipLaunchKernelGGL((k_dw_pairs<U, V, XFv, 1>), grid, dim3(64), 0, st, a); \
        else if (prec == 2) hipLaunchKernelGGL((k_dw_pairs<U, V, XFv, 2>), grid, dim3(64), 0, st, a); \
        else hipLaunchKernelGGL((k_dw_pairs<U, V, XFv, 0>), grid, dim3(64), 0, st, a); } while (0)
    if (a.xf_scale) URN_DWP_L(1); else URN_DWP_L(0);
#undef URN_DWP_L
}

extern "C" int urn_gconv_bwd_dw_pairs(const float *x, int64_t ldx, const float *xf_scale, const float *xf_shift, const float *dy,
                                      int64_t ld_dy, const int32_t *pairs, int tile, int K, int64_t n_out, int cin, int cout,
                                      float *dw, void *scratch, int64_t scratch_bytes, void *stream)
{
    if (n_out <= 0) return URN_OK;
    URN_CHECK_ARG(x && dy && dw && scratch, "null pointer");
    URN_CHECK_ARG(K > 0 && K <= 27 && cin > 0 && cout > 0 && cin % 16 == 0 && cout % 16 == 0, "channel counts must be multiples of 16, K <= 27");
    URN_CHECK_ARG(tile == 32 || tile == 64 || tile == 128, "tile must be 32, 64 or 128");
    URN_CHECK_ARG(pairs != nullptr || K == 1, "a NULL list means the identity table of a 1x1 convolution");
    URN_CHECK_ARG((xf_scale == nullptr) == (xf_shift == nullptr), "scale and shift go together");
    if (ldx <= 0) ldx = cin;
    if (ld_dy <= 0) ld_dy = cout;
    URN_CHECK_ARG(ldx >= cin && ld_dy >= cout && ldx % 4 == 0 && ld_dy % 4 == 0, "row strides smaller than the rows or not multiples of 4");
    URN_CHECK_ARG(n_out < (1 << 24), "fewer than 2^24 rows");
    int U, V, S;
    dwp_plan(n_out, tile, K, cin, cout, U, V, S);
    URN_CHECK_ARG(scratch_bytes >= (int64_t)S * K * cin * cout * 4, "scratch smaller than urn_gconv_dw_pairs_scratch_bytes");
    hipStream_t st = (hipStream_t)stream;
    DwpArgs a;
    a.x = x; a.xf_scale = xf_scale; a.xf_shift = xf_shift; a.dy = dy; a.ldx = (long)ldx; a.ld_dy = (long)ld_dy; a.pairs = pairs;
    a.T = tile; a.K = K; a.S = S; a.n_out = (long)n_out; a.cin = cin; a.cout = cout; a.n_ci = cin / (16 * U); a.slab = (float *)scratch;
    const dim3 grid(S, K, a.n_ci * (cout / (16 * V)));
    const bool prof = urn_prof_on();
    if (prof) urn_prof_begin(URN_PROF_DW, st);
#define URN_DWP(Uv, Vv) if (U == Uv && V == Vv && !(g_dwp_dbg & 2)) launch_dwp<Uv, Vv>(a, grid, st);
    URN_DWP(1, 1) URN_DWP(1, 2) URN_DWP(1, 3) URN_DWP(1, 4) URN_DWP(1, 5)
    URN_DWP(2, 1) URN_DWP(2, 2) URN_DWP(2, 3) URN_DWP(2, 4) URN_DWP(2, 5)
    URN_DWP(3, 1) URN_DWP(3, 2) URN_DWP(3, 3) URN_DWP(3, 4) URN_DWP(3, 5)
    URN_DWP(4, 1) URN_DWP(4, 2) URN_DWP(4, 3) URN_DWP(4, 4) URN_DWP(4, 5)
    URN_DWP(5, 1) URN_DWP(5, 2) URN_DWP(5, 3) URN_DWP(5, 4) URN_DWP(5, 5)
#undef URN_DWP
    const long n = (long)K * cin * cout;
    if (!(g_dwp_dbg & 1)) hipLaunchKernelGGL(k_dw_reduce, dim3(urn_cdiv((n + 3) / 4, 256)), dim3(256), 0, st, (const float *)scratch, S, n, dw);
    if (prof) urn_prof_end(st);
    URN_LAUNCH_CHECK();
    return URN_OK;
}
