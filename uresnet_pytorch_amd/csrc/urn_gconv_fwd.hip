// Gather convolution forward / input-gradient on gfx950: the arithmetic behind
// scn.SubmanifoldConvolution, scn.Convolution(k2,s2), scn.Deconvolution(k2,s2) and
// scn.NetworkInNetwork (reference call sites uresnet/models/uresnet_sparse.py:21-22).
//
//   y[j,:] = sum_{o<K} T(x[tbl[t(o)*ld + j], :]) @ W[o]   (+ res[j,:])
//
// Output stationary, no atomics, bit-reproducible.  A workgroup owns one tile of MB*16
// output rows x NB*16 output columns (SPLIT=4: its four waves share the tile and split the
// ACTIVE filter offsets between them, partial tiles are summed through LDS in wave order) or
// four such tiles (SPLIT=1: one per wave).  Per tile:
//   1. all K table rows of the tile are fetched at once (7 coalesced loads per lane and
//      16-row block), wave64 ballots turn them into bit masks of the offsets that have any
//      active neighbour; inactive offsets cost nothing;
//   2. for each active offset the wave gathers its A operand straight from HBM/L2 in MFMA
//      layout (16-byte loads: 4 k-values of one gathered row per lane) and the B operand from
//      the pre-transposed, L2-resident weights (K, cout, cin); the k loop is fully unrolled
//      (KS = cin/16 is a template parameter) so every load of an offset is in flight before
//      the first v_mfma_f32_16x16x4_f32 issues;
//   3. optional fusions that remove whole HBM passes of the surrounding BatchNorm+ReLU layers:
//      T = relu(x*scale + shift) applied to the gathered rows (BatchNormReLU forward folded
//      into the load), per-tile column sums / sums of squares of y in fp64 (the NEXT
//      BatchNorm's statistics), or the BatchNorm backward reduction (sum g, sum g*xhat with
//      the ReLU mask applied) when the kernel computes an input gradient.
#include "urn_common.h"
#include "urn_prof.h"
#include "urn_gconv_int.h"
#include <stdlib.h>
#include <string.h>

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int KS, int MB, int NB, int SPLIT, int PIPE>
__global__ __launch_bounds__(256) void k_gconv_fwd(GArgs g)
{
    constexpr int CIN = KS * 16;
    __shared__ int s_idx[4][MB][28 * 16];
    __shared__ f32x4 s_red[SPLIT == 4 ? 4 * MB * NB * 64 : 1];

    const long n_out = g.n_dev ? (long)*g.n_dev : g.n_cap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const long tile = (SPLIT == 4) ? (long)blockIdx.x : (long)blockIdx.x * 4 + wave;
    const long row_base = tile * (MB * 16);
    if (SPLIT == 1 && row_base >= n_out) return;  // wave-uniform, no barriers in this variant
    const bool tile_ok = row_base < n_out;         // block-uniform when SPLIT == 4
    const int col_base = blockIdx.y * (NB * 16);
    const int K = g.K, cout = g.cout;

    // 1. the tile's table: lane (r, q) fetches table rows q, q+4, ... for its output row of every block
    unsigned amask[MB], any_mask = 0u;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        amask[mb] = 0u;
        const long row = row_base + mb * 16 + r;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int t = 4 * i + q;
            int v = -1;
            if (t < K && row < n_out) v = g.tbl[(long)t * g.ld + row];
            s_idx[wave][mb][i * 64 + lane] = v;  // == [t][r]
            const unsigned long long b = __ballot(v >= 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((b >> (16 * j)) & 0xFFFFull) amask[mb] |= 1u << (4 * i + j);
        }
        any_mask |= amask[mb];
    }

    f32x4 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool xf = g.xf_scale != nullptr;

    // 2. active offsets (table rows), split round-robin between the waves when SPLIT == 4.
    //    Software-pipelined: the operands of the NEXT active offset are in flight while the MFMAs of the
    //    current one issue (two named register sets, ping-pong, no moves).
    int cnt = 0;
    unsigned m = any_mask;
    auto next_active = [&]() -> int {
        while (m) {
            const int t = __builtin_ctz(m);
            m &= m - 1u;
            if (SPLIT == 4 && ((cnt++ & 3) != wave)) continue;
            return t;
        }
        return -1;
    };
    struct Ops { f32x4 a[MB][KS]; f32x4 b[KS][NB]; int idx[MB]; };
    auto load_ops = [&](Ops &p, int t) {
        const int o = g.flip ? (K - 1 - t) : t;  // weight index
        const float *wo = g.wt + ((long)o * cout + col_base + r) * CIN + 4 * q;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            p.idx[mb] = s_idx[wave][mb][t * 16 + r];
            const float *xa = g.x + (long)(p.idx[mb] < 0 ? 0 : p.idx[mb]) * CIN + 4 * q;
            if ((amask[mb] >> t) & 1u) {  // wave-uniform
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) p.a[mb][ks] = *(const f32x4 *)(xa + ks * 16);
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) p.b[ks][nb] = *(const f32x4 *)(wo + (long)nb * 16 * CIN + ks * 16);
    };
    auto compute = [&](Ops &p, int t) {
        if (xf) {  // BatchNormReLU folded into the load: u = relu(x*scale + shift)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f32x4 sc = *(const f32x4 *)(g.xf_scale + ks * 16 + 4 * q);
                const f32x4 sh = *(const f32x4 *)(g.xf_shift + ks * 16 + 4 * q);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    if ((amask[mb] >> t) & 1u) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) p.a[mb][ks][e] = fmaxf(fmaf(p.a[mb][ks][e], sc[e], sh[e]), 0.f);
                    }
            }
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            if (!((amask[mb] >> t) & 1u)) continue;  // wave-uniform
            if (p.idx[mb] < 0) {                     // a missing neighbour contributes zero, not T(0)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) p.a[mb][ks] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = MFMA16(p.a[mb][ks][tt], p.b[ks][nb][tt], acc[mb][nb]);
        }
    };
    if (PIPE) {
        Ops pa, pb;
        int ta = next_active(), tb = -1;
        if (ta >= 0) load_ops(pa, ta);
        while (ta >= 0) {
            tb = next_active();
            if (tb >= 0) load_ops(pb, tb);
            __builtin_amdgcn_sched_barrier(0);
            compute(pa, ta);
            if (tb < 0) break;
            ta = next_active();
            if (ta >= 0) load_ops(pa, ta);
            __builtin_amdgcn_sched_barrier(0);
            compute(pb, tb);
        }
    } else {
        Ops pa;
        for (int ta = next_active(); ta >= 0; ta = next_active()) {
            load_ops(pa, ta);
            compute(pa, ta);
        }
    }

    // 3. (SPLIT == 4) sum the four partial tiles in wave order
    if (SPLIT == 4) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) s_red[((wave * MB + mb) * NB + nb) * 64 + lane] = acc[mb][nb];
        __syncthreads();
        if (wave != 0 || !tile_ok) return;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x4 s = s_red[((0 * MB + mb) * NB + nb) * 64 + lane];
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const f32x4 p = s_red[((w * MB + mb) * NB + nb) * 64 + lane];
                    s[0] += p[0]; s[1] += p[1]; s[2] += p[2]; s[3] += p[3];
                }
                acc[mb][nb] = s;
            }
    }

    // 4. epilogue.  C layout of 16x16x4: col = lane&15, row = (lane>>4)*4 + reg
    double s0[NB], s1[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { s0[nb] = 0.0; s1[nb] = 0.0; }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = col_base + nb * 16 + r;
        float esc = 0.f, esh = 0.f, emu = 0.f, eis = 0.f;
        if (g.epi == 2) { esc = g.e_scale[col]; esh = g.e_shift[col]; emu = g.e_mean[col]; eis = g.e_invstd[col]; }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long row = row_base + mb * 16 + q * 4 + i;
                if (row >= n_out) continue;
                const long off = row * cout + col;
                float v = acc[mb][nb][i];
                if (g.res) v += g.res[off];
                if (g.epi == 1) {
                    s0[nb] += (double)v;
                    s1[nb] += (double)v * (double)v;
                } else if (g.epi == 2) {
                    const float xv = g.e_x[off];
                    if (!(fmaf(xv, esc, esh) > 0.f)) v = 0.f;             // ReLU mask of the forward
                    const double xh = ((double)xv - (double)emu) * (double)eis;
                    s0[nb] += (double)v;
                    s1[nb] += (double)v * xh;
                }
                g.y[off] = v;
            }
    }
    if (g.epi != 0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            double a0 = s0[nb], a1 = s1[nb];
            a0 += __shfl_xor(a0, 16); a1 += __shfl_xor(a1, 16);
            a0 += __shfl_xor(a0, 32); a1 += __shfl_xor(a1, 32);
            if (q == 0) {
                const int col = col_base + nb * 16 + r;
                g.part[(tile * 2 + 0) * cout + col] = a0;
                g.part[(tile * 2 + 1) * cout + col] = a1;
            }
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


// The 1-channel stem (reference uresnet_sparse.py:21: SubmanifoldConvolution(d, 1, m, 3)): y[j][:] = sum_o x[tbl[o][j]] * w[o][:].
// Thread = (row, 4 output columns); the 27 table words of the row are requested together, then the 27 input values, then
// 27 x 4 FMAs -- two round trips per row where the generic fallback (one thread per output element, table word -> value ->
// next offset) makes 54 dependent ones (25.6 us at 50k rows, plus two more launches for the BatchNorm statistics).  With
// part_slots > 0 the column sums / sums of squares of y are ADDED into row (workgroup % part_slots) of `part`
// ([slots][2][cout] fp64, the layout the consuming convolution reads: urn_gconv_args.xs_sums).
template <int CG>   // cout / 4
__global__ __launch_bounds__(256) void k_gconv_stem(const float *__restrict__ x, const float *__restrict__ wt, const int *__restrict__ tbl,
                                                    long ld, int K, long n_out, float *__restrict__ y, double *part, int part_slots)
{
    constexpr int COUT = 4 * CG, RPB = 256 / CG;           // rows per workgroup
    __shared__ float s_w[27 * COUT];
    __shared__ double s_red[4][2][COUT];
    const int tid = threadIdx.x, cg = tid % CG, rl = tid / CG;
    for (int e = tid; e < K * COUT; e += 256) s_w[e] = wt[e];     // wt[o][c] (cin == 1)
    const long row = (long)blockIdx.x * RPB + rl;
    const bool ok = row < n_out;
    int idx[27];
#pragma unroll
    for (int o = 0; o < 27; ++o) idx[o] = (ok && o < K) ? tbl[(long)o * ld + row] : -1;
    float xv[27];
#pragma unroll
    for (int o = 0; o < 27; ++o) xv[o] = x[idx[o] >= 0 ? idx[o] : 0];
    __syncthreads();
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 27; ++o) {
        const float v = idx[o] >= 0 ? xv[o] : 0.f;
        const f32x4 w = *(const f32x4 *)(s_w + o * COUT + 4 * cg);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = fmaf(v, w[k], acc[k]);
    }
    if (ok) *(f32x4 *)(y + row * COUT + 4 * cg) = acc;
    if (part_slots <= 0) return;
    double s0[4], s1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { s0[k] = ok ? (double)acc[k] : 0.0; s1[k] = ok ? (double)acc[k] * (double)acc[k] : 0.0; }
    // lanes with the same column group are CG apart: butterfly over the row bits of the lane index
#pragma unroll
    for (int m = CG; m < 64; m <<= 1)
#pragma unroll
        for (int k = 0; k < 4; ++k) { s0[k] += __shfl_xor(s0[k], m); s1[k] += __shfl_xor(s1[k], m); }
    const int lane = tid & 63, wave = tid >> 6;
    if (lane < CG)
#pragma unroll
        for (int k = 0; k < 4; ++k) { s_red[wave][0][4 * lane + k] = s0[k]; s_red[wave][1][4 * lane + k] = s1[k]; }
    __syncthreads();
    if (tid < 2 * COUT) {
        const int which = tid / COUT, col = tid - which * COUT;
        const double v = s_red[0][which][col] + s_red[1][which][col] + s_red[2][which][col] + s_red[3][which][col];
        unsafeAtomicAdd(&part[((long)(blockIdx.x % (unsigned)part_slots) * 2 + which) * COUT + col], v);
    }
}

struct Pick { int mb, nb; };

// tuning knobs (urn_set_option): software pipelining of the offset loop, and how many waves a launch must keep
// before the column tile is widened
int g_opt_precision = 0;   // default MFMA operand precision of the gather convolutions: 0 fp32, 1 bf16, 2 fp16
extern int g_net_skip_dw;
extern int g_dw_pairs;
extern int g_pairs_v3;
extern int g_dw_2stage;
extern long long *g_dense_stamps;
extern int g_dense_strided_nrb;
extern int g_net_wfrag, g_net_side2;
extern int g_tile_rb, g_tile_cb, g_tile_kc, g_tile_depth, g_dw_blocks, g_tile_il, g_dw_kernel, g_dw_split, g_dw_group, g_tile_il_min_ks, g_tile_min_wgs, g_net_side_probe, g_net_side_verbose;
static int g_opt_dbg = 0;
static long long *g_opt_stamps = nullptr;
static const int g_opt_fin_in_kernel = 0;   // (the in-kernel finalize lived in the removed 64x16 LDS kernel)
static int g_opt_pipe = 0;
static int g_opt_kernel = 7;   // 7 = compacted rule lists when the call carries them (urn_gconv_pairs.hip), else the 2-D tile; 6 = 2-D workgroup tile (urn_gconv_tile.hip); 3 = register gather (fallback for shapes without a tile instantiation)
extern int g_pairs_split_kc[9];
extern int g_pairs_lds_cap16, g_dw_rowmode;
extern int g_pairs_waves, g_pairs_waves_fwd, g_pairs_nc, g_pairs_split, g_pairs_cbg, g_pairs_wgs, g_pairs_wgs16, g_dwp_waves, g_dwp_smax, g_dwp_dbg, g_dwp_cap;
// which calls that carry a pair list run on the pair-list kernel (measured per shape on the cfg3 geometry, tools/bench_pairs.py:
// it wins for the strided pair and the narrow levels; the wide, small levels are faster on the LDS-staged 2-D tile kernel):
// K == 8, or K == 27 with cin <= g_pairs_max_cin and cout <= g_pairs_max_cout; K == 1 only when g_pairs_nin is set
static int g_pairs_max_cin = 80, g_pairs_max_cout = 999, g_pairs_nin = 0;
// with fragment-ordered weights (urn_gconv_args.wt_frag) the wide inputs win on the pair lists too (96 -> 48: 33 us against 43 on
// the tile kernel and 42 without fragments; 128 -> 64: 40 against 41 and 49)
static int g_pairs_frag_wide = 1;
static int g_pairs_prec = 1;        // reduced-precision calls on the pair lists too (urn_set_option "pairs_prec"; 0: the 2-D tile kernel as before)
static long g_opt_min_waves = 8192;

extern "C" int urn_set_option(const char *key, int64_t value)
{
    URN_CHECK_ARG(key, "null key");
    if (!strcmp(key, "gconv_pipe")) { g_opt_pipe = value != 0; return URN_OK; }
    if (!strcmp(key, "gconv_min_waves")) { g_opt_min_waves = value; return URN_OK; }
    if (!strcmp(key, "gconv_kernel")) { g_opt_kernel = (int)value; return URN_OK; }
    if (!strncmp(key, "pairs_split_kc", 14) && key[14] >= '1' && key[14] <= '8' && !key[15]) { g_pairs_split_kc[key[14] - '0'] = (int)value; return URN_OK; }
    if (!strcmp(key, "pairs_waves")) { g_pairs_waves = (int)value; return URN_OK; }
    if (!strcmp(key, "pairs_frag_wide")) { g_pairs_frag_wide = value != 0; return URN_OK; }
    if (!strcmp(key, "pairs_prec")) { g_pairs_prec = value != 0; return URN_OK; }
    if (!strcmp(key, "pairs_max_cin")) { g_pairs_max_cin = (int)value; return URN_OK; }
    if (!strcmp(key, "pairs_max_cout")) { g_pairs_max_cout = (int)value; return URN_OK; }
    if (!strcmp(key, "pairs_nin")) { g_pairs_nin = (int)value; return URN_OK; }
    if (!strcmp(key, "pairs_nc")) { g_pairs_nc = (int)value; return URN_OK; }
    if (!strcmp(key, "pairs_cbg")) { g_pairs_cbg = (int)value; return URN_OK; }
    if (!strcmp(key, "dw_rowmode")) { g_dw_rowmode = (int)value; return URN_OK; }
    if (!strcmp(key, "pairs_lds_cap16")) { g_pairs_lds_cap16 = (int)value; return URN_OK; }
    if (!strcmp(key, "pairs_wgs")) { g_pairs_wgs = (int)value; return URN_OK; }
    if (!strcmp(key, "pairs_waves_fwd")) { g_pairs_waves_fwd = (int)value; return URN_OK; }
    if (!strcmp(key, "pairs_wgs16")) { g_pairs_wgs16 = (int)value; return URN_OK; }
    if (!strcmp(key, "pairs_v3")) { g_pairs_v3 = (int)value; return URN_OK; }
    if (!strcmp(key, "dw_pairs")) { g_dw_pairs = value != 0; return URN_OK; }
    if (!strcmp(key, "net_dbg_skip_dw")) { g_net_skip_dw = value != 0; return URN_OK; }
    if (!strcmp(key, "net_wfrag")) { g_net_wfrag = value != 0; return URN_OK; }
    if (!strcmp(key, "net_side2")) { g_net_side2 = value != 0; return URN_OK; }
    if (!strcmp(key, "dw_2stage")) { g_dw_2stage = value != 0; return URN_OK; }
    if (!strcmp(key, "dwp_cap")) { g_dwp_cap = value >= 1 && value <= 5 ? (int)value : 2; return URN_OK; }
    if (!strcmp(key, "dwp_dbg")) { g_dwp_dbg = (int)value; return URN_OK; }
    if (!strcmp(key, "dwp_waves")) { g_dwp_waves = (int)value; return URN_OK; }
    if (!strcmp(key, "dwp_smax")) { g_dwp_smax = value > 0 && value <= 256 ? (int)value : 16; return URN_OK; }
    if (!strcmp(key, "pairs_split")) { g_pairs_split = (int)value; return URN_OK; }
    if (!strcmp(key, "dense_strided_nrb")) { g_dense_strided_nrb = value == 16 ? 16 : 4; return URN_OK; }
    if (!strcmp(key, "dense_stamp_ptr")) { g_dense_stamps = (long long *)(uintptr_t)value; return URN_OK; }
    if (!strcmp(key, "gconv_stamp_ptr")) { g_opt_stamps = (long long *)(uintptr_t)value; return URN_OK; }
    if (!strcmp(key, "gconv_dbg")) { g_opt_dbg = (int)value; return URN_OK; }
    if (!strcmp(key, "tile_rb")) { g_tile_rb = (int)value; return URN_OK; }
    if (!strcmp(key, "tile_cb")) { g_tile_cb = (int)value; return URN_OK; }
    if (!strcmp(key, "tile_kc")) { g_tile_kc = (int)value; return URN_OK; }
    if (!strcmp(key, "tile_depth")) { g_tile_depth = (int)value; return URN_OK; }
    if (!strcmp(key, "tile_il")) { g_tile_il = (int)value; return URN_OK; }
    if (!strcmp(key, "tile_min_wgs")) { g_tile_min_wgs = (int)value; return URN_OK; }
    if (!strcmp(key, "tile_il_min_ks")) { g_tile_il_min_ks = (int)value; return URN_OK; }
    if (!strcmp(key, "gconv_precision")) { g_opt_precision = value >= 0 && value <= 2 ? (int)value : 0; return URN_OK; }
    if (!strcmp(key, "dw_kernel")) { g_dw_kernel = value == 1 ? 1 : 2; return URN_OK; }
    if (!strcmp(key, "dw_split")) { g_dw_split = (int)value; return URN_OK; }
    if (!strcmp(key, "net_side_probe")) { g_net_side_probe = (int)value; return URN_OK; }
    if (!strcmp(key, "net_side_verbose")) { g_net_side_verbose = (int)value; return URN_OK; }
    if (!strcmp(key, "dw_group")) { g_dw_group = value > 0 ? (int)value : 1; return URN_OK; }
    if (!strcmp(key, "dw_blocks")) { g_dw_blocks = value > 0 ? (int)value : 768; return URN_OK; }
    urn_set_error("urn_set_option: unknown key %s", key);
    return URN_EINVAL;
}

// Tile shape.  Measured on MI355X: these kernels are bound by the per-wave latency chain (table fetch ->
// gather -> MFMA), so more, smaller waves win over wider register tiles; columns are widened (fewer
// re-gathers of the A rows) only while the launch keeps >= 8 waves per SIMD.
static Pick pick_tile(int ks, int nblk, long n_out, bool split)
{
    const long tiles = (n_out + 15) / 16;
    for (int nb = 5; nb >= 2; --nb) {
        if (nblk % nb) continue;
        if (2 * (4 * ks * (1 + nb)) + 4 * nb > 200) continue;   // two operand sets (software pipeline)
        if (tiles * (split ? 4 : 1) * (nblk / nb) >= g_opt_min_waves) return Pick{1, nb};
    }
    return Pick{1, 1};
}

template <int KS, int MB, int NB>
static void launch_tile(const GArgs &a, long n_out, bool split, hipStream_t st)
{
    const long tiles = (n_out + MB * 16 - 1) / (MB * 16);
    const int gy = a.cout / (NB * 16);
    if (split && g_opt_pipe)
        hipLaunchKernelGGL((k_gconv_fwd<KS, MB, NB, 4, 1>), dim3((unsigned)tiles, gy), dim3(256), 0, st, a);
    else if (split)
        hipLaunchKernelGGL((k_gconv_fwd<KS, MB, NB, 4, 0>), dim3((unsigned)tiles, gy), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((k_gconv_fwd<KS, MB, NB, 1, 0>), dim3((unsigned)((tiles + 3) / 4), gy), dim3(256), 0, st, a);
}

template <int KS>
static bool launch_ks(const GArgs &a, long n_out, Pick p, bool split, hipStream_t st)
{
#define URN_T(MBv, NBv)                                                     \
    if (p.mb == MBv && p.nb == NBv) {                                       \
        if constexpr (8 * KS * (MBv + NBv) + 4 * MBv * NBv <= 200) {        \
            launch_tile<KS, MBv, NBv>(a, n_out, split, st);                 \
            return true;                                                    \
        }                                                                   \
    }
    URN_T(1, 1) URN_T(1, 2) URN_T(1, 3) URN_T(1, 4) URN_T(1, 5)
#undef URN_T
    return false;
}

// the finalize of the epilogue partials as separate launches
static int finalize_launches(const urn_gconv_args *u, int np, void *stream)
{
    if (u->epilogue == 1) {
        for (int i = 0; i < 2; ++i) {
            if (!u->fin_bn[i].mean) continue;
            int rc = urn_bn_finalize_fwd(u->part, np, u->fin_n, u->cout, u->cout, u->fin_eps, u->fin_bn[i].gamma, u->fin_bn[i].beta,
                                         u->fin_bn[i].mean, u->fin_bn[i].invstd, u->fin_bn[i].scale, u->fin_bn[i].shift,
                                         u->fin_bn[i].running_mean, u->fin_bn[i].running_var, u->fin_momentum, stream);
            if (rc) return rc;
        }
        return URN_OK;
    }
    return urn_bn_finalize_bwd(u->part, np, u->fin_n, u->cout, u->fin_dgamma, u->fin_dbeta, u->fin_coef0, u->fin_coef1, stream);
}

extern "C" int urn_gconv_fwd_ex(const urn_gconv_args *u, int *n_tiles, void *stream)
{
    URN_CHECK_ARG(u, "null args");
    if (n_tiles) *n_tiles = 0;
    if (u->n_out <= 0) return URN_OK;
    URN_CHECK_ARG(u->x && u->wt && u->tbl && u->y, "null pointer");
    URN_CHECK_ARG(u->K > 0 && u->cin > 0 && u->cout > 0 && u->ld >= u->n_out, "bad shape");
    URN_CHECK_ARG((const void *)u->x != (const void *)u->y, "y aliases x");
    URN_CHECK_ARG(u->epilogue == 0 || u->part, "epilogue needs a partial slab");
    URN_CHECK_ARG(u->epilogue != 2 || (u->e_x && u->e_scale && u->e_shift && u->e_mean && u->e_invstd), "epilogue 2 needs the BatchNorm inputs");
    hipStream_t st = (hipStream_t)stream;
    const int ks = u->cin / 16;
    const bool mfma_ok = (u->cin % 16 == 0) && (u->cout % 16 == 0) && u->K <= 28;
    if (!mfma_ok && u->cin == 1 && (u->cout == 16 || u->cout == 32 || u->cout == 64) && u->K <= 27 && !u->flip && !u->res && !u->xf_scale &&
        !u->xs_sums[0] && (u->epilogue == 0 || (u->epilogue == 1 && u->part_slots > 0 && u->part)) && (u->ldx == 0 || u->ldx == 1) &&
        (u->ldy == 0 || u->ldy == u->cout)) {
        // the 1-channel stem, with the statistics of its output accumulated for the consuming BatchNorm
        const int slots = u->epilogue == 1 ? u->part_slots : 0;
        const int cg = u->cout / 4, rpb = 256 / cg;
        const dim3 grid((unsigned)urn_cdiv(u->n_out, rpb));
        if (cg == 4) hipLaunchKernelGGL(k_gconv_stem<4>, grid, dim3(256), 0, st, u->x, u->wt, u->tbl, (long)u->ld, u->K, (long)u->n_out, u->y, u->part, slots);
        else if (cg == 8) hipLaunchKernelGGL(k_gconv_stem<8>, grid, dim3(256), 0, st, u->x, u->wt, u->tbl, (long)u->ld, u->K, (long)u->n_out, u->y, u->part, slots);
        else hipLaunchKernelGGL(k_gconv_stem<16>, grid, dim3(256), 0, st, u->x, u->wt, u->tbl, (long)u->ld, u->K, (long)u->n_out, u->y, u->part, slots);
        if (n_tiles) *n_tiles = slots;
        URN_LAUNCH_CHECK();
        return URN_OK;
    }
    if (!mfma_ok) {
        if (u->xf_scale || u->epilogue) { urn_set_error("urn_gconv_fwd_ex: fusions need channel counts that are multiples of 16"); return URN_EUNSUPPORTED; }
        hipLaunchKernelGGL(k_gconv_small, dim3(urn_cdiv(u->n_out * u->cout, 256)), dim3(256), 0, st, u->x, u->wt, u->tbl,
                           (long)u->ld, u->K, u->flip, (long)u->n_out, u->cin, u->cout, u->res, u->y);
        URN_LAUNCH_CHECK();
        return URN_OK;
    }
    static const bool env_once = [] {
        if (const char *e = getenv("URN_GCONV_KERNEL")) g_opt_kernel = atoi(e);   // A/B switch: 7 pair lists, 6 2-D tile, 3 register
        return true;
    }();
    (void)env_once;
    const bool split = u->K >= 8;
    const Pick p = pick_tile(ks, u->cout / 16, u->n_out, split);
    GArgs a;
    memset(&a, 0, sizeof(a));
    a.x = u->x; a.wt = u->wt; a.wfrag = u->wt_frag; a.wfrag_prec = u->wt_frag_prec; a.tbl = u->tbl; a.ld = (long)u->ld; a.K = u->K; a.flip = u->flip; a.n_cap = (long)u->n_out;
    a.cout = u->cout; a.cin = u->cin; a.res = u->res; a.y = u->y; a.xf_scale = u->xf_scale; a.xf_shift = u->xf_shift; a.epi = u->epilogue;
    a.part = u->part; a.e_x = u->e_x; a.e_scale = u->e_scale; a.e_shift = u->e_shift; a.e_mean = u->e_mean;
    a.e_invstd = u->e_invstd; a.dbg = g_opt_dbg; a.stamps = g_opt_stamps;
    a.prec = u->precision > 0 ? u->precision - 1 : g_opt_precision;   // 0 fp32, 1 bf16, 2 fp16
    a.ldx = u->ldx > 0 ? (long)u->ldx : (long)u->cin;
    a.ldy = u->ldy > 0 ? (long)u->ldy : (long)u->cout;
    const bool strided = a.ldx != u->cin || a.ldy != u->cout;
    // the interleaved tile kernel addresses gathered rows with 32-bit byte offsets: rows (< ld) * ldx * 4 must fit
    const bool off32_ok = (double)u->ld * (double)a.ldx * 4.0 < 4294967296.0 && (double)u->K * u->cin * u->cout * 4.0 < 2147483648.0;
    URN_CHECK_ARG(a.ldx >= u->cin && a.ldy >= u->cout && a.ldx % 4 == 0, "ldx / ldy smaller than the row, or ldx not a multiple of 4");
    // finalize requested?  In-kernel (last workgroup) only on request: measured on MI355X the tail work (every
    // workgroup drains its stores and takes a ticket, the last one reduces the slab while the chip idles) costs
    // more than the separate finalize launch it saves (6.58 vs 6.14 ms per cfg3 step), so the default is launches.
    const bool want_fin = u->epilogue != 0 && (u->fin_bn[0].mean != nullptr || u->fin_dgamma != nullptr);
    const bool in_kernel = want_fin && u->sync_word != nullptr && g_opt_fin_in_kernel;
    if (want_fin) {
        a.sync_word = in_kernel ? u->sync_word : nullptr; a.fin_n = (long)u->fin_n; a.fin_eps = u->fin_eps; a.fin_momentum = u->fin_momentum;
        for (int i = 0; i < 2; ++i) {
            a.fin_bn[i].gamma = u->fin_bn[i].gamma; a.fin_bn[i].beta = u->fin_bn[i].beta; a.fin_bn[i].mean = u->fin_bn[i].mean;
            a.fin_bn[i].invstd = u->fin_bn[i].invstd; a.fin_bn[i].scale = u->fin_bn[i].scale; a.fin_bn[i].shift = u->fin_bn[i].shift;
            a.fin_bn[i].running_mean = u->fin_bn[i].running_mean; a.fin_bn[i].running_var = u->fin_bn[i].running_var;
        }
        a.fin_dgamma = u->fin_dgamma; a.fin_dbeta = u->fin_dbeta; a.fin_coef0 = u->fin_coef0; a.fin_coef1 = u->fin_coef1;
    }
    // accumulated statistics: only the 2-D tile kernel implements them
    const bool sums_mode = u->part_slots > 0 || u->xs_sums[0] != nullptr;
    if (sums_mode) {
        URN_CHECK_ARG(u->part_slots >= 0 && (u->part_slots == 0 || u->epilogue != 0), "part_slots without an epilogue");
        if (u->xs_sums[0]) {
            URN_CHECK_ARG(u->xs_slots > 0 && u->xs_split > 0 && u->xs_split <= u->cin && u->xs_gamma && u->xs_beta && u->xs_mean &&
                          u->xs_invstd && u->xs_scale && u->xs_shift && (u->xs_split == u->cin || u->xs_sums[1]), "incomplete xs_* fields");
        }
        a.part_slots = u->part_slots; a.xs_slots = u->xs_slots; a.xs_split = u->xs_split;
        a.xs_ld[0] = u->xs_ld[0]; a.xs_ld[1] = u->xs_ld[1]; a.xs_sums[0] = u->xs_sums[0]; a.xs_sums[1] = u->xs_sums[1];
        a.xs_n = (long)u->xs_n; a.xs_gamma = u->xs_gamma; a.xs_beta = u->xs_beta; a.xs_mean = u->xs_mean; a.xs_invstd = u->xs_invstd;
        a.xs_scale = u->xs_scale; a.xs_shift = u->xs_shift; a.xs_rm = u->xs_running_mean; a.xs_rv = u->xs_running_var;
        a.fin_eps = u->fin_eps; a.fin_momentum = u->fin_momentum;
    }
    const bool prof = urn_prof_on();
    if (prof) urn_prof_begin(URN_PROF_GCONV, st);
    bool ok = false;
    if (!off32_ok && (sums_mode || strided)) {
        urn_set_error("urn_gconv_fwd_ex: accumulated statistics / strided operands need the 2-D tile kernel, whose 32-bit row offsets "
                      "do not cover ld=%ld x ldx=%ld", (long)u->ld, a.ldx);
        return URN_EUNSUPPORTED;
    }
    // compacted rule lists: the call carries the list of its table (or asks for the identity list of a 1x1 convolution)
    const bool pairs_shape = u->K == 8 || (u->K == 1 && g_pairs_nin) || (u->K > 8 && (u->cin <= g_pairs_max_cin || (u->wt_frag && g_pairs_frag_wide)) && u->cout <= g_pairs_max_cout);
    if (g_opt_kernel >= 7 && off32_ok && u->pairs_tile != 0 && (a.prec == 0 || g_pairs_prec) && !in_kernel && pairs_shape && (u->pairs != nullptr || u->K == 1)) {
        a.pairs = u->pairs; a.p_tile = u->pairs_tile;
        const int npp = urn_gconv_pairs_launch(a, u->n_out, st);
        if (npp > 0) {
            if (prof) urn_prof_end(st);
            const int np = u->part_slots > 0 ? u->part_slots : npp;
            if (n_tiles) *n_tiles = np;
            URN_LAUNCH_CHECK();
            if (want_fin) return finalize_launches(u, np, stream);
            return URN_OK;
        }
        a.pairs = nullptr; a.p_tile = 0;
    }
    if (off32_ok && ((g_opt_kernel >= 6 && !in_kernel) || sums_mode || a.prec || strided)) {
        const int np6 = urn_gconv_tile_launch(a, ks, u->n_out, st);
        if (np6 > 0) {
            if (prof) urn_prof_end(st);
            const int np = u->part_slots > 0 ? u->part_slots : np6;
            if (n_tiles) *n_tiles = np;
            URN_LAUNCH_CHECK();
            if (want_fin) return finalize_launches(u, np, stream);
            return URN_OK;
        }
        if (strided) {
            if (prof) urn_prof_end(st);
            urn_set_error("urn_gconv_fwd_ex: strided x / y need the 2-D tile kernel (cin=%d cout=%d has none)", u->cin, u->cout);
            return URN_EUNSUPPORTED;
        }
        if (sums_mode) {
            if (prof) urn_prof_end(st);
            urn_set_error("urn_gconv_fwd_ex: accumulated statistics need the 2-D tile kernel (cin=%d cout=%d has none)", u->cin, u->cout);
            return URN_EUNSUPPORTED;
        }
    }
    a.sync_word = nullptr;   // the register-gather kernel below has no in-kernel finalize
    switch (ks) {
    case 1: ok = launch_ks<1>(a, u->n_out, p, split, st); break;
    case 2: ok = launch_ks<2>(a, u->n_out, p, split, st); break;
    case 3: ok = launch_ks<3>(a, u->n_out, p, split, st); break;
    case 4: ok = launch_ks<4>(a, u->n_out, p, split, st); break;
    case 5: ok = launch_ks<5>(a, u->n_out, p, split, st); break;
    case 6: ok = launch_ks<6>(a, u->n_out, p, split, st); break;
    case 8: ok = launch_ks<8>(a, u->n_out, p, split, st); break;
    case 10: ok = launch_ks<10>(a, u->n_out, p, split, st); break;
    case 12: ok = launch_ks<12>(a, u->n_out, p, split, st); break;
    case 14: ok = launch_ks<14>(a, u->n_out, p, split, st); break;
    default: ok = false; break;
    }
    if (prof) urn_prof_end(st);
    if (!ok) { urn_set_error("urn_gconv_fwd_ex: no kernel for cin=%d tile %dx%d", u->cin, p.mb, p.nb); return URN_EUNSUPPORTED; }
    const int np = (int)((u->n_out + p.mb * 16 - 1) / (p.mb * 16));
    if (n_tiles) *n_tiles = np;
    URN_LAUNCH_CHECK();
    if (want_fin) return finalize_launches(u, np, stream);
    return URN_OK;
}

extern "C" int urn_gconv_fwd(const float *x, const float *wt, const int32_t *tbl, int64_t ld, int K, int flip,
                             int64_t n_out, int cin, int cout, const float *res, float *y, void *stream)
{
    urn_gconv_args u;
    memset(&u, 0, sizeof(u));
    u.x = x; u.wt = wt; u.tbl = tbl; u.ld = ld; u.K = K; u.flip = flip; u.n_out = n_out; u.cin = cin; u.cout = cout;
    u.res = res; u.y = y;
    return urn_gconv_fwd_ex(&u, nullptr, stream);
}

extern "C" int64_t urn_gconv_part_bytes(int64_t n_out, int cout)
{
    return ((n_out + 15) / 16) * 2 * (int64_t)cout * 8 + 256;
}
