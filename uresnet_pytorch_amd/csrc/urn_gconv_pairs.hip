// Gather convolution over COMPACTED rule lists ("pair lists"), wave-autonomous, no barriers in the offset loop.
//
// Why: the 2-D tile kernel (urn_gconv_tile.hip) is output-stationary over 16-row blocks and executes every offset that
// is active for ANY row of the block: on the cfg3 geometry that is 2.4-3.4x the rules that exist (R/N = 4.4-10 of 27
// offsets, but a 16-row block touches 14-24 of them), and its offset step is one workgroup barrier per 4-16 MFMAs.
// Here the rules of a tile of T output rows are compacted per table row t (= filter offset) into dense blocks of 16
// (in_row, out_row) pairs by the integer phase (k_pairs_build below): 1.2-1.6x the rules instead of 2.4-3.4x, and for
// the strided pair (one parent per fine row) 1.4-2x instead of 7x.
//
// One wave owns 16*NC output columns of a tile and a contiguous share [b0, b1) of the tile's block list (the list is
// split G ways so that deep, small levels still fill the chip).  Per block it gathers the 16 input rows straight into
// MFMA fragments (global -> registers, 64 B per row and 16-channel group), keeps the weight fragments of the current
// offset in registers (reloaded only when the offset changes), runs v_mfma_f32_16x16x4_f32 with the weights as the A
// operand (lane (r, q) then holds 4 consecutive output columns of pair r) and adds the result into its PRIVATE fp32 slab
// in LDS with one 16-byte read-modify-write per column block at the rows' positions in the tile (LDS float atomics were
// 3-5x slower).  A wave's adds execute in program order and no two waves share a slab element, so the sum order is fixed
// by the list: results are bitwise reproducible.  After one workgroup barrier the G partial slabs are summed in order g = 0..G-1 and the epilogue
// (residual, BatchNorm statistics or BatchNorm-backward reduce, exactly as in the tile kernel) writes y.
//
// Padding pairs of a partly filled block gather row 0 and add into a trash row (index T) of the slab: no masks.
#include "urn_common.h"
#include "urn_gconv_int.h"
#include "urn_prof.h"
#include <string.h>

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// timing-only ablation switches (urn_set_option "gconv_dbg"; tools/bench_pairs.py abl) exist in -DURN_DIAG builds only
// (make EXTRA=-DURN_DIAG): the product kernels carry no diagnostic branch
#ifdef URN_DIAG
#define URN_DBG(g, bits) ((g).dbg & (bits))
#else
#define URN_DBG(g, bits) 0
#endif

// ------------------------------------------------------------------------------------------------ list builder
#define URN_PAIRS_MAX_TABLES 16
struct PairsDesc { const int *tbl; long ld; int K; const int *n_dev; long n_cap; int T; int *out; int ntiles; };
struct PairsDescs { int n; PairsDesc d[URN_PAIRS_MAX_TABLES]; };

// KK = compile-time table height (27 / 8; 0 = generic): with it the K table words of a row are requested together and the
// loop over the table rows is straight-line code (a load inside a rolled loop is a round trip per table row)
// The file is compiled five times (Makefile: -DURN_PAIRS_PART=0..4) so that the build parallelises: part 0 holds the host
// code and the fp32 kernels, parts 1 / 2 the bf16 kernels (one or two | four column blocks per wave), parts 3 / 4 the fp16 ones.
#ifndef URN_PAIRS_PART
#define URN_PAIRS_PART 0
#endif
#if URN_PAIRS_PART == 0
template <int KK>
__device__ __forceinline__ void pairs_build_tile(const PairsDesc &d, int *s_cnt)
{
    const int T = d.T, K = KK ? KK : d.K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long n_out = d.n_dev ? (long)*d.n_dev : d.n_cap;
    const long row = (long)blockIdx.x * T + tid;
    int *out = d.out + (long)blockIdx.x * urn_pairs_words(K, T);
    int *out_t = out + URN_PAIRS_HDR, *out_p = out_t + urn_pairs_tpad(K, T);
    int nb = 0;
    constexpr int NV = KK ? KK : 1;
    int vals[NV];
    if constexpr (KK != 0) {
#pragma unroll
        for (int t = 0; t < KK; ++t) vals[t] = (tid < T && row < n_out) ? d.tbl[(long)t * d.ld + row] : -1;
    }
    auto row_of_table = [&](int t, int v) {
        const unsigned long long b = __ballot(v >= 0);
        const int my = __popcll(b & ((1ull << lane) - 1ull)), wcnt = __popcll(b);
        int base = 0, total = wcnt;
        if (T == 128) {
            if (lane == 0) s_cnt[wave] = wcnt;
            __syncthreads();
            base = wave ? s_cnt[0] : 0;
            total = s_cnt[0] + s_cnt[1];
            __syncthreads();
        }
        if (v >= 0) out_p[(long)nb * 16 + base + my] = v | (tid << 24);
        const int nblk_t = (total + 15) >> 4;
        if (tid < nblk_t * 16 - total) out_p[(long)nb * 16 + total + tid] = T << 24;   // padding: gather row 0, add into the trash row
        if (tid < nblk_t) out_t[nb + tid] = t;
        if (tid == 0) ((unsigned char *)(out + 1))[t] = (unsigned char)nb;   // first block of table row t (the weight gradient walks by t)
        nb += nblk_t;
    };
    if constexpr (KK != 0) {
#pragma unroll
        for (int t = 0; t < KK; ++t) row_of_table(t, vals[t]);
    } else {
        for (int t = 0; t < K; ++t) row_of_table(t, (tid < T && row < n_out) ? d.tbl[(long)t * d.ld + row] : -1);
    }
    if (tid == 0) out[0] = nb;
}

__global__ __launch_bounds__(128) void k_pairs_build(PairsDescs ds)
{
    const PairsDesc d = ds.d[blockIdx.y];
    if ((int)blockIdx.x >= d.ntiles) return;
    if (d.T <= 64 && threadIdx.x >= 64) return;   // one wave per tile of 32 / 64 rows (no barrier is executed for them)
    __shared__ int s_cnt[2];
    if (d.K == 27) pairs_build_tile<27>(d, s_cnt);
    else if (d.K == 8) pairs_build_tile<8>(d, s_cnt);
    else pairs_build_tile<0>(d, s_cnt);
}

extern "C" int64_t urn_pairs_bytes(int64_t n_cap, int K, int tile)
{
    if (n_cap < 0 || K <= 0 || (tile != 32 && tile != 64 && tile != 128)) return -1;
    const int64_t ntiles = (n_cap + tile - 1) / tile;
    return (ntiles > 0 ? ntiles : 1) * urn_pairs_words(K, tile) * 4;
}

extern "C" int urn_pairs_build(int n_tables, const int32_t *const *tbl, const int64_t *ld, const int *K,
                               const int32_t *const *n_dev, const int64_t *n_cap, const int *tile, int32_t *const *pairs,
                               void *stream)
{
    URN_CHECK_ARG(n_tables >= 0 && (n_tables == 0 || (tbl && ld && K && n_cap && tile && pairs)), "null pointer");
    for (int base = 0; base < n_tables; base += URN_PAIRS_MAX_TABLES) {
        PairsDescs ds;
        ds.n = n_tables - base < URN_PAIRS_MAX_TABLES ? n_tables - base : URN_PAIRS_MAX_TABLES;
        int max_tiles = 0;
        for (int i = 0; i < ds.n; ++i) {
            const int j = base + i;
            URN_CHECK_ARG(tbl[j] && pairs[j] && K[j] > 0 && K[j] <= 27 && (tile[j] == 32 || tile[j] == 64 || tile[j] == 128) && ld[j] >= n_cap[j] &&
                          n_cap[j] >= 0 && n_cap[j] < (1 << 24), "bad table (K <= 27, tile 32, 64 or 128, fewer than 2^24 rows)");
            PairsDesc &d = ds.d[i];
            d.tbl = tbl[j]; d.ld = (long)ld[j]; d.K = K[j]; d.n_dev = n_dev ? n_dev[j] : nullptr; d.n_cap = (long)n_cap[j];
            d.T = tile[j]; d.out = pairs[j]; d.ntiles = (int)((n_cap[j] + tile[j] - 1) / tile[j]);
            if (d.ntiles > max_tiles) max_tiles = d.ntiles;
        }
        if (max_tiles == 0) continue;
        hipLaunchKernelGGL(k_pairs_build, dim3(max_tiles, ds.n), dim3(128), 0, (hipStream_t)stream, ds);
        URN_LAUNCH_CHECK();
    }
    return URN_OK;
}

#endif   // URN_PAIRS_PART == 0

typedef short urn_s16x4 __attribute__((ext_vector_type(4)));
template <int PREC>
__device__ __forceinline__ urn_s16x4 pairs_cvt16(f32x4 v)
{
    return __builtin_bit_cast(urn_s16x4, urn_round16x4<PREC>(v));
}
// registers of one weight fragment: 16 bytes per lane in fp32, 8 with 16-bit fragments (urn_gconv_args.wt_frag_prec)
template <int PREC> struct PairsW { typedef f32x4 type; };
template <> struct PairsW<1> { typedef urn_s16x4 type; };
template <> struct PairsW<2> { typedef urn_s16x4 type; };
// two 16-channel groups at once: the lane's eight contraction slots = [group j | group j + 1] on BOTH operands (gfx950)
template <int PREC>
__device__ __forceinline__ f32x4 pairs_mfma32(urn_s16x4 a0, urn_s16x4 a1, urn_s16x4 b0, urn_s16x4 b1, f32x4 c)
{
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7), b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    if constexpr (PREC == 1) {
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    } else {
        typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
    }
}
template <int PREC>
__device__ __forceinline__ f32x4 pairs_mfma16(urn_s16x4 a, urn_s16x4 b, f32x4 c)
{
    if constexpr (PREC == 1) return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    else {
        typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, a), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------ convolution
// KC: 16-channel groups per contraction chunk (cin = 16 * KC * nch); NC: 16-column blocks per wave; XF: rows are used as
// relu(x * scale + shift) (folded BatchNorm+ReLU of the input)
// PREC: MFMA operand precision, 0 fp32 (v_mfma_f32_16x16x4_f32), 1 bf16, 2 fp16 (v_mfma_f32_16x16x16_*: the 16 bytes a lane
// gathers -- channels 4q..4q+3 of its pair -- rounded to 16 bits ARE its B operand, the weight fragment likewise its A
// operand: one MFMA per 16-channel group instead of four; rows and weights stay fp32 in HBM, accumulation is fp32)
// (register estimate of the loop: rows as loaded + operand registers 8 KC, weight fragments 4 KC NC)
#define URN_PAIRS_REGS(KC, NC, PREC) ((KC) * ((NC) + 2) * 4)
template <int KC, int NC, int XF, int DEEP, int PREC = 0>
__global__ __launch_bounds__(URN_PAIRS_REGS(KC, NC, PREC) <= 64 ? 1024 : 512, (KC <= 2 && NC == 1) ? (DEEP == 2 ? 4 : 5) : 1) void k_gconv_pairs(GArgs g)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // diagnostics, compiled only with -DURN_PAIRS_STAMP (make CXXFLAGS+=...): s_memtime at the phase boundaries of every wave
    // (urn_set_option "gconv_stamp_ptr"; tools/stamp_pairs.py)
#ifdef URN_PAIRS_STAMP
    long long *stamp = g.stamps ? g.stamps + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 : nullptr;
#define URN_STAMP(i) do { if (stamp && (threadIdx.x & 63) == 0) stamp[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define URN_STAMP(i) do { } while (0)
#endif
    URN_STAMP(0);
    const int T = g.p_tile, G = g.p_split;
    const int cin = g.cin, cout = g.cout, K = g.K;
    const int nch = cin / (16 * KC);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
    const int r = lane & 15, q = lane >> 4;
    const int cw = g.p_cw;                         // columns of this workgroup (blockIdx.y selects the column range)
    const int CBG = cw / (16 * NC);
    const int cg = wave % CBG, gi = wave / CBG;
    // XCD-aware work assignment.  Workgroups are dealt round-robin to the 8 XCDs in launch order (x fastest, then y), and every
    // XCD has its own L2: XCD x takes a contiguous range of the (tile, column group) items, tile-major -- the column groups of
    // a tile gather the SAME rows and now do so through one L2, neighbouring tiles (whose rows overlap) likewise.  (Mapping the
    // tiles alone, by blockIdx.x, put the column groups of a tile on different XCDs whenever the tile count is not a multiple
    // of 8: every row of the deep levels was fetched by four or five L2s.)
    unsigned tile, cgroup;
    {
        const unsigned gy = gridDim.y, total = gridDim.x * gy, lin = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned xq = total >> 3, xr = total & 7u, xcd = lin & 7u, slot = lin >> 3;
        const unsigned v = xcd * xq + (xcd < xr ? xcd : xr) + slot;
        tile = v / gy; cgroup = v - tile * gy;
    }
    const int colw = (int)cgroup * cw;             // first column of the workgroup
    const int lcol0 = cg * 16 * NC;                // first column of the wave inside the workgroup's range
    const int col0 = colw + lcol0;
    const int LDW = cw + 4;                        // +4 floats: consecutive rows start 16 B apart in the bank row
    // LDS: [2][cin] folded affine | G slabs of (T + 1) x cw | [2][G][cw] doubles (epilogue reduction)
    float *s_xf = smem;
    float *s_slab = smem + (XF ? 2 * cin : 0);
    const long slab_words = (long)(T + 1) * LDW;
    double *s_p = (double *)(s_slab + (((long)G * slab_words + 1) & ~1L));

    const long n_out = g.n_dev ? (long)*g.n_dev : g.n_cap;
    const long row0 = (long)tile * T;
    const int rows_here = (int)(n_out - row0 < (long)T ? n_out - row0 : (long)T);
    if (rows_here <= 0) return;   // workgroup-uniform

    const bool ident = g.pairs == nullptr;   // 1x1 convolution on the identity table: blocks of 16 consecutive rows
    const int *hdr = ident ? nullptr : g.pairs + (long)tile * urn_pairs_words(K, T);
    const int nblk = ident ? (rows_here + 15) >> 4 : hdr[0];
    // ADDRESSING of the block loop.  With 64-bit pointers every gathered row cost two v_mad_u64_u32, a v_lshl_add_u64 and moves,
    // every pair word and weight block the same again: ~130 vector instructions per block of 8 MFMAs at 32 -> 32, the loop was
    // bound by VALU issue (quarter-rate 64-bit multiplies), not by its loads.  Buffer loads instead: one descriptor per operand
    // array, the wave-uniform part of an address in the scalar offset (block number, offset, chunk: SALU), the lane's part a
    // 32-bit VGPR -- one v_mad_u32_u24 per gathered row, nothing per pair word or weight block.  (Anything derived from
    // threadIdx is divergent to the compiler, also the wave number: made scalar with readfirstlane once.)
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)g.x, 0, -1, 0x00020000);
    // (the list's descriptor is bounded: the strip variant fills whole 16-block pieces, which may reach past the last tile's record -- zeros)
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc((void *)g.pairs, 0, ident ? 0 : (int)((long)gridDim.x * urn_pairs_words(K, T) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)g.wfrag, 0, -1, 0x00020000);
    const int pw_t = (int)((long)tile * urn_pairs_words(K, T)) + URN_PAIRS_HDR;   // word index of blk_t[0] / blk_p[0] in g.pairs
    const int pw_p = pw_t + (int)urn_pairs_tpad(K, T);
    const int ldxb = (int)g.ldx * 4;
    // strip variant: the pair words and table rows of the TILE's first blocks are requested HERE -- wave w takes blocks
    // [16 w, 16 w + 16), whatever the tile's block count turns out to be: the request does not wait for the header, so the
    // head of the launch is two dependent round trips (header + pair words, then rows), not three.  They are parked in ONE
    // strip per workgroup behind the statistics prologue and the slab zero-fill; tiles with more blocks than the waves
    // cover fetch the rest once the header is there.
    typedef int urn_i32x4 __attribute__((ext_vector_type(4)));
    urn_i32x4 sw0 = (urn_i32x4){0, 0, 0, 0};
    int stw = 0;
    if constexpr (DEEP == 2) {
        if (wave * 16 < g.p_strip) {
            sw0 = __builtin_bit_cast(urn_i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, lane * 16, (pw_p + wave * 256) * 4, 0));
            if (lane < 16) stw = __builtin_amdgcn_raw_buffer_load_b32(rs_p, lane * 4, (pw_t + wave * 16) * 4, 0);
        }
    }

    const int b0 = __builtin_amdgcn_readfirstlane(nblk * gi / G);
    // (g.dbg: timing-only ablations of whole phases, tools/bench_pairs.py abl: 32 = no block loop, 64 = return before the epilogue,
    //  128 = one block per wave)
    const int b1 = __builtin_amdgcn_readfirstlane(URN_DBG(g, 32) ? b0 : (URN_DBG(g, 128) ? min(b0 + 1, nblk * (gi + 1) / G) : nblk * (gi + 1) / G));   // nblk <= 27 * 8, G <= 16

    URN_STAMP(7);   // (the header has arrived: b0 / b1 are scalars)
    // GATHER MAPPING.  The MFMA wants lane (r, q) to hold channels 4q..4q+3 of pair r -- but a wave-load in that shape costs the
    // CU ~66 cycles whatever the cache level (tools/ubench/gather_map.hip): the texture addresser takes the lanes four at a time,
    // and four consecutive lanes are four different rows = four lines.  Loaded row-major instead -- lane l reads piece l & 3 of
    // pair l >> 2, a quad of lanes = 64 contiguous bytes -- the same 16 rows x 64 bytes cost 22 cycles; the fragment shape is
    // then restored across the lanes (ds_bpermute_b32: lane (r, q) takes the registers of lane 4 r + q).
    const int gp = lane >> 2, gq = lane & 3;               // gather pair / 16-byte piece of this lane
    const int bp_addr = 4 * (4 * r + q);                   // byte address of the source lane for ds_bpermute
    // per block ONE vector load (the pair word of lane & 15) and one scalar load (the block's table row): the gather row of
    // lane l -- the word of pair l >> 2 -- is taken from lane l >> 2 with ds_bpermute when the rows are requested (the loop is
    // bound by the vector-memory instructions a CU can issue; this was three per block)
    auto load_idx = [&](int b, int &pv, int &tv) {
        if (ident) {
            const int lr = 16 * b + r;
            pv = lr < rows_here ? (((int)row0 + lr) | (lr << 24)) : (T << 24);
            tv = 0;
            return;
        }
        const int so = (pw_p + b * 16) * 4;   // scalar (b is)
        pv = __builtin_amdgcn_raw_buffer_load_b32(rs_p, r * 4, so, 0);
        tv = URN_DBG(g, 512) ? 0 : g.pairs[pw_t + b];   // wave-uniform address: s_load
    };
    auto gather_word = [&](int pv) { return __builtin_amdgcn_ds_bpermute(4 * gp, pv); };
    // the first pair words of the wave's share are requested BEFORE the statistics of the folded BatchNorm are fetched and
    // finalized below: two round trips side by side instead of one after the other at the head of every folding launch
    int pv_c = 0, tv_c = 0, pv_n = 0, tv_n = 0, pv_nn = 0, tv_nn = 0;
    if constexpr (DEEP == 0) {
        if (b0 < b1) {
            load_idx(b0, pv_c, tv_c);
            load_idx(b0 + 1 < b1 ? b0 + 1 : b1 - 1, pv_n, tv_n);
            load_idx(b0 + 2 < b1 ? b0 + 2 : b1 - 1, pv_nn, tv_nn);
        }
    }
    if constexpr (XF != 0) {
        if (g.xs_sums[0] != nullptr) {
            // statistics of the input rows accumulated by the producers: same arithmetic as k_bn_finalize_fwd_f
            const bool keep = tile == 0 && cgroup == 0;
            const double inv_n = g.xs_n > 0 ? 1.0 / (double)g.xs_n : 0.0;
            for (int e = tid; e < cin; e += nthreads) {
                const int sl = e >= g.xs_split ? 1 : 0;
                const int ch = sl ? e - g.xs_split : e;
                const double *p = g.xs_sums[sl] + ch;
                const int ld = g.xs_ld[sl];
                const float gam = g.xs_gamma[e], bet = g.xs_beta[e];   // requested together with the slab rows
                double v0, v1;
                urn_slab_sum2(p, ld, g.xs_slots, v0, v1);
                const double mu = v0 * inv_n;
                double var = v1 * inv_n - mu * mu;
                if (var < 0.0) var = 0.0;
                const double is = rsqrt(var + g.fin_eps);
                const float sc = gam * (float)is;
                const float sh = fmaf(-(float)mu, sc, bet);
                s_xf[e] = sc; s_xf[cin + e] = sh;
                if (keep) {
                    g.xs_mean[e] = (float)mu; g.xs_invstd[e] = (float)is; g.xs_scale[e] = sc; g.xs_shift[e] = sh;
                    if (g.xs_rm) g.xs_rm[e] = (float)(g.fin_momentum * g.xs_rm[e] + (1.0 - g.fin_momentum) * mu);
                    if (g.xs_rv) g.xs_rv[e] = (float)(g.fin_momentum * g.xs_rv[e] + (1.0 - g.fin_momentum) * var);
                }
            }
        } else {
            for (int e = tid; e < cin; e += nthreads) { s_xf[e] = g.xf_scale[e]; s_xf[cin + e] = g.xf_shift[e]; }
        }
    }
    // this wave's slab: rows 0..T (T = trash) of slab gi, columns [col0, col0 + 16 NC)
    float *slab = s_slab + (long)gi * slab_words + lcol0;
    for (int row = q; row <= T; row += 4)
#pragma unroll
        for (int c = 0; c < NC; ++c) slab[(long)row * LDW + 16 * c + r] = 0.f;
    if constexpr (DEEP == 2) {
        // (16-byte aligned: the strip starts at the next multiple of four words behind the epilogue's doubles: p_strip blocks of
        //  16 words, then their p_strip table rows; p_strip a multiple of 16, at least the longest list of a tile)
        int *strip_e = (int *)smem + (((long)((int *)(s_p + 2 * (long)G * cw) - (int *)smem) + 3) & ~3L);
        int *strip_te = strip_e + (long)g.p_strip * 16;
        const int nw = nthreads >> 6;
        if (wave * 16 < g.p_strip) {
            *(urn_i32x4 *)(strip_e + wave * 256 + lane * 4) = sw0;
            if (lane < 16) strip_te[wave * 16 + lane] = stw;
        }
        for (int c = 16 * nw; c < nblk; c += 16 * nw) {      // workgroup-uniform: lists longer than the waves covered
            const int ch = c / 16 + wave;
            if (ch * 16 < nblk && ch * 16 < g.p_strip) {
                const urn_i32x4 w = __builtin_bit_cast(urn_i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, lane * 16, (pw_p + ch * 256) * 4, 0));
                int tw = 0;
                if (lane < 16) tw = __builtin_amdgcn_raw_buffer_load_b32(rs_p, lane * 4, (pw_t + ch * 16) * 4, 0);
                *(urn_i32x4 *)(strip_e + ch * 256 + lane * 4) = w;
                if (lane < 16) strip_te[ch * 16 + lane] = tw;
            }
        }
        __syncthreads();
    }
    if constexpr (XF != 0 && DEEP != 2) __syncthreads();
    URN_STAMP(1);

    // pair word of block b for lane (r, q): pair r of the block = input row | row in the tile << 24.  With the weights as
    // the MFMA's A operand and the gathered rows as its B operand, lane (r, q) ends up with output columns 4q..4q+3 of
    // pair r: one 16-byte read-modify-write of the slab per column block.  Within one block the pairs have distinct
    // tile rows, so the plain (non-atomic) update is race-free; padding pairs all hit the trash row, which nobody reads.
    typedef typename PairsW<PREC>::type wfrag_t;
    f32x4 a_nxt[KC];                 // rows as loaded (row-major lane mapping)
    wfrag_t a_cur[KC];               // the MFMA's B operand: folded BatchNorm applied, rounded (PREC != 0), in fragment shape
    wfrag_t w_cur[KC][NC];
    // (g.dbg 256: every pair gathers row 0 -- the loads stay, their cache lines collapse to one; 512: the weight block of offset 0
    //  for every block -- no reloads.  Timing only.)
    const int row_mask = URN_DBG(g, 256) ? 0 : 0xFFFFFF;
    auto load_a = [&](f32x4 (&a)[KC], int pl, int ch) {
        const int vo = (int)__umul24((unsigned)(pl & row_mask), (unsigned)ldxb) + 16 * gq;
        const int so = ch * (64 * KC);
#pragma unroll
        for (int j = 0; j < KC; ++j)
            a[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo + 64 * j, so, 0));
    };
    // loaded rows of channel chunk ch -> operand registers: relu(x * scale + shift) where the input BatchNorm is folded (this
    // lane holds channels 4 gq..+3 of the group), rounding, then the transpose across the lanes
    auto ready = [&](const f32x4 (&raw)[KC], wfrag_t (&dst)[KC], int ch) {
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            f32x4 v = raw[j];
            if constexpr (XF != 0) {
                const f32x4 sc = *(const f32x4 *)(s_xf + ch * (16 * KC) + 16 * j + 4 * gq);
                const f32x4 sh = *(const f32x4 *)(s_xf + cin + ch * (16 * KC) + 16 * j + 4 * gq);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaf(v[k], sc[k], sh[k]), 0.f);
            }
            if constexpr (PREC == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float f = v[k];   // (__builtin_bit_cast applied to a vector ELEMENT takes element 0 with this hipcc)
                    dst[j][k] = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(f)));
                }
            } else {
                const uint2 h = urn_round16x4<PREC>(v);
                uint2 t;
                t.x = (unsigned)__builtin_amdgcn_ds_bpermute(bp_addr, (int)h.x);
                t.y = (unsigned)__builtin_amdgcn_ds_bpermute(bp_addr, (int)h.y);
                dst[j] = __builtin_bit_cast(urn_s16x4, t);
            }
        }
    };
    // weight block of offset t, columns of this wave, channel chunk ch: from the fragment-ordered copy when the call has one
    // (one contiguous kilobyte per 16 x 16 block: 8 cache lines; the rows of wt are 16 half-used lines per block, and the
    // kernel is bound by the lines a CU can address per cycle)
    const int kbn = cin / 16;
    const int cb0 = __builtin_amdgcn_readfirstlane(col0 / 16);   // first column block of the wave
    auto load_w = [&](wfrag_t (&w)[KC][NC], int t, int ch) {
        const int o = g.flip ? (K - 1 - t) : t;
        if constexpr (PREC != 0) {
            // 16-bit fragments (the launcher guarantees g.wfrag_prec == PREC): 8 bytes per lane, 512 per block
            const int so = ((o * (cout / 16) + cb0) * kbn + ch * KC) * 512;
            if constexpr (KC % 2 == 0) {
                // paired fragments (urn_frag16_slot: kbn = KC * nch is even): one 16-byte load = groups j and j + 1
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int j = 0; j < KC; j += 2) {
                        typedef short s16x8 __attribute__((ext_vector_type(8)));
                        const s16x8 v = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, so + (c * kbn + j) * 512, 0));
                        w[j][c] = __builtin_shufflevector(v, v, 0, 1, 2, 3);
                        w[j + 1][c] = __builtin_shufflevector(v, v, 4, 5, 6, 7);
                    }
                return;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int j = 0; j < KC; ++j)
                    w[j][c] = __builtin_bit_cast(urn_s16x4, __builtin_amdgcn_raw_buffer_load_b64(rs_w, lane * 8, so + (c * kbn + j) * 512, 0));
            return;
        } else
        if (DEEP != 0 || g.wfrag) {   // (the strip variant is launched with fragments only: no branch around these requests)
            const int so = ((o * (cout / 16) + cb0) * kbn + ch * KC) * 1024;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int j = 0; j < KC; ++j)
                    w[j][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, so + (c * kbn + j) * 1024, 0));
            return;
        }
        if constexpr (PREC == 0 && DEEP == 0) {
            const float *src = g.wt + ((long)o * cout + col0 + r) * cin + ch * (16 * KC) + 4 * q;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int j = 0; j < KC; ++j) w[j][c] = *(const f32x4 *)(src + (long)(16 * c) * cin + 16 * j);
        }
    };

    if constexpr (DEEP == 2) {
        // STRIP variant (one channel chunk: cin = 16 KC; launcher: bit KC of `pairs_v3`).  What a CU's vector-memory path can
        // take is instructions, not bytes (any wave-wide load holds the address unit >= 16 cycles): the loop above issues
        // three dword loads per block for the pair words and the table row next to KC row loads and KC * NC weight loads.
        // Here the pair words and table rows of the TILE are copied once into one LDS strip per workgroup with 16-byte loads
        // (one per wave and 16 blocks, requested at the kernel's head without waiting for the header, see above); per block
        // the loop reads them back (ds_read_b32), so its vector-memory instructions are the gathered rows and the weight
        // blocks only.  The weight block of the next block, when its
        // offset differs, is requested together with the next block's rows BEFORE this block's MFMAs (second register set,
        // copied over behind the MFMAs; KC <= 4) -- one round trip per block where the loop above has two dependent ones
        // (weights at the head, rows at the tail).  The old slab values are the MFMA's C operand: no zero-fill, no adds.
        const int *strip = (const int *)smem + (((long)((int *)(s_p + 2 * (long)G * cw) - (int *)smem) + 3) & ~3L) + (long)b0 * 16;   // the wave's share of the workgroup's strip
        const int *strip_t = (const int *)smem + (((long)((int *)(s_p + 2 * (long)G * cw) - (int *)smem) + 3) & ~3L) + (long)g.p_strip * 16 + b0;
        const int nb = b1 - b0;
        URN_STAMP(5);
        if (nb > 0) {
            constexpr bool W2 = KC * NC <= 4;     // second weight register set (16 KC NC bytes per lane)
            constexpr bool XREG = XF != 0 && KC <= 3;
            f32x4 xsc[XREG ? KC : 1], xsh[XREG ? KC : 1];
            if constexpr (XREG) {
#pragma unroll
                for (int j = 0; j < KC; ++j) { xsc[j] = *(const f32x4 *)(s_xf + 16 * j + 4 * gq); xsh[j] = *(const f32x4 *)(s_xf + cin + 16 * j + 4 * gq); }
            }
            auto ready2 = [&](const f32x4 (&raw)[KC], wfrag_t (&dst)[KC]) {
#pragma unroll
                for (int j = 0; j < KC; ++j) {
                    f32x4 v = raw[j];
                    if constexpr (XF != 0) {
                        f32x4 sc, sh;
                        if constexpr (XREG) { sc = xsc[j]; sh = xsh[j]; }
                        else { sc = *(const f32x4 *)(s_xf + 16 * j + 4 * gq); sh = *(const f32x4 *)(s_xf + cin + 16 * j + 4 * gq); }
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaf(v[k], sc[k], sh[k]), 0.f);
                    }
                    if constexpr (PREC == 0) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float f = v[k];
                            dst[j][k] = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(f)));
                        }
                    } else {
                        const uint2 h = urn_round16x4<PREC>(v);
                        uint2 t;
                        t.x = (unsigned)__builtin_amdgcn_ds_bpermute(bp_addr, (int)h.x);
                        t.y = (unsigned)__builtin_amdgcn_ds_bpermute(bp_addr, (int)h.y);
                        dst[j] = __builtin_bit_cast(urn_s16x4, t);
                    }
                }
            };
            wfrag_t w_nxt[W2 ? KC : 1][W2 ? NC : 1];
            int t_c = __builtin_amdgcn_readfirstlane(strip_t[0]);
            load_w(w_cur, t_c, 0);
            load_a(a_nxt, strip[gp], 0);
            int pv_c2 = strip[r];
            const int i1 = nb > 1 ? 1 : 0;
            int pv_n2 = strip[i1 * 16 + r], pl_n2 = strip[i1 * 16 + gp];
            int t_n = URN_DBG(g, 512) ? t_c : __builtin_amdgcn_readfirstlane(strip_t[i1]);
            ready2(a_nxt, a_cur);
            for (int i = 0; i < nb; ++i) {
                const int i2 = i + 2 < nb ? i + 2 : nb - 1;
                float *dptr = slab + (long)((unsigned)pv_c2 >> 24) * LDW + 4 * q;
                f32x4 acc[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = *(const f32x4 *)(dptr + 16 * c);
                const int pv_nn = strip[i2 * 16 + r], pl_nn = strip[i2 * 16 + gp], tv_nn = strip_t[i2];
                const bool more = i + 1 < nb;                 // wave-uniform
                const bool chg = more && t_n != t_c;
                if constexpr (W2) { if (chg) load_w(w_nxt, t_n, 0); }
                if (more) load_a(a_nxt, pl_n2, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (PREC == 0) {
#pragma unroll
                    for (int j = 0; j < KC; ++j)
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                            for (int c = 0; c < NC; ++c) acc[c] = MFMA16(w_cur[j][c][tt], a_cur[j][tt], acc[c]);
                } else if constexpr (KC % 2 == 0) {
#pragma unroll
                    for (int j = 0; j < KC; j += 2)
#pragma unroll
                        for (int c = 0; c < NC; ++c) acc[c] = pairs_mfma32<PREC>(w_cur[j][c], w_cur[j + 1][c], a_cur[j], a_cur[j + 1], acc[c]);
                } else {
#pragma unroll
                    for (int j = 0; j < KC; ++j)
#pragma unroll
                        for (int c = 0; c < NC; ++c) acc[c] = pairs_mfma16<PREC>(w_cur[j][c], a_cur[j], acc[c]);
                }
                if constexpr (!W2) { if (chg) load_w(w_cur, t_n, 0); }      // (issued behind the MFMAs that read w_cur)
#pragma unroll
                for (int c = 0; c < NC; ++c) *(f32x4 *)(dptr + 16 * c) = acc[c];
                if (more) ready2(a_nxt, a_cur);
                if constexpr (W2) {
                    if (chg) {
#pragma unroll
                        for (int j = 0; j < KC; ++j)
#pragma unroll
                            for (int c = 0; c < NC; ++c) w_cur[j][c] = w_nxt[j][c];
                    }
                }
                pv_c2 = pv_n2; pv_n2 = pv_nn; pl_n2 = pl_nn;
                t_c = t_n; t_n = URN_DBG(g, 512) ? t_c : __builtin_amdgcn_readfirstlane(tv_nn);
            }
        }
    }
    if constexpr (DEEP == 0)
    // Steps s = (block, channel chunk).  The gathered rows of step s + 1 are requested before the MFMAs of step s and the
    // pair words three blocks ahead.  The weight fragments stay in registers while the offset (and chunk) does not
    // change; when it does they are requested BEFORE the next step's rows, so that waiting for them (s_waitcnt vmcnt(N)
    // counts in issue order) leaves the younger row loads in flight.
    if (b0 < b1) {
        int t_c = __builtin_amdgcn_readfirstlane(tv_c);
        int pl_c = gather_word(pv_c);
        load_a(a_nxt, pl_c, 0);
        ready(a_nxt, a_cur, 0);
        int w_key = -1;
        for (int b = b0; b < b1; ++b) {
            int pv_n3, tv_n3;
            load_idx(b + 3 < b1 ? b + 3 : b1 - 1, pv_n3, tv_n3);
            const int t_n = __builtin_amdgcn_readfirstlane(tv_n);
            const int pl_n = gather_word(pv_n);
            // old slab values of this block's rows: requested now, needed after the MFMAs
            float *dptr = slab + (long)((unsigned)pv_c >> 24) * LDW + 4 * q;
            f32x4 old[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) old[c] = *(const f32x4 *)(dptr + 16 * c);
            f32x4 acc[NC], acc2[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) { acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            for (int ch = 0; ch < nch; ++ch) {
                const bool last = ch + 1 == nch;
                const int key = t_c * nch + ch;
                if (key != w_key) { load_w(w_cur, t_c, ch); w_key = key; }   // wave-uniform
                load_a(a_nxt, last ? pl_n : pl_c, last ? 0 : ch + 1);
                __builtin_amdgcn_sched_barrier(0);   // keep the requests in front of the MFMAs (the scheduler sinks them otherwise)
                // D[column 4q+i of the block][pair r] = sum_k W[k][column] x[pair][k]: weights are the A operand.  Two
                // accumulators alternate so that consecutive MFMAs are independent (40-cycle dependent latency vs 32 issue)
                if constexpr (PREC == 0) {
#pragma unroll
                    for (int j = 0; j < KC; ++j)
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                            for (int c = 0; c < NC; ++c) {
                                if (tt & 1) acc2[c] = MFMA16(w_cur[j][c][tt], a_cur[j][tt], acc2[c]);
                                else acc[c] = MFMA16(w_cur[j][c][tt], a_cur[j][tt], acc[c]);
                            }
                } else if constexpr (KC % 2 == 0) {
#pragma unroll
                    for (int j = 0; j < KC; j += 2)
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            if (j & 2) acc2[c] = pairs_mfma32<PREC>(w_cur[j][c], w_cur[j + 1][c], a_cur[j], a_cur[j + 1], acc2[c]);
                            else acc[c] = pairs_mfma32<PREC>(w_cur[j][c], w_cur[j + 1][c], a_cur[j], a_cur[j + 1], acc[c]);
                        }
                } else {
#pragma unroll
                    for (int j = 0; j < KC; ++j)
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            if (j & 1) acc2[c] = pairs_mfma16<PREC>(w_cur[j][c], a_cur[j], acc2[c]);
                            else acc[c] = pairs_mfma16<PREC>(w_cur[j][c], a_cur[j], acc[c]);
                        }
                }
                // the next step's rows (they have had the MFMAs to arrive): fold, round, transpose into the operand registers
                // (the weight block one step ahead as well, unconditionally into a second register set, was measured: 2.74 ->
                // 2.81 ms per cfg3 step, 12.4 -> 13.2 at the cfg5 shape in fp16 -- the extra requests cost more than the wait)
                ready(a_nxt, a_cur, last ? 0 : ch + 1);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) *(f32x4 *)(dptr + 16 * c) = old[c] + (acc[c] + acc2[c]);
            pv_c = pv_n; pv_n = pv_nn; pv_nn = pv_n3;
            pl_c = pl_n;
            t_c = t_n; tv_n = tv_nn; tv_nn = tv_n3;
        }
    }
    URN_STAMP(2);
    if (URN_DBG(g, 64)) return;

    // epilogue: wave (cg, gi) finishes columns [col0, col0 + 16 NC) of the row groups gi, gi + G, ...  A lane owns FOUR
    // consecutive columns of a row (16-byte slab reads, residual / BatchNorm-input loads and stores; one float per lane made
    // the epilogue 2-3 us of a 12-25 us launch): 4 NC lanes per row, 64 / (4 NC) rows per wave instruction.
    constexpr int LPR = 4 * NC, RPI = 64 / LPR;
    const int c4 = lane % LPR, rl = lane / LPR;
    const int lcol = lcol0 + 4 * c4, gcol = col0 + 4 * c4;
    double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
    f32x4 esc = (f32x4){0.f, 0.f, 0.f, 0.f}, esh = esc, emu = esc, eis = esc;
    if (g.epi == 2) {
        esc = *(const f32x4 *)(g.e_scale + gcol); esh = *(const f32x4 *)(g.e_shift + gcol);
        emu = *(const f32x4 *)(g.e_mean + gcol); eis = *(const f32x4 *)(g.e_invstd + gcol);
    }
    // two row groups per pass: their residual / BatchNorm-input elements are requested together before the first is used
    // (four per pass pushed every instantiation to the 128-VGPR cap: 2.82 -> 3.14 ms per step).  The FIRST pass' requests --
    // with 64-row tiles the only pass -- go out in front of the barrier that ends the block loop: their round trip runs
    // beside the wait for the workgroup's slowest wave (strip variant: it has the registers; the others load behind it)
    f32x4 rv[2], xg[2];
    auto request = [&](int lr0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int lr = lr0 + G * RPI * u;
            const long row = row0 + lr;
            rv[u] = (g.res && lr < rows_here) ? *(const f32x4 *)(g.res + row * cout + gcol) : (f32x4){0.f, 0.f, 0.f, 0.f};
            xg[u] = (g.epi == 2 && lr < rows_here) ? *(const f32x4 *)(g.e_x + row * cout + gcol) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    if constexpr (DEEP == 2) request(gi * RPI + rl);
    __syncthreads();
    URN_STAMP(3);
#pragma unroll 1
    for (int lr0 = gi * RPI + rl; lr0 < rows_here; lr0 += 2 * G * RPI) {
        if (DEEP != 2 || lr0 != gi * RPI + rl) request(lr0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int lr = lr0 + G * RPI * u;
            if (lr >= rows_here) break;
            const long row = row0 + lr;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
            for (int k = 0; k < G; ++k) v += *(const f32x4 *)(s_slab + (long)k * slab_words + (long)lr * LDW + lcol);
            v += rv[u];
            if (g.epi == 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { s0[k] += (double)v[k]; s1[k] += (double)v[k] * (double)v[k]; }
            } else if (g.epi == 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xv = xg[u][k];
                    if (!(fmaf(xv, esc[k], esh[k]) > 0.f)) v[k] = 0.f;
                    const double xh = ((double)xv - (double)emu[k]) * (double)eis[k];
                    s0[k] += (double)v[k];
                    s1[k] += (double)v[k] * xh;
                }
            }
            *(f32x4 *)(g.y + row * g.ldy + gcol) = v;
        }
    }
    URN_STAMP(4);
    if (g.epi == 0) return;
    // lanes with the same four columns are LPR apart
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1)
#pragma unroll
        for (int k = 0; k < 4; ++k) { s0[k] += __shfl_xor(s0[k], m); s1[k] += __shfl_xor(s1[k], m); }
    if (lane < LPR) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s_p[(long)(0 * G + gi) * cw + lcol + k] = s0[k];
            s_p[(long)(1 * G + gi) * cw + lcol + k] = s1[k];
        }
    }
    __syncthreads();
    for (int lc = tid; lc < cw; lc += nthreads) {
        const int col = colw + lc;
        double v0 = 0.0, v1 = 0.0;
        for (int k = 0; k < G; ++k) { v0 += s_p[(long)(0 * G + k) * cw + lc]; v1 += s_p[(long)(1 * G + k) * cw + lc]; }
        if (g.part_slots > 0) {   // accumulate: hardware fp64 add at the memory side, no return value
            const long slot = tile % (unsigned)g.part_slots;
            unsafeAtomicAdd(&g.part[(slot * 2 + 0) * cout + col], v0);
            unsafeAtomicAdd(&g.part[(slot * 2 + 1) * cout + col], v1);
        } else {
            g.part[((long)tile * 2 + 0) * cout + col] = v0;
            g.part[((long)tile * 2 + 1) * cout + col] = v1;
        }
    }
    URN_STAMP(6);
}


// ---- 16-bit variants, one translation unit per (precision, column blocks per wave) group
bool urn_pairs16_p1(int kc, int nc, const GArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);   // bf16, NC 1 | 2
bool urn_pairs16_p2(int kc, int nc, const GArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);   // bf16, NC 4
bool urn_pairs16_p3(int kc, int nc, const GArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);   // fp16, NC 1 | 2
bool urn_pairs16_p4(int kc, int nc, const GArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);   // fp16, NC 4
#if URN_PAIRS_PART != 0
template <typename F>
static void pairs_big_lds(F f, size_t lds)
{
    if (lds > 65536) (void)hipFuncSetAttribute((const void *)f, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
}
template <int KC, int NC, int PREC>
static void launch_pairs16(const GArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st)
{
    const bool xf = a.xf_scale != nullptr || a.xs_sums[0] != nullptr;
    if (lds > 65536) {
        pairs_big_lds(k_gconv_pairs<KC, NC, 1, 0, PREC>, lds); pairs_big_lds(k_gconv_pairs<KC, NC, 0, 0, PREC>, lds);
        if constexpr (NC <= 2) { pairs_big_lds(k_gconv_pairs<KC, NC, 1, 2, PREC>, lds); pairs_big_lds(k_gconv_pairs<KC, NC, 0, 2, PREC>, lds); }
    }
    if constexpr (NC <= 2) {
        if (a.p_deep == 2) {
            if (xf) hipLaunchKernelGGL((k_gconv_pairs<KC, NC, 1, 2, PREC>), grid, block, lds, st, a);
            else hipLaunchKernelGGL((k_gconv_pairs<KC, NC, 0, 2, PREC>), grid, block, lds, st, a);
            return;
        }
    }
    if (xf) hipLaunchKernelGGL((k_gconv_pairs<KC, NC, 1, 0, PREC>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((k_gconv_pairs<KC, NC, 0, 0, PREC>), grid, block, lds, st, a);
}
#define URN_PL(KCv, NCv) if (kc == KCv && nc == NCv) { launch_pairs16<KCv, NCv, URN_P16_PREC>(a, grid, block, lds, st); return true; }
#if URN_PAIRS_PART == 1 || URN_PAIRS_PART == 3
#define URN_P16_PREC (URN_PAIRS_PART == 1 ? 1 : 2)
#if URN_PAIRS_PART == 1
bool urn_pairs16_p1(int kc, int nc, const GArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st)
#else
bool urn_pairs16_p3(int kc, int nc, const GArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st)
#endif
{
    URN_PL(1, 1) URN_PL(2, 1) URN_PL(3, 1) URN_PL(4, 1) URN_PL(5, 1) URN_PL(6, 1) URN_PL(8, 1)
    URN_PL(1, 2) URN_PL(2, 2) URN_PL(3, 2) URN_PL(4, 2) URN_PL(5, 2) URN_PL(6, 2) URN_PL(8, 2)
    return false;
}
#else
#define URN_P16_PREC (URN_PAIRS_PART == 2 ? 1 : 2)
#if URN_PAIRS_PART == 2
bool urn_pairs16_p2(int kc, int nc, const GArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st)
#else
bool urn_pairs16_p4(int kc, int nc, const GArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st)
#endif
{
    URN_PL(1, 4) URN_PL(2, 4) URN_PL(3, 4) URN_PL(4, 4) URN_PL(5, 4) URN_PL(6, 4) URN_PL(8, 4)
    return false;
}
#endif
#undef URN_PL
#endif   // URN_PAIRS_PART != 0

#if URN_PAIRS_PART == 0
int g_pairs_waves = 2560;    // a launch is split G ways until it has about this many waves (urn_set_option "pairs_waves"): alone more is faster (4096+), beside the weight gradients of the training step fewer are (A/B in one process, tools/ab_options.py: 3.21 ms per step at 4096, 3.12 at 3072, 3.06 at 2560, 3.09 at 2304, 3.12 at 2048, 3.19 at 1536)
int g_pairs_wgs16 = 256;     // 16-bit operands: four column blocks per wave only while the launch keeps this many workgroups ("pairs_wgs16")
int g_pairs_waves_fwd = 0;   // the same for launches that do not run beside the weight gradients (forward: epilogue != 2), 0 = g_pairs_waves ("pairs_waves_fwd")
int g_pairs_nc = 0;          // force the column blocks per wave (urn_set_option "pairs_nc"), 0 = automatic
int g_pairs_split = 0;
int g_pairs_split_kc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // force G for inputs of 16 KC channels (urn_set_option "pairs_split_kc<KC>"), 0 = automatic
int g_pairs_wgs = 512;       // a workgroup takes several column groups only while the launch keeps this many workgroups ("pairs_wgs")
int g_pairs_v3 = 0x17E;       // bit KC set: the STRIP variant (pair words of a wave's share in LDS, weight block of the next offset requested with the next rows) for one-chunk inputs of 16 KC channels (urn_set_option "pairs_v3")
int g_pairs_lds_cap16 = 0;   // 16-bit operands: LDS bytes a workgroup may take (urn_set_option "pairs_lds_cap16", 0 = 64 KB; a workgroup may declare up to 160 KB): more column groups / list shares per workgroup at the wide layers.  Measured at the cfg5 shape in fp16: 11.69 -> 11.48 ms per step at 96 KB, the same at 160 KB; 128-row tiles with it 13.3 ms.  Opt-in: the other launch shapes move the small fp16 network's gradient noise (ReLU flips) to the edge of its test bound
int g_pairs_cbg = 0;          // most column groups per workgroup (urn_set_option "pairs_cbg"), 0 = as many as fit       // force G (urn_set_option "pairs_split"), 0 = automatic

template <int KC, int NC>
static void launch_pairs2(const GArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st)
{
    const bool xf = a.xf_scale != nullptr || a.xs_sums[0] != nullptr;
    if (a.p_deep == 2) {
        if (xf) hipLaunchKernelGGL((k_gconv_pairs<KC, NC, 1, 2>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((k_gconv_pairs<KC, NC, 0, 2>), grid, block, lds, st, a);
        return;
    }
    if (xf) hipLaunchKernelGGL((k_gconv_pairs<KC, NC, 1, 0>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((k_gconv_pairs<KC, NC, 0, 0>), grid, block, lds, st, a);
}

// returns the number of partial rows (tiles), 0 when the shape has no instantiation
int urn_gconv_pairs_launch(GArgs a, long n_out, hipStream_t st)
{
    const int T = a.p_tile;
    if ((T != 32 && T != 64 && T != 128) || a.cin % 16 || a.cout % 16 || a.K > 27) return 0;
    if (a.prec != 0 && (a.wfrag == nullptr || a.wfrag_prec != a.prec)) return 0;   // the 16-bit variants read 16-bit fragments only
    if (a.prec == 0 && a.wfrag_prec != 0) a.wfrag = nullptr;                        // 16-bit fragments are of no use to the fp32 variants: rows of wt
    if (a.ldy % 4 || ((uintptr_t)a.y & 15) || (a.res && ((uintptr_t)a.res & 15)) || (a.e_x && ((uintptr_t)a.e_x & 15))) return 0;   // 16-byte epilogue accesses
    const int ks = a.cin / 16, nblk = a.cout / 16;
    int kc = 1;
    for (int d : {8, 6, 5, 4, 3, 2})
        if (ks % d == 0 && (a.prec == 0 || (ks & 1) || !(d & 1))) { kc = d; break; }   // 16-bit fragments of an even group count are PAIRED (urn_frag16_slot): even chunks
    const long ntiles = (n_out + T - 1) / T;
    if ((double)ntiles * (double)urn_pairs_words(a.K, T) * 4.0 >= 2147483648.0) return 0;   // 32-bit offsets into the lists (rows / weights: the dispatcher's off32_ok)
    // two column blocks per wave halve the gathers (every wave of a tile gathers the same rows) when the launch still has
    // enough waves; the weight fragments of an offset must stay in registers (KC * NC * 4 <= 48)
    int nc = 1;
    if (nblk % 2 == 0 && kc <= 4 && ntiles * (nblk / 2) >= 2 * g_pairs_wgs) nc = 2;
    // 16-bit operands: one MFMA per 16-channel group, the launch is bound by its gathers -- two column blocks per wave
    // wherever the fragments fit (768^3 / 200k voxels / uf 32 / uns 7 in fp16: 21.1 -> 19.0 ms per step)
    // with 16-bit fragments (8 bytes per lane) the weight blocks of any chunk fit: four column blocks while the launch keeps
    // g_pairs_wgs16 workgroups
    if (a.prec != 0) {
        nc = nblk % 2 == 0 ? 2 : 1;
        if (nblk % 4 == 0 && ntiles * (nblk / 4) >= g_pairs_wgs16) nc = 4;
    }
    if (g_pairs_nc == 1 || (g_pairs_nc == 2 && nblk % 2 == 0 && (kc <= 6 || a.prec != 0)) || (g_pairs_nc == 4 && nblk % 4 == 0 && a.prec != 0)) nc = g_pairs_nc;
    const int cbg_all = nblk / nc;
    const bool strip = ((g_pairs_v3 >> kc) & 1) && a.cin == 16 * kc && a.pairs != nullptr && a.wfrag != nullptr && nc <= 2;
    const int maxw = URN_PAIRS_REGS(kc, nc, a.prec) <= 64 ? 16 : 8;   // waves per workgroup (register budget, see __launch_bounds__)
    // Workgroups first: the deep levels have few tiles (103 of 64 rows at level 3 of cfg3), and one workgroup per tile left
    // most of the 256 CUs idle (measured 56 -> 15 us at level 4, 80 -> 80, with one column group per workgroup): give a
    // workgroup fewer column groups (its rows are then gathered by several workgroups -- L2 hits) until the launch has
    // g_pairs_wgs workgroups; then split the block lists G ways.
    int cbg = 1;
    for (int d = cbg_all < maxw ? cbg_all : maxw; d >= 1; --d)
        if (cbg_all % d == 0 && (d == 1 || ntiles * (cbg_all / d) >= g_pairs_wgs)) { cbg = d; break; }
    if (g_pairs_cbg > 0 && cbg_all % g_pairs_cbg == 0 && g_pairs_cbg <= maxw) cbg = g_pairs_cbg;
    int gy = cbg_all / cbg, cw = 16 * nc * cbg;
    const int maxb = (int)urn_pairs_maxb(a.K, T);
    auto strip_blocks = [&](int) { return (maxb + 15) & ~15; };   // ONE strip per workgroup: the longest list a tile can have, in whole 16-block fills
    auto lds_bytes = [&](int G) {
        size_t w = (size_t)2 * a.cin + (((size_t)G * (T + 1) * (cw + 4) + 1) & ~(size_t)1);
        return w * 4 + (size_t)2 * G * cw * 8 + (strip ? (size_t)strip_blocks(G) * 17 * 4 + 16 : 0);
    };
    const int lds_cap = (a.prec != 0 && g_pairs_lds_cap16 > 0) ? g_pairs_lds_cap16 : 65536;   // a workgroup may declare up to 160 KB (one resident workgroup per CU then)
    int G = 1;
    const int want_waves = (a.epi != 2 && g_pairs_waves_fwd > 0) ? g_pairs_waves_fwd : g_pairs_waves;
    for (;;) {
        G = 1;
        while (G < 8 && cbg * (G + 1) <= maxw && ntiles * cbg_all * G < want_waves && (G + 1) * 2 <= (a.K * (T / 16) + 1) && lds_bytes(G + 1) <= (size_t)lds_cap) ++G;
        if (g_pairs_split > 0 && cbg * g_pairs_split <= maxw && lds_bytes(g_pairs_split) <= (size_t)lds_cap) G = g_pairs_split;
        if (kc <= 8 && g_pairs_split_kc[kc] > 0 && cbg * g_pairs_split_kc[kc] <= maxw && lds_bytes(g_pairs_split_kc[kc]) <= (size_t)lds_cap) G = g_pairs_split_kc[kc];
        if (lds_bytes(G) <= (size_t)lds_cap || cbg == 1) break;
        // slabs + strips of this many column groups do not fit 64 KB of LDS: fewer column groups per workgroup
        do { --cbg; } while (cbg > 1 && cbg_all % cbg);
        gy = cbg_all / cbg; cw = 16 * nc * cbg;
    }
    if (lds_bytes(G) > (size_t)lds_cap) return 0;
    a.p_split = G; a.p_cw = cw; a.p_deep = strip ? 2 : 0; a.p_strip = strip ? strip_blocks(G) : 0;
    const dim3 grid((unsigned)ntiles, gy), block(64 * cbg * G);
    const size_t lds = lds_bytes(G);
#define URN_PL(KCv, NCv) if (kc == KCv && nc == NCv) { launch_pairs2<KCv, NCv>(a, grid, block, lds, st); return (int)ntiles; }
    if (a.prec != 0) {
        const bool ok = nc == 4 ? (a.prec == 1 ? urn_pairs16_p2 : urn_pairs16_p4)(kc, nc, a, grid, block, lds, st)
                                : (a.prec == 1 ? urn_pairs16_p1 : urn_pairs16_p3)(kc, nc, a, grid, block, lds, st);
        return ok ? (int)ntiles : 0;
    }
    URN_PL(1, 1) URN_PL(2, 1) URN_PL(3, 1) URN_PL(4, 1) URN_PL(5, 1) URN_PL(6, 1) URN_PL(8, 1)
    URN_PL(1, 2) URN_PL(2, 2) URN_PL(3, 2) URN_PL(4, 2) URN_PL(5, 2) URN_PL(6, 2)
#undef URN_PL
    return 0;
}
#endif   // URN_PAIRS_PART == 0
