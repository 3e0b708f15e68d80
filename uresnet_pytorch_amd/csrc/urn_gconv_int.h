// Internal launch record shared by the gather-conv kernels.
#pragma once
#include <hip/hip_runtime.h>

struct GArgs {
    const float *x, *wt;
    const float *wfrag;   // optional (pair-list kernel): the same weights in MFMA-fragment order, see urn_gconv_args.wt_frag
    int wfrag_prec;       // element type of wfrag: 0 fp32, 1 bf16, 2 fp16
    const int *tbl;
    long ld;
    int K, flip;
    const int *n_dev;
    long n_cap;
    int cout;
    int cin;   // total input channels (the tile kernel walks them in chunks of KS*16)
    const float *res;
    float *y;
    const float *xf_scale, *xf_shift;
    int epi;
    double *part;
    const float *e_x, *e_scale, *e_shift, *e_mean, *e_invstd;
    // in-kernel finalize of the epilogue partials (see urn_gconv_args in include/uresnet_hip.h)
    unsigned *sync_word;
    long fin_n;
    double fin_eps, fin_momentum;
    struct FinBN {
        const float *gamma, *beta;
        float *mean, *invstd, *scale, *shift, *running_mean, *running_var;
    } fin_bn[2];
    float *fin_dgamma, *fin_dbeta, *fin_coef0, *fin_coef1;
    // accumulated statistics (see urn_gconv_args)
    int part_slots, xs_slots, xs_split;
    int xs_ld[2];
    const double *xs_sums[2];
    long xs_n;
    const float *xs_gamma, *xs_beta;
    float *xs_mean, *xs_invstd, *xs_scale, *xs_shift, *xs_rm, *xs_rv;
    long ldx, ldy;   // row strides of x and y in floats (cin / cout unless a tensor is a column block of a wider matrix)
    int prec;  // MFMA operand precision: 0 fp32, 1 bf16, 2 fp16 (urn_gconv_args.precision / option "gconv_precision")
    // compacted rule lists (urn_gconv_pairs.hip): list of the table (NULL with p_tile != 0 = identity table of a 1x1 conv),
    // rows per tile (64 / 128), split of a tile's block list over waves, columns per workgroup
    const int *pairs;
    int p_tile, p_split, p_cw, p_deep;   // p_deep: loop variant, 0 = block loop with per-block index loads (several channel chunks, or no fragment-ordered weights), 2 = strip (pair words of the tile in LDS; default)
    int p_strip;   // variant 2: blocks per wave-private strip (a multiple of 16)
    long long *stamps;   // diagnostics (urn_set_option "gconv_stamp_ptr"): 8 s_memtime values per wave of the pair-list kernel
    int dbg;   // timing-only ablation mask (urn_set_option "gconv_dbg"): 1 no MFMA, 2 no A fetch, 4 no B fetch, 8 no barrier, 16 no offsets
};

#define URN_PAIRS_HDR 16   // int32 words in front of a tile's block list ([0] = number of blocks, bytes 4..31 = first block of every table row)
// layout of one tile of a pair list: header | table row of every block (padded to a multiple of 4 words) | 16 words per block
static __host__ __device__ inline long urn_pairs_maxb(int K, int T) { return (long)K * (T / 16); }
static __host__ __device__ inline long urn_pairs_tpad(int K, int T) { return (urn_pairs_maxb(K, T) + 3) & ~3L; }
static __host__ __device__ inline long urn_pairs_words(int K, int T) { return URN_PAIRS_HDR + urn_pairs_tpad(K, T) + urn_pairs_maxb(K, T) * 16; }
// compacted rule lists (urn_gconv_pairs.hip): returns the number of partial rows (tiles), 0 = no instantiation
int urn_gconv_pairs_launch(GArgs a, long n_out, hipStream_t st);
// 2-D workgroup tile (urn_gconv_tile.hip): returns the number of partial rows, 0 = no instantiation
int urn_gconv_tile_launch(const GArgs &a, int ks, long n_out, hipStream_t st);
