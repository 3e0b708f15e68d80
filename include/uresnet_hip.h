/*
 * uresnet_hip.h -- C ABI of liburesnet_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the U-ResNet sparse/dense 3-D convolution forward+backward
 * path of Temigo/uresnet_pytorch.  The reference exposes no FFI of its own: its
 * operator surface for this path is the Python nn.Module API
 * (reference uresnet/models/uresnet_sparse.py:7-37, uresnet/trainval.py:139-146),
 * and all sparse arithmetic is reached through `sparseconvnet` call sites.  Each
 * entry point below names the reference call site whose native work it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the
 *     caller (PyTorch's caching allocator) and borrowed for the duration of the call;
 *   - every launch goes on the hipStream_t passed as `stream` (void*); the library
 *     never synchronises the device (exceptions, both explicit: urn_net_probe, urn_prof_read)
 *     and never allocates device memory;
 *   - return 0 on success, a negative URN_E* code otherwise, message through
 *     urn_last_error() (thread-local); nothing throws or aborts across the ABI;
 *   - coords rows are (x, y, z, batch) int32; gather tables are [K][ld] int32 with
 *     -1 = no active neighbour; feature matrices are row-major (N, C) fp32;
 *   - site counts produced on the device live in int32 device words so that the
 *     whole integer phase runs without a host round trip; kernels take
 *     (n_dev, n_cap): n_dev may be NULL, then n_cap is the exact count.
 */
#ifndef URESNET_HIP_H
#define URESNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define URN_OK 0
#define URN_EINVAL (-1)       /* bad argument (shape, alignment, null) */
#define URN_EHIP (-2)         /* a HIP runtime call failed */
#define URN_EUNSUPPORTED (-3) /* shape outside what the kernels were built for */

int urn_version(void);
const char *urn_last_error(void);

/* ------------------------------------------------------------------ integer phase
 * Hash tables: `hcap` slots (power of two, from urn_hash_capacity), laid out as
 * keys uint64[hcap] | first int32[hcap] | site int32[hcap]; the caller hands one
 * buffer of urn_hash_bytes(hcap) bytes and clears it with urn_hash_clear. */
int64_t urn_hash_capacity(int64_t n_rows);
int64_t urn_hash_bytes(int64_t hcap);
int64_t urn_unique_scratch_bytes(int64_t n_rows);
int urn_hash_clear(void *hash, int64_t bytes, void *stream);

/* scn.InputLayer(dimension, SPATIAL_SIZE, mode=3)  (reference uresnet_sparse.py:20,33-35):
 * active sites in first-occurrence order.  row2site[n], site_coords[n_cap][4],
 * n_active = device int32.  `hash` must be cleared; afterwards it maps coord -> site. */
int urn_sites_build(const int32_t *coords, int64_t n, int spatial, void *hash, int64_t hcap,
                    void *scratch, int64_t scratch_bytes, int32_t *row2site,
                    int32_t *site_coords, int32_t *n_active, void *stream);

/* InputLayer feature merge: site_feats[s,:] = sum (mode 3) of feats[i,:] over rows with
 * row2site[i]==s, accumulated in fp64 (order independent), rounded once.
 * acc64 is scratch of n_cap*nf doubles. */
int urn_input_features(const float *feats, const int32_t *row2site, int64_t n, int nf,
                       const int32_t *n_active, int64_t n_cap, double *acc64, float *site_feats,
                       void *stream);

/* scn.SubmanifoldConvolution rulebook (reference uresnet_sparse.py:21,22): 27-offset
 * neighbour table nbr[27][ld], o = ((dx+1)*3+(dy+1))*3+(dz+1), neighbour = site+(dx,dy,dz).
 * n_rules (device int32, may be NULL) receives the number of valid entries. */
int urn_rulebook_subm(const int32_t *site_coords, const int32_t *n_dev, int64_t n_cap, int spatial,
                      const void *hash, int64_t hcap, int32_t *nbr, int64_t ld, int32_t *n_rules,
                      void *stream);

/* scn.Convolution(d, a, b, 2, 2, False) site creation inside scn.UNet (reference
 * uresnet_sparse.py:22): coarse = fine>>1 in first-touch order.  Outputs
 * coarse_coords[n_cap][4], parent[n_cap], off[n_cap] (0..7), n_coarse (device), and the
 * coarse level's hash (cleared by the caller beforehand). */
int urn_level_down(const int32_t *fine_coords, const int32_t *n_fine, int64_t n_cap, void *hash,
                   int64_t hcap, void *scratch, int64_t scratch_bytes, int32_t *coarse_coords,
                   int32_t *parent, int32_t *off, int32_t *n_coarse, void *stream);

/* urn_level_down and urn_down_tables in one pass (chd/up as below, pre-filled with -1; both NULL = level_down only) */
int urn_level_down_tables(const int32_t *fine_coords, const int32_t *n_fine, int64_t n_cap, void *hash,
                          int64_t hcap, void *scratch, int64_t scratch_bytes, int32_t *coarse_coords,
                          int32_t *parent, int32_t *off, int32_t *n_coarse, int32_t *chd, int64_t ld_c,
                          int32_t *up, int64_t ld_f, void *stream);
/* Sites of ALL levels of one geometry from the input rows in four launches: the results of urn_sites_build followed by
 * num_levels-1 urn_level_down_tables calls (same site numbering: level l's sites are the distinct (coords >> l, batch) of
 * the rows, numbered by first occurrence -- see urn_hash.hip for why that equals the level-by-level numbering).
 * Arrays of num_levels host pointers: hash (each hcap slots, cleared), site_coords [n][4]; of num_levels-1: parent, off
 * [n], chd, up ([8][ld], pre-filled with -1; the arrays may be NULL).  n_sites: device int32[num_levels].
 * scratch: urn_levels_scratch_bytes(n, num_levels).  Replaces scn.InputLayer + the metadata of scn.Convolution(2,2)
 * (reference uresnet/models/uresnet_sparse.py:20-22). */
int64_t urn_levels_scratch_bytes(int64_t n, int num_levels);
int urn_sites_build_levels(const int32_t *coords, int64_t n, int spatial, int num_levels, void *const *hash,
                           int64_t hcap, void *scratch, int64_t scratch_bytes, int32_t *row2site,
                           int32_t *const *site_coords, int32_t *n_sites, int32_t *const *parent,
                           int32_t *const *off, int32_t *const *chd, int32_t *const *up, int64_t ld, void *stream);
/* urn_rulebook_subm for up to 8 levels of one geometry in a single launch: arrays of num_levels host pointers /
 * values (site coordinates, device counts, spatial sizes, hashes of capacity hcap, tables [27][ld]) */
int urn_rulebook_subm_multi(int num_levels, const int32_t *const *site_coords, const int32_t *const *n_dev,
                            int64_t n_cap, const int *spatial, const void *const *hash, int64_t hcap,
                            int32_t *const *nbr, int64_t ld, void *stream);

/* Gather tables of the strided pair: chd[8][ld_c] (coarse j <- fine child at offset o) and
 * up[8][ld_f] (fine i <- parent[i] at o == off[i]).  Both must be pre-filled with -1
 * by the caller (urn_fill_i32). */
int urn_down_tables(const int32_t *parent, const int32_t *off, const int32_t *n_fine, int64_t n_cap,
                    int32_t *chd, int64_t ld_c, int32_t *up, int64_t ld_f, void *stream);

int urn_fill_i32(int32_t *p, int64_t n, int32_t v, void *stream);

/* Compacted rule lists ("pair lists") of gather tables, the form the MFMA kernels consume (the rulebook of
 * scn.SubmanifoldConvolution / Convolution / Deconvolution, reference uresnet_sparse.py:21-22, grouped per tile of
 * `tile` (32, 64 or 128) output rows).  One tile = int32 words [0] number of blocks | bytes 4..31: first block of every
 * table row t | the table row t of every block (K*tile/16 words, padded to a multiple of 4) | 16 words per block, word =
 * input row | (row inside the tile << 24); blocks are ordered by t, a table row's last block is padded with words
 * (tile << 24) (gather row 0, discard).  urn_pairs_bytes = size of one list; urn_pairs_build compacts n_tables tables (host
 * arrays of n_tables entries; n_dev[i] = device int32 row count or NULL = n_cap[i]) in one launch.
 * Deterministic: the list depends on the table only. */
int64_t urn_pairs_bytes(int64_t n_cap, int K, int tile);
int urn_pairs_build(int n_tables, const int32_t *const *tbl, const int64_t *ld, const int *K,
                    const int32_t *const *n_dev, const int64_t *n_cap, const int *tile, int32_t *const *pairs,
                    void *stream);

/* -------------------------------------------------------------------- float phase
 * Gather convolution, the one arithmetic kernel behind SubmanifoldConvolution,
 * Convolution(k2,s2) and Deconvolution(k2,s2) forward and input-gradient:
 *     y[j,:] = sum_{o<K} x[tbl[t(o)*ld + j], :] @ W[o]   (+ res[j,:] if res)
 * with t(o) = flip ? K-1-o : o.  `wt` is W pre-transposed to (K, cout, cin).
 * cin, cout multiples of 16 use the MFMA kernel (v_mfma_f32_16x16x4_f32); other
 * widths (the 1-channel stem) use a VALU kernel.  y may not alias x. */
int urn_gconv_fwd(const float *x, const float *wt, const int32_t *tbl, int64_t ld, int K, int flip,
                  int64_t n_out, int cin, int cout, const float *res, float *y, void *stream);

/* Weight gradient of the same op: dw[o] (+)= sum_j x[tbl[o*ld+j],:]^T (x) dy[j,:],
 * dw is (K, cin, cout) and is ACCUMULATED into (fp32 atomics; caller zeroes). */
int urn_gconv_bwd_dw(const float *x, const float *dy, const int32_t *tbl, int64_t ld, int K,
                     int64_t n_out, int cin, int cout, float *dw, void *stream);

/* Weight gradient on the compacted rule list of the table (NULL list = identity table, K == 1), two stages without
 * atomics: per (share of the tiles, table row) partial sums in `scratch` (urn_gconv_dw_pairs_scratch_bytes), then
 * dw += the partials in a fixed order -- bitwise reproducible.  x rows are used as relu(x*scale+shift) when xf_scale is
 * given; ldx / ld_dy = row strides of x / dy in floats (0 = cin / cout).  cin, cout multiples of 16. */
int64_t urn_gconv_dw_pairs_scratch_bytes(int64_t n_out, int tile, int K, int cin, int cout);
int urn_gconv_bwd_dw_pairs(const float *x, int64_t ldx, const float *xf_scale, const float *xf_shift, const float *dy,
                           int64_t ld_dy, const int32_t *pairs, int tile, int K, int64_t n_out, int cin, int cout,
                           float *dw, void *scratch, int64_t scratch_bytes, void *stream);

/* (K, a, b) -> (K, b, a) */
int urn_transpose_w(const float *w, int K, int a, int b, float *wt, void *stream);

/* scn.BatchNormReLU / BatchNormLeakyReLU(leak 0) over the (n, c) row matrix, batch
 * statistics (biased variance), fp64 accumulation.  scratch: urn_bn_scratch_bytes(c).
 * mean/invstd (c) are outputs of fwd and inputs of bwd.  relu: 0/1.
 * bwd ACCUMULATES into dgamma/dbeta (caller zeroes), like every parameter gradient here.
 * running_mean/var may be NULL; otherwise updated with `momentum` (new = m*old + (1-m)*batch). */
int64_t urn_bn_scratch_bytes(int c);
int urn_bn_relu_fwd(const float *x, int64_t n, int c, const float *gamma, const float *beta,
                    double eps, int relu, float *y, float *mean, float *invstd,
                    float *running_mean, float *running_var, double momentum, void *scratch,
                    void *stream);
/* Affine+ReLU with given statistics (eval mode: running stats). */
int urn_bn_relu_apply(const float *x, int64_t n, int c, const float *gamma, const float *beta,
                      const float *mean, const float *invstd, int relu, float *y, void *stream);
int urn_bn_relu_bwd(const float *x, const float *y, const float *dy, int64_t n, int c,
                    const float *gamma, const float *mean, const float *invstd, int relu,
                    float *dx, float *dgamma, float *dbeta, void *scratch, void *stream);

/* ---- fused BatchNorm+ReLU pieces (what the executor uses) --------------------------------
 * The BatchNorm passes are folded into the neighbouring gather convolutions:
 *   forward : the producing conv writes per-tile column partials (sum, sum of squares; fp64) of its
 *             output [epilogue 1]; urn_bn_finalize_fwd turns them into mean/invstd and the folded
 *             affine scale = gamma*invstd, shift = beta - mean*scale; the consuming conv applies
 *             relu(x*scale + shift) to the rows it gathers [xf_scale/xf_shift] -- the normalised
 *             tensor never exists in HBM;
 *   backward: the conv computing the gradient w.r.t. the BatchNorm OUTPUT applies the ReLU mask and
 *             reduces (sum g, sum g*xhat) per tile [epilogue 2]; urn_bn_finalize_bwd accumulates
 *             dgamma/dbeta and emits the two coefficients; urn_bn_bwd_apply forms
 *             dx = gamma*invstd*(g - c0 - xhat*c1) (+ extra, e.g. the residual branch's gradient).
 * Partial slabs are [n_part][2][c] doubles, summed in a fixed order (deterministic). */
typedef struct {
    const float *x, *wt;           /* gathered rows (n_in, cin); weights pre-transposed (K, cout, cin) */
    const int32_t *tbl;            /* [K][ld] */
    int64_t ld;
    int K, flip;
    int64_t n_out;
    int cin, cout;
    const float *res;              /* optional residual added to y */
    float *y;
    const float *xf_scale, *xf_shift; /* optional (cin): rows are used as relu(x*scale+shift) */
    int epilogue;                  /* 0 none | 1 column stats of y | 2 BatchNorm-backward reduce */
    double *part;                  /* epilogue != 0: slab of urn_gconv_part_bytes(n_out, cout) bytes */
    const float *e_x, *e_scale, *e_shift, *e_mean, *e_invstd; /* epilogue 2: the BatchNorm's input (n_out, cout) and folded affine */
    /* Optional: finalize the epilogue partials before the call's work completes, without a separate
     * urn_bn_finalize_* launch where the kernel supports it (the last workgroup to finish reduces the slab:
     * agent-scope release / ticket / acquire on `sync_word`, a zeroed device uint32 that the kernel resets).
     * fin_n = row count of the statistics.  Epilogue 1: up to two consuming BatchNorms each receive
     * mean/invstd/scale/shift over all cout columns (+ running-stat update when the pointers are set).
     * Epilogue 2: dgamma/dbeta are accumulated into, coef0/coef1 written. */
    uint32_t *sync_word;
    int64_t fin_n;
    double fin_eps, fin_momentum;
    struct {
        const float *gamma, *beta;
        float *mean, *invstd, *scale, *shift, *running_mean, *running_var;
    } fin_bn[2];
    float *fin_dgamma, *fin_dbeta, *fin_coef0, *fin_coef1;
    /* Accumulated statistics (2-D tile kernel only; removes the urn_bn_finalize_* launches from the chain).
     * Producer side: part_slots > 0 makes the epilogue ADD its column sums with fp64 atomics into row
     * (workgroup % part_slots) of `part`, a [part_slots][2][cout] slab the caller has zeroed -- the same layout
     * as a partial slab with n_part = part_slots, so urn_bn_finalize_* still apply to it.  The order of the
     * additions is not fixed: the fp64 sums can differ in their last bits from run to run.
     * Consumer side: xs_sums[0] != NULL makes every workgroup derive the folded affine of its input rows from
     * such slabs instead of reading xf_scale/xf_shift: channels [0, xs_split) from xs_sums[0] (row length
     * xs_ld[0]), channels [xs_split, cin) from xs_sums[1] (a channel concat; xs_split = cin when there is one
     * slab); statistics over xs_n rows, fin_eps; workgroup 0 also stores mean/invstd/scale/shift (cin each, for
     * the backward pass) and updates the running statistics with fin_momentum when those pointers are set. */
    int part_slots;
    int xs_slots, xs_split;
    int xs_ld[2];
    const double *xs_sums[2];
    int64_t xs_n;
    const float *xs_gamma, *xs_beta;
    float *xs_mean, *xs_invstd, *xs_scale, *xs_shift, *xs_running_mean, *xs_running_var;
    /* MFMA operand precision of this call: 0 = the library default (fp32 unless urn_set_option("gconv_precision", p)),
     * 1 = fp32 explicitly, 2 = bf16, 3 = fp16.  Reduced precision rounds the gathered rows and the weights (RNE) while
     * they are staged in LDS; tensors in HBM and the accumulation stay fp32 (BASELINE configs[1] bf16, configs[4] fp16). */
    int precision;
    /* Row strides (in floats) of x and y when they are column blocks of wider row matrices -- the halves of a channel
     * concat: 0 = dense (cin / cout).  res, e_x and the statistics slabs are always dense.  2-D tile kernel only. */
    int64_t ldx, ldy;
    /* Compacted rule list of `tbl` (urn_pairs_build) and its tile size (32, 64 or 128 rows): the call then runs on the
     * pair-list kernel -- per tile of output rows and table row, the valid (input row, output row) pairs packed into
     * blocks of 16, so that the matrix pipe multiplies rules, not (16-row block, offset) products that are mostly
     * absent neighbours.  pairs == NULL with pairs_tile != 0 and K == 1 declares tbl the identity (a 1x1 convolution:
     * scn.NetworkInNetwork).  pairs_tile == 0: the dense-table kernels.  Reduced precision: see wt_frag_prec. */
    const int32_t *pairs;
    int pairs_tile;
    /* Optional, pair-list kernel only: the weights of `wt` once more in MFMA-fragment order,
     *   wt_frag[((o * (cout / 16) + cb) * (cin / 16) + kb) * 256 + (q * 16 + r) * 4 + i] = wt[o][16 cb + r][16 kb + 4 q + i]
     * (urn_weight_fragments).  A wave then reads the 16 x 16 block of an offset as ONE contiguous kilobyte (8 cache lines)
     * instead of 16 rows of 64 bytes (16 half-used lines): the kernel is bound by the cache lines a CU can address per
     * cycle, and the weight block is fetched again for nearly every block of 16 rules.  NULL: rows of wt. */
    const float *wt_frag;
    /* Element type of wt_frag: 0 = fp32 (above), 1 = bf16, 2 = fp16 (urn_weight_fragments16: the same order with 16-bit
     * elements, 512 bytes per 16 x 16 block -- a lane's 8 bytes ARE its A operand of v_mfma_f32_16x16x16_*; when cin / 16
     * is even, blocks kb and kb + 1 (kb even) share one kilobyte with a lane's 16 bytes = [its 8 of kb | its 8 of kb + 1]:
     * one 16-byte load per lane, the A operand of v_mfma_f32_16x16x32_*).  The
     * reduced-precision variants of the pair-list kernel run only on fragments of their own precision (they keep the
     * weight blocks of an offset in half the registers and take two to four column blocks per wave); any other
     * combination runs on the 2-D tile kernel. */
    int wt_frag_prec;
} urn_gconv_args;
/* wt (K, cout, cin) -> fragment order (see urn_gconv_args.wt_frag); cin, cout multiples of 16 */
int urn_weight_fragments(const float *wt, int K, int cout, int cin, float *wt_frag, void *stream);
/* the same with elements rounded (RNE) to bf16 (precision 1) or fp16 (2): K * cout * cin 16-bit words */
int urn_weight_fragments16(const float *wt, int K, int cout, int cin, int precision, void *wt_frag16, void *stream);
int64_t urn_gconv_part_bytes(int64_t n_out, int cout);
int urn_gconv_fwd_ex(const urn_gconv_args *args, int *n_part, void *stream);
/* weight gradient with the same input transform: x rows are used as relu(x*scale+shift) */
int urn_gconv_bwd_dw_ex(const float *x, const float *xf_scale, const float *xf_shift, const float *dy,
                        const int32_t *tbl, int64_t ld, int K, int64_t n_out, int cin, int cout, float *dw,
                        void *stream);
/* the same with a row stride ld_dy >= cout of dy (dy is a column block of a wider matrix) */
int urn_gconv_bwd_dw_strided(const float *x, const float *xf_scale, const float *xf_shift, const float *dy, int64_t ld_dy,
                             const int32_t *tbl, int64_t ld, int K, int64_t n_out, int cin, int cout, float *dw,
                             void *stream);
/* the same without atomics: every (row chunk, offset, channel tile) workgroup STORES its partial into `scratch`
 * (urn_gconv_dw_2stage_scratch_bytes), a second launch adds the partials to dw in a fixed order -- bitwise reproducible.
 * cin, cout multiples of 16. */
int64_t urn_gconv_dw_2stage_scratch_bytes(int K, int64_t n_out, int cin, int cout);
int64_t urn_gconv_dw_2stage_scratch_max(void);   /* enough for every shape (K <= 27): a few tens of MB */
int urn_gconv_bwd_dw_2stage(const float *x, const float *xf_scale, const float *xf_shift, const float *dy, int64_t ld_dy,
                            const int32_t *tbl, int64_t ld, int K, int64_t n_out, int cin, int cout, float *dw, void *scratch,
                            int64_t scratch_bytes, void *stream);
/* column partials of a tensor whose producer cannot fuse them (the 1-channel stem) */
int urn_bn_stats_partial(const float *x, int64_t n, int c, double *part, int *n_part, void *stream);
int urn_bn_finalize_fwd(const double *part, int n_part, int64_t n, int c, int part_ld, double eps,
                        const float *gamma, const float *beta, float *mean, float *invstd, float *scale,
                        float *shift, float *running_mean, float *running_var, double momentum,
                        void *stream);
int urn_bn_finalize_bwd(const double *part, int n_part, int64_t n, int c, float *dgamma, float *dbeta,
                        float *coef0, float *coef1, void *stream);
int urn_bn_bwd_apply(const float *x, const float *g, const float *extra, int64_t n, int c,
                     const float *gamma, const float *mean, const float *invstd, const float *coef0,
                     const float *coef1, float *dx, void *stream);
/* the same with the coefficients taken from an accumulated slab ([slots][2][c], see part_slots): c0 = sum g / n,
 * c1 = sum g*xhat / n; dgamma/dbeta are ACCUMULATED into by the first workgroup.  c <= 512, c % 4 == 0.
 * ld_extra = row stride of `extra` in floats (0 = c; larger when it is a column block of a wider matrix). */
int urn_bn_bwd_apply_sums(const float *x, const float *g, const float *extra, int64_t ld_extra, int64_t n, int c,
                          const float *gamma, const float *mean, const float *invstd, const double *sums,
                          int slots, float *dgamma, float *dbeta, float *dx, void *stream);

/* scn.OutputLayer (reference uresnet_sparse.py:24): y[i,:] = x[idx[i],:]; and its
 * backward dx[idx[i],:] += dy[i,:] (fp32 atomics; caller zeroes dx). */
int urn_rows_gather(const float *x, const int32_t *idx, int64_t n, int c, float *y, void *stream);
int urn_rows_scatter_add(const float *dy, const int32_t *idx, int64_t n, int c, float *dx,
                         void *stream);

/* ---- head and loss on the device -------------------------------------------------
 * urn_head_fwd: logits[i,:] = x[row2site[i],:] @ W^T + b -- scn.OutputLayer + torch.nn.Linear (reference
 * uresnet_sparse.py:24-25,36) in one pass; row2site NULL = identity.  W is (nc, m) like Linear.weight.
 * urn_head_bwd: dx (+= with atomics when row2site is given, plain stores otherwise), dW and db ACCUMULATED.
 * urn_ce_fwd / urn_ce_bwd: SegmentationLoss (reference uresnet_sparse.py:46-82): per-event mean of the
 * (optionally weighted) voxel cross-entropy, summed over events; out[0] = loss, out[1] = sum of per-event
 * accuracies.  batch ids and labels are the float columns the reference passes (data[:, -2], label[:, 0]);
 * ev = scratch of urn_ce_scratch_bytes(), row_lse (n) is kept for the backward.  No host sync per event. */
int urn_head_fwd(const float *x, const int32_t *row2site, int64_t n, int m, int nc, const float *W,
                 const float *b, float *logits, void *stream);
int urn_head_bwd(const float *dlogits, const float *x, const int32_t *row2site, int64_t n, int m, int nc,
                 const float *W, float *dx, float *dW, float *db, void *stream);
int64_t urn_ce_scratch_bytes(void);
int urn_ce_fwd(const float *logits, const float *label, const float *batch_id, int batch_id_stride,
               const float *weight, int64_t n, int nc, float *row_lse, double *ev, float *out, void *stream);
int urn_ce_bwd(const float *logits, const float *label, const float *batch_id, int batch_id_stride,
               const float *weight, const float *row_lse, const double *ev, const float *grad_out, int64_t n,
               int nc, float *dlogits, void *stream);

/* torch.optim.Adam (reference uresnet/trainval.py:37) over one contiguous fp32 segment: p, g, exp_avg m and
 * exp_avg_sq v of n elements each; `step` is the 1-based step count AFTER the increment (bias corrections
 * 1 - beta^step are formed on the host in double).  weight_decay is the L2 form (added to the gradient). */
int urn_adam_flat(float *p, const float *g, float *m, float *v, int64_t n, double lr, double beta1, double beta2,
                  double eps, double weight_decay, int64_t step, void *stream);

/* ------------------------------------------------------------------ dense model
 * Dense 2-D / 3-D convolution family of the dense U-ResNet (reference uresnet/models/uresnet_dense.py:29-83 ResNetModule,
 * :164-175 ConvTranspose k3 s2 p1 op1, :201-226 forward): implicit GEMM on the matrix cores, the input box of a
 * workgroup's outputs staged once in LDS, F.pad(mode='replicate') folded into the addressing -- no index tables, no padded
 * copies.  Activations are channels-last row matrices (batch * Z * Y * X rows of C floats, row stride ld) in fp32;
 * wt = weights as [tap][cout][cin] fp32 (tap = (kz * kdim[1] + ky) * kdim[2] + kx).  precision 0: fp32 operands
 * (v_mfma_f32_16x16x4_f32), 1: operands rounded to bf16 in LDS (v_mfma_f32_16x16x16_bf16), fp32 accumulation.
 * One call computes a sub-grid of the output: out[p + os * u] (u over Sub) = bias + sum over the taps j (nt per dimension)
 * of wt[tap(wi[.][j])] applied to in[s * u + e[.][j]], out-of-range inputs clamped (mode 0) or skipped (mode 1); dimensions
 * in z, y, x order, a 2-D volume has Z = 1 and one tap in z.  The four forms of the model (conv forward, its input
 * gradient on the padded volume, transposed conv forward per output parity class, its input gradient) are geometry
 * descriptions of this one kernel (uresnet_pytorch_amd/dense_hip.py builds them).  cin, cout multiples of 16. */
typedef struct {
    int In[3], Out[3], Sub[3], p[3], os[3], s[3];
    int nt[3], e[3][3], wi[3][3], kdim[3];
    int mode;
} urn_dense_geom;
/* scratch (urn_dense_conv_scratch_bytes, may be NULL): launches with few output tiles (the deep levels: 4^3 voxels x 512
 * channels) split the contraction over workgroups into partial outputs there and sum them in a fixed order. */
int64_t urn_dense_conv_scratch_bytes(int cout, int batch, const urn_dense_geom *geom);
/* stats (may be NULL): a zeroed [stat_slots][2][cout] fp64 slab; the call ADDS the column sums and sums of squares of the
 * outputs it writes (fp64 atomics from the epilogue, workgroup % stat_slots picks the row) -- the batch statistics of the
 * BatchNorm that follows every convolution of the model without a pass over y; several calls (the parity classes of a
 * transposed conv) accumulate into one slab; urn_bn_finalize_fwd(stats, stat_slots, ...) turns it into mean / invstd /
 * scale / shift.  Needs 256 % (cout / 4) == 0.
 * xf_scale / xf_shift (cin each, may be NULL): the input is used as x * scale + shift -- the BatchNorm of the producing
 * convolution folded into this one's load (reference uresnet_dense.py:78-81: no ReLU between residual1 and residual2); the
 * normalised tensor then never exists in HBM.  urn_dense_dw takes the same pair for its x operand. */
int urn_dense_conv(const float *x, int64_t ldx, int cin, const float *wt, const float *bias, float *y, int64_t ldy, int cout,
                   int batch, const urn_dense_geom *geom, int precision, double *stats, int stat_slots, const float *xf_scale,
                   const float *xf_shift, void *scratch, int64_t scratch_bytes, void *stream);
/* Gradient of F.pad(mode='replicate') (reference uresnet_dense.py:75-80): dx[i] = sum of dxp over the padded positions
 * that clamp to voxel i.  dxp: rows of the padded volume (dims + pad_lo + pad_hi), dx: rows of the volume; c % 4 == 0. */
/* Weight gradient of the same convolutions: dw[tap][ci][co] (+)= sum over outputs o of x[in(o, tap)][ci] * dy[o][co] with
 * in = s * o + tap - lo, clamped (mode 0: the replicate-padded conv) or skipped when out of range (mode 1: the transposed
 * conv, called with the roles of input and output swapped).  in_dims / out_dims / k / s / lo in z, y, x order.  Two stages,
 * no atomics: per-share partial sums in `scratch` (urn_dense_dw_scratch_bytes), then the shares are summed in a fixed
 * order: dw_layout 0 ADDS them to dw[tap][cin][cout]; dw_layout 1 WRITES torch's parameter layout
 * dw[(co * cin_valid + ci) * taps + tap] for ci < cin_valid, co < cout_valid (the zero-padded channels dropped): the
 * .grad of nn.Conv's (cout, cin, *k) weight -- and of nn.ConvTranspose's (cin, cout, *k) when the call has the roles swapped. */
int64_t urn_dense_dw_scratch_bytes(int batch, const int *out_dims, const int *k, int cin, int cout);
int urn_dense_dw(const float *x, int64_t ldx, int cin, const float *dy, int64_t ld_dy, int cout, int batch, const int *in_dims,
                 const int *out_dims, const int *k, const int *s, const int *lo, int mode, float *dw, int dw_layout, int cin_valid,
                 int cout_valid, const float *xf_scale, const float *xf_shift, void *scratch, int64_t scratch_bytes, int precision,
                 void *stream);
int urn_dense_fold(const float *dxp, float *dx, int batch, const int *dims, const int *pad_lo, const int *pad_hi, int c,
                   void *stream);
/* Input gradient of a stride-1 convolution on the replicate-padded input, un-padding included: urn_dense_conv with gm =
 * the geometry of the PADDED volume (dy rows -> dxp rows, weights [tap][cin][cout]) followed by urn_dense_fold into dx
 * (rows of the volume itself, dims = Z, Y, X) -- as one call, so that the 3 x 3 x 3 fast path can write the voxels inside
 * the volume straight to dx (only the border shell goes to dxp, and only the boundary voxels are folded).  dxp: the padded
 * volume's rows (scratch, contents undefined afterwards); ld_dx == cin. */
int urn_dense_conv_dgrad_fold(const float *dy, int64_t ld_dy, int cout, const float *wb, float *dxp, float *dx, int64_t ld_dx,
                              int cin, int batch, const urn_dense_geom *gm, const int *dims, const int *pad_lo, const int *pad_hi,
                              int precision, void *scratch, int64_t scratch_bytes, void *stream);
/* Both weight layouts urn_dense_conv reads -- forward [tap][cout_p][cin_p], input gradient [tap][cin_p][cout_p], channel
 * counts zero-padded -- of n convolutions in ONE launch, from torch's parameter layouts: descs = n records of 11 int64:
 * src, fwd, bwd (device pointers), taps, transposed (0: nn.Conv (cout, cin, taps); 1: nn.ConvTranspose (cin, cout, taps)),
 * cin, cout, cin_p, cout_p, 0, 0. */
int urn_dense_weight_layouts(int n, const int64_t *descs, void *stream);
/* Row passes around the convolutions (reference uresnet_dense.py:72-83: conv -> BatchNorm(batch statistics) [-> + shortcut]
 * [-> ReLU]); (n, c) fp32 row matrices, c % 4 == 0 and 256 % (c / 4) == 0.
 * urn_dense_bn_act_fwd:  out = [relu](raw * scale + shift [+ res | + res * res_scale + res_shift]) in ONE pass; res NULL =
 *   no shortcut, res_scale NULL = identity shortcut, else res is the raw output of the shortcut conv with its own folded
 *   BatchNorm.  scale / shift come from urn_bn_finalize_fwd on the statistics slab of urn_dense_conv.
 * urn_dense_bn_act_bwd_reduce: with g = d_out * [out > 0] (out NULL = no ReLU), ADDS (sum g, sum g * xhat) per channel into
 *   the zeroed [slots][2][c] fp64 slab `sums` (xhat = (raw - mean) * invstd) and, with res_raw, the same for the shortcut
 *   BatchNorm into `res_sums`; urn_bn_finalize_bwd(sums, slots, ...) gives dgamma / dbeta and the two coefficients.
 * urn_dense_bn_act_bwd_apply: d_raw = gamma * invstd * (g - coef0 - xhat * coef1); d_res (may be NULL) = the same for the
 *   shortcut BatchNorm when res_raw is given, else g itself (the gradient of an identity shortcut). */
int urn_dense_bn_act_fwd(const float *raw, const float *scale, const float *shift, const float *res, const float *res_scale,
                         const float *res_shift, int relu, float *out, int64_t n, int c, void *stream);
int urn_dense_bn_act_bwd_reduce(const float *d_out, const float *out, const float *raw, const float *mean, const float *invstd,
                                const float *res_raw, const float *res_mean, const float *res_invstd, int64_t n, int c,
                                double *sums, double *res_sums, int slots, void *stream);
/* (sum g, sum g * xhat) slabs -> dgamma, dbeta (WRITTEN) and coef0 = sum g / n, coef1 = sum g * xhat / n; nb = 1 or 2
 * BatchNorms in one launch: slab b at sums + b * slots * 2 * c, outputs as rows of out[nb][4][c] = dgamma, dbeta, coef0, coef1 */
int urn_dense_bn_bwd_finalize(const double *sums, int nb, int slots, int64_t n, int c, float *out, void *stream);
int urn_dense_bn_act_bwd_apply(const float *d_out, const float *out, const float *raw, const float *gamma, const float *mean,
                               const float *invstd, const float *coef0, const float *coef1, const float *res_raw,
                               const float *res_gamma, const float *res_mean, const float *res_invstd, const float *res_coef0,
                               const float *res_coef1, float *d_raw, float *d_res, int64_t n, int c, void *stream);
/* DenseSegmentationLoss of ONE event (reference uresnet_dense.py:246-258) on the device, no host synchronisation: over the n
 * voxels (logits rows of nc floats, row stride ld; label / data / weight one float per voxel, weight may be NULL), with
 * mask = data > 1e-6: out[0] = sum(ce * weight * mask) / sum(mask), out[1] = sum(mask * [argmax == label]) / sum(mask).
 * acc: 3 zeroed doubles (the three sums, kept for the backward), row_lse (n) kept too.
 * urn_dense_ce_bwd: dlogits (n, nc, dense) = grad_out[0] * weight * mask / sum(mask) * (softmax - onehot(label)). */
int urn_dense_ce_fwd(const float *logits, int64_t ld, const float *label, const float *data, const float *weight, int64_t n,
                     int nc, float *row_lse, double *acc, float *out, void *stream);
int urn_dense_ce_bwd(const float *logits, int64_t ld, const float *label, const float *data, const float *weight,
                     const float *row_lse, const double *acc, const float *grad_out, int64_t n, int nc, float *dlogits,
                     void *stream);

/* ------------------------------------------------------------------ whole-network executor
 * The trunk of the sparse model -- everything between scn.InputLayer and torch.nn.Linear at
 * reference uresnet_sparse.py:19-25 -- run from C++: the same kernels as the per-layer entry
 * points, issued back-to-back on `stream`, activations carved from ONE caller-provided workspace.
 * Parameters (and gradients) are one flat fp32 buffer in module registration order; running BN
 * statistics are one flat buffer ([mean(c) | var(c)] per BatchNorm, registration order).
 * Geometry is handed over as plain arrays (per-level counts and table pointers, leading dim ld).
 * urn_net_backward replays the LAST urn_net_forward of the same handle and needs the same
 * workspace untouched; it ACCUMULATES into `grads` (caller zeroes). */
typedef struct urn_net urn_net;
#define URN_NET_UNFUSED 1        /* keep BatchNorm as separate passes (debug / A-B) */
#define URN_NET_SINGLE_STREAM 2  /* do not run weight gradients on a side stream */
#define URN_NET_SLAB_STATS 4     /* BatchNorm statistics through per-workgroup slabs + finalize launches (fixed summation
                                    order) instead of the accumulated slabs (part_slots) */
int urn_net_create(int m, int num_levels, int reps, int num_class, double eps, double momentum, int flags,
                   urn_net **out);
void urn_net_destroy(urn_net *net);
int64_t urn_net_param_count(const urn_net *net);
int64_t urn_net_running_count(const urn_net *net);
int urn_net_num_tensors(const urn_net *net);
int urn_net_tensor(const urn_net *net, int i, int64_t *off, int64_t *numel);
int64_t urn_net_workspace_bytes(urn_net *net, int num_levels, const int64_t *n, int64_t n_rows, int with_backward);
/* Optional, before a forward: the compacted rule lists (urn_pairs_build) of that forward's tables -- arrays of
 * num_levels host pointers (the last entry of chd/up unused; an array may be NULL) and, per level, their tile sizes.  They apply to the
 * NEXT urn_net_forward (and its backward) only; without them the executor runs on the dense tables. */
int urn_net_set_pairs(urn_net *net, int num_levels, const void *const *nbr_pairs, const void *const *chd_pairs,
                      const void *const *up_pairs, const int *tile_nbr, const int *tile_chd, const int *tile_up);
/* Optional, before the integer phase of a step is enqueued: write the weight copies of the coming forward (transposed and
 * both MFMA-fragment orders) into wbuf (3 * urn_net_param_count floats, owned by the caller, valid until that forward's
 * backward has run), on the executor's side stream behind everything queued on `stream` so far -- beside the integer phase
 * instead of in front of the first convolution.  urn_net_forward orders them in front of its launches. */
int urn_net_prepare_weights(urn_net *net, const float *params, float *wbuf, int64_t wbuf_floats, void *stream);
int urn_net_forward(urn_net *net, int num_levels, int64_t ld, const int64_t *n, const void *const *nbr,
                    const void *const *chd, const void *const *up, const int32_t *row2site, int64_t n_rows,
                    const float *params, float *running, const float *site_feats, void *ws, int64_t ws_bytes,
                    float *out_rows, int training, void *stream);
int urn_net_backward(urn_net *net, const float *d_rows, float *grads, void *stream);
/* Overlap of the gradient all-reduce with the backward pass (data-parallel ranks).  The flat buffers are laid out
 * [stem | encoder levels 0..L-2 | bottom level | decoder levels L-2..0 | last BatchNorm | (head)], and the backward pass
 * finishes them from the back: [urn_net_suffix_offset, end) is complete when the bottom level's backward has run, the
 * encoder prefix only at the end.  urn_net_backward_cb is urn_net_backward with a call-back at that point: every kernel that
 * writes into the suffix has been enqueued -- input gradients and BatchNorm gradients on `stream`, weight gradients on the
 * executor's side stream (urn_net_side_stream; NULL for a single-stream handle), which has been ordered behind `stream` --
 * so a collective issued behind the side stream's tail reduces the suffix while the encoder half of the backward pass runs. */
int64_t urn_net_suffix_offset(const urn_net *net);
void *urn_net_side_stream(const urn_net *net);
int urn_net_backward_cb(urn_net *net, const float *d_rows, float *grads, void *stream, void (*bottom_done)(void *), void *user);
/* Optional, before a forward: the Linear head (reference uresnet_sparse.py:25,36; W (num_class, m) row-major, b (num_class))
 * run INSIDE the executor for the NEXT urn_net_forward and its backward: the last BatchNormReLU, the OutputLayer and the
 * Linear are one kernel (urn_tail_fwd), their gradients one kernel + the BatchNorm's apply (urn_tail_bwd).  out_rows of
 * that forward is then the (n_rows, num_class) LOGITS, d_rows of its backward their gradient, and the head's parameter
 * gradients are accumulated at grads[urn_net_param_count ..] (W, then b).  Fused path with accumulated statistics only
 * (flags 0), m <= 32; otherwise URN_EUNSUPPORTED and nothing is armed. */
int urn_net_set_head(urn_net *net, const float *W, const float *b);
/* the two kernels of the tail, for direct use (see urn_net_set_head): sums = accumulated statistics slab of x
 * ([slots][2][m] doubles) from which scale / shift (and mean / invstd / running statistics) are derived, or NULL = scale /
 * shift given; part = the slab the BatchNorm-backward sums are accumulated into (zeroed by the caller).  urn_tail_bwd ADDS
 * the row gradients onto their sites (gsite (n_sites, m), zeroed by the caller) -- unless n_sites == n: every site then has
 * exactly one row (row2site is onto the sites), the gradients are stored and gsite need not be zeroed.  m 16 or 32. */
int urn_tail_fwd(const float *x, const int32_t *row2site, int64_t n, int m, int nc, const float *W, const float *b,
                 const double *sums, int slots, int64_t n_sites, double eps, const float *gamma, const float *beta,
                 float *mean, float *invstd, float *scale, float *shift, float *running_mean, float *running_var,
                 double momentum, float *logits, void *stream);
int urn_tail_bwd(const float *dlogits, const float *x, const int32_t *row2site, int64_t n, int m, int nc, const float *W,
                 const float *scale, const float *shift, const float *mean, const float *invstd, float *gsite,
                 int64_t n_sites, float *dW, float *db, double *part, int slots, void *stream);
/* Side-stream probe: the ONE call of the executor that synchronises (`stream` and its candidate streams), therefore
 * explicit and optional -- once per handle at initialisation.  Keeps the fastest of a few candidate side streams for the
 * fork/join pattern of the backward pass against `stream` (see urn_net.hip: hardware-queue aliasing). */
int urn_net_probe(urn_net *net, void *stream);

/* Test hook: the folded BatchNorm+ReLU of the last TRAINING forward on the fused path.  BatchNorm i (0 <= i <
 * urn_net_num_bn, executor order; w_off = offset of its weight in the flat parameter buffer) consumed x (rows, c) as
 * relu(x * scale + shift); urn_net_bn_export copies x, scale and shift into caller buffers (device to device, on stream).
 * Parity tests pin the oracle's ReLU masks with them (the normalised tensor itself is never written). */
int urn_net_num_bn(urn_net *net);
int urn_net_bn_info(urn_net *net, int i, int64_t *w_off, int64_t *rows, int *c);
int urn_net_bn_export(urn_net *net, int i, float *x, float *scale, float *shift, void *stream);

/* ----------------------------------------------------------------------- measurement
 * Optional per-kernel timing (HIP events on the launch stream), off by default.
 * kind 0 = gather-conv forward/input-gradient kernel, 1 = weight-gradient kernel, 2 = integer phase (one record per
 * urn_sites_build / urn_sites_build_levels / urn_level_down_tables / urn_rulebook_subm_multi call).
 * urn_prof_enable resets the records; urn_prof_read waits for the recorded events. */
int urn_prof_enable(int on);
/* Library-wide options.  Behaviour:
 *   "gconv_precision" 0 fp32 (default) | 1 bf16 | 2 fp16 -- MFMA operand precision of calls that do not set
 *                     urn_gconv_args.precision, and of the weight gradient.
 * Tuning / A-B switches (defaults are the measured optima; tools/ use them):
 *   "gconv_kernel" 7 pair-list kernel for calls that carry a list, 2-D tile otherwise (default) | 6 2-D tile | 3 register
 *                  gather; "pairs_max_cin" / "pairs_max_cout" / "pairs_nin" which shapes take the pair-list kernel;
 *   "pairs_nc" / "pairs_split" / "pairs_cbg" / "pairs_wgs" / "pairs_waves" its column blocks per wave, split of a tile's
 *   block list, column groups per workgroup and the workgroup / wave targets (0 = automatic);
 *   "dw_pairs" 1 = weight gradients on the two-stage pair-list kernel (bitwise reproducible; default 0 = dense-table kernel
 *   with fp32 atomics), "dwp_waves" / "dwp_smax" / "dwp_cap" its wave target, share limit and register block;
 *   "tile_il" 1/0 interleaved offset step of the 2-D tile kernel, "tile_il_min_ks" narrowest channel step that takes it,
 *   "tile_rb" / "tile_cb" / "tile_kc" force its workgroup tile / channels per step (0 = automatic);
 *   "dw_kernel" 2/1, "dw_blocks" workgroup target, "dw_split" 0 never | 1 automatic | 2 always, "dw_group"
 *   weight gradients per fork to the side stream; "net_side_probe" candidate side streams urn_net_probe times against the
 *   caller's stream (default 4; 0 keeps the first), "net_side_verbose" 1 prints the probe times to stderr;
 *   "gconv_dbg" timing-only ablation / probe mask of the 2-D tile kernel (results are garbage with most bits);
 *   "gconv_pipe", "gconv_min_waves" knobs of the register-gather fallback. */
int urn_set_option(const char *key, int64_t value);
int urn_prof_read(int kind, double *total_ms, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* URESNET_HIP_H */
