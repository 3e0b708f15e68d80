// Dense 3-D / 2-D convolution family of the dense U-ResNet (reference uresnet/models/uresnet_dense.py:29-83, 164-175,
// 201-226): implicit GEMM on the matrix cores with the input box of a workgroup's outputs staged ONCE in LDS (a k3
// convolution reads every input voxel 27 times: from LDS, not from L2), no index tables, no padded copies.
//
// Activations are channels-last row matrices (B * Z * Y * X rows, C floats, row stride ld) in fp32; weights come
// pre-arranged as wt[tap][cout][cin] fp32.  Operands are rounded to bf16 (PREC 1, v_mfma_f32_16x16x16_bf16) or kept in
// fp32 (PREC 0, v_mfma_f32_16x16x4_f32) while they are parked in LDS; accumulation is fp32.
//
// ONE kernel serves the four gather forms of the model.  A launch computes a SUB-GRID of the output volume:
// out[o] with o_d = p_d + os_d * u_d (u over Sub_d), and per dimension a short list of taps j < nt_d:
//     in_d = s_d * u_d + e_d[j],     weight tap index w_d[j]
// with out-of-range in_d either clamped (F.pad(mode='replicate') folded into the addressing) or skipped (zero padding):
//   A  Conv k{1,3} s{1,2} forward              os 1, s = stride, e = j - pad_lo, w = j, clamp
//   B  its input gradient on the PADDED input   s1: os 1, s 1, e = -j, w = j, zero;  s2: per parity class of the output
//      (urn_dense_fold then adds the replicated border back: the gradient of a clamp is a sum)
//   C  ConvTranspose k3 s2 p1 op1 forward       per parity class c: os 2, p = c, s 1, taps with (c + 1 - t) even, e = (c + 1 - t) / 2, zero
//   D  its input gradient                       os 1, s 2, e = t - 1, w = t, zero
// so a stride-2 transposed form is eight dense launches over same-parity outputs, each with only its valid taps.
//
// Workgroup = 8 waves, 256 outputs = 16 row blocks of 16 consecutive u_x (TY x TZ row blocks in y, z), up to 64 output
// columns (grid.y walks wider layers), input channels in chunks of 16*KC that fit the LDS budget.  Per chunk: stage the
// input box; per tap: the weight tile wt[tap][cols][chunk] is staged (double-buffered, one barrier per tap) and every
// wave multiplies its 2 row blocks x all column blocks: A fragments are rows of the box shifted by the tap.
#include "urn_common.h"
#include <string.h>

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct DenseArgs {
    const float *x, *wt, *bias;
    const float *xf_scale, *xf_shift;               // optional (cin): x is used as x * scale + shift
    float *y;
    long ldx, ldy;
    int cin, cout;
    int B;
    int In[3], Out[3], Sub[3], p[3], os[3], s[3];   // z, y, x
    int nt[3], e[3][3], wi[3][3], kdim[3];
    int mode;                                       // 0 clamp (replicate), 1 zero (skip)
    int TY, TZ;                                     // row blocks per workgroup in y and z (TY * TZ == NRB)
    int NRB;                                        // 16, or 4 for strided inputs (the staged box grows with the stride)
    int box[3];                                     // staged input box per dim: s * (tile - 1) + (emax - emin) + 1
    int emin[3];
    int kc;                                         // 16-channel groups per chunk
    int cw;                                         // output columns per workgroup (64, 32 or 16)
    int tg;                                         // taps whose weight tiles are staged together (one barrier pair per group)
    int zc, zt;                                     // split of the contraction over workgroups (blockIdx.z): chunk slices x tap slices
    float *slab;                                    // split > 1: partial outputs [z][sub-grid row][cout] instead of y
    int ntaps;
    int tap_w[27], tap_box[27];                     // per tap: weight tap index, offset of the tap inside the staged box (voxels)
    long long *stamps;                              // diagnostics (-DURN_DENSE_STAMP, urn_set_option "dense_stamp_ptr")
    double *stats;                                  // optional [stat_slots][2][cout]: column sums / sums of squares of y ADDED (fp64 atomics)
    int stat_slots;
    // input gradient on the padded volume with the un-padding folded in (k_dense_conv3 only): outputs whose padded position
    // minus fold_lo lies inside fold_in go straight to fold_y (rows of the un-padded volume, row stride ldy), only the
    // border shell is written to y -- k_dense_fold_border then adds the shell onto the boundary voxels
    float *fold_y;
    int fold_lo[3], fold_in[3];
};

// Column statistics of the outputs a workgroup has just computed (the BatchNorm that follows every convolution of the dense
// model, reference uresnet_dense.py:45,57,68: batch statistics): lane (r, q) hands in, per column block, the sums over ITS
// rows; the lanes of a wave are combined with shuffles, the waves through LDS (free after the last MFMA), the workgroup
// adds 2 * ncols doubles into slot (workgroup % slots) of the slab.
template <int MAXCB>
__device__ __forceinline__ void dense_stats_flush(double (&d0)[MAXCB], double (&d1)[MAXCB], int cb_lo, int cb_hi, int ncols, int col_w0,
                                                  int cout, double *stats, int slots, unsigned wg, unsigned char *smem_raw)
{
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4, nthreads = blockDim.x;
    double *s_red = (double *)smem_raw;
    __syncthreads();                                       // every wave is done with the staged box / weight tiles
    for (int e = tid; e < 2 * ncols; e += nthreads) s_red[e] = 0.0;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < MAXCB; ++c) {
        if (c < cb_lo || c >= cb_hi) continue;
        double v0 = d0[c], v1 = d1[c];
        v0 += __shfl_xor(v0, 16); v1 += __shfl_xor(v1, 16);
        v0 += __shfl_xor(v0, 32); v1 += __shfl_xor(v1, 32);
        if (q == 0) { atomicAdd(&s_red[16 * c + r], v0); atomicAdd(&s_red[ncols + 16 * c + r], v1); }
    }
    __syncthreads();
    const long slot = wg % (unsigned)slots;
    for (int e = tid; e < 2 * ncols; e += nthreads) {
        const int which = e >= ncols ? 1 : 0, col = e - which * ncols;
        unsafeAtomicAdd(&stats[(slot * 2 + which) * cout + col_w0 + col], s_red[e]);
    }
}

// the producing layer's BatchNorm folded into the consumer's load (reference uresnet_dense.py:78-81: residual1's BatchNorm
// output feeds residual2's convolution with no ReLU between them): x * scale + shift per channel while the box is staged
__device__ __forceinline__ f32x4 dense_xf(f32x4 v, const float *__restrict__ sc, const float *__restrict__ sh, int c)
{
    if (sc) {
        const f32x4 a = *(const f32x4 *)(sc + c), b = *(const f32x4 *)(sh + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaf(v[k], a[k], b[k]);
    }
    return v;
}

__device__ __forceinline__ unsigned short f2bf(float f)
{
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}

template <int PREC>
__global__ __launch_bounds__(512) void k_dense_conv(DenseArgs g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int ES = PREC ? 2 : 4;                       // bytes per LDS element
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int KC = g.kc, CH = 16 * KC;
    const int rowb = CH * ES + 16;                         // LDS bytes per box voxel / weight column (+16: bank spread)
    const int nbox = g.box[0] * g.box[1] * g.box[2];
    unsigned char *s_box = smem_raw;
    unsigned char *s_w = smem_raw + (((long)nbox * rowb + 15) & ~15L);   // weight tiles of a group of taps

    // tile position in the sub-grid
    const int tiles_x = (g.Sub[2] + 15) / 16, tiles_y = (g.Sub[1] + g.TY - 1) / g.TY, tiles_z = (g.Sub[0] + g.TZ - 1) / g.TZ;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    const int tz = t % tiles_z; const int b = t / tiles_z;
    const int ux0 = tx * 16, uy0 = ty * g.TY, uz0 = tz * g.TZ;
    const int col_w0 = blockIdx.y * g.cw;                  // first output column of this workgroup
    const int ncols = min(g.cw, g.cout - col_w0);          // multiple of 16 (host pads)
    const int NCB = ncols / 16;
    const int wtile = ncols * rowb;
    // origin of the staged box in input coordinates
    const int iz0 = g.s[0] * uz0 + g.emin[0], iy0 = g.s[1] * uy0 + g.emin[1], ix0 = g.s[2] * ux0 + g.emin[2];
    const long in_rows_b = (long)b * g.In[0] * g.In[1] * g.In[2];

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the two row blocks of this wave (rb = 2 rp + i -> (ry, rz)) and its column blocks [cb_lo, cb_hi): 16 row blocks =
    // one pair per wave and all columns; 4 row blocks = two pairs, the column blocks dealt to the waves
    const int rp = g.NRB == 16 ? wave : (wave & 1);
    const int cb_lo = g.NRB == 16 ? 0 : (wave >> 1), cb_hi = g.NRB == 16 ? NCB : min(NCB, (wave >> 1) + 1);
    int ry[2], rz[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int rb = 2 * rp + i; ry[i] = rb % g.TY; rz[i] = rb / g.TY; }
    int rbase[2];   // box voxel of this lane's row (output u_x = ux0 + r) of either row block, before the tap offset
#pragma unroll
    for (int i = 0; i < 2; ++i) rbase[i] = (g.s[0] * rz[i] * g.box[1] + g.s[1] * ry[i]) * g.box[2] + g.s[2] * r;

    // per-tap constants (weight tap index, offset inside the box) come from host-built tables: lane t of a VGPR holds tap
    // t's, v_readlane fetches them (the first version decoded the tap with integer divisions and indexed the launch
    // record dynamically: ~3000 scalar instructions and 56 % wait cycles per wave for 54 MFMAs)
    const int ntaps = g.ntaps;
    const int v_tap_w = g.tap_w[lane < 27 ? lane : 26], v_tap_box = g.tap_box[lane < 27 ? lane : 26];
    auto stage_w = [&](int tap, int ch0, int slot) {
        const int wtap = __builtin_amdgcn_readlane(v_tap_w, tap);   // tile = wt[wtap][col_w0 + c][ch0 .. ch0 + CH)
        const float *src = g.wt + ((long)wtap * g.cout + col_w0) * g.cin + ch0;
        unsigned char *dst = s_w + slot * wtile;
        const int per_log = KC == 4 ? 4 : (KC == 2 ? 3 : 2);
        const int total = ncols << per_log;
        for (int base = tid; base < total; base += 4 * nthreads) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(base + u * nthreads, total - 1);
                v[u] = *(const f32x4 *)(src + (long)(e >> per_log) * g.cin + 4 * (e & ((1 << per_log) - 1)));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = base + u * nthreads;
                if (e >= total) continue;
                const int c = e >> per_log, k4 = e & ((1 << per_log) - 1);
                if constexpr (PREC) {
                    uint2 pk;
                    pk.x = f2bf(v[u][0]) | ((unsigned)f2bf(v[u][1]) << 16); pk.y = f2bf(v[u][2]) | ((unsigned)f2bf(v[u][3]) << 16);
                    *(uint2 *)(dst + c * rowb + 8 * k4) = pk;
                } else {
                    *(f32x4 *)(dst + c * rowb + 16 * k4) = v[u];
                }
            }
        }
    };

    // split contraction: this workgroup's slice of the channel chunks and of the taps (deep levels: few output tiles,
    // thousands of weights per output -- the slices are summed by k_dense_splitk_reduce in a fixed order)
    const int nchunks = g.cin / CH;
    const int zci = blockIdx.z / g.zt, zti = blockIdx.z - zci * g.zt;
    const int chunk_lo = nchunks * zci / g.zc, chunk_hi = nchunks * (zci + 1) / g.zc;
    const int tap_lo = ntaps * zti / g.zt, tap_hi = ntaps * (zti + 1) / g.zt;
    for (int ch0 = chunk_lo * CH; ch0 < chunk_hi * CH; ch0 += CH) {
        __syncthreads();                                   // the previous chunk's readers are done with the box
        // stage the input box: voxel (bz, by, bx) of the box = input (iz0 + bz, ...), clamped or zero.  One (bz, by) row of
        // the box per wave and pass (its index arithmetic is scalar), the lanes walk (bx, 4-channel piece): no integer
        // division per element.  (A flat index with four loads in flight per thread was measured slower at 128^3 x 16:
        // 389 vs 295 us; per-element div/mod by the box sizes: 478 us.)
        {
            const int per_log = KC == 4 ? 4 : (KC == 2 ? 3 : 2);          // float4 per voxel = 4 KC (KC in {1, 2, 4})
            const int per = 1 << per_log;
            const int nrows = g.box[0] * g.box[1], row_elems = g.box[2] << per_log;
            for (int rowi = wave; rowi < nrows; rowi += nwaves) {
                const int bz = rowi / g.box[1], by = rowi - bz * g.box[1];
                int iz = iz0 + bz, iy = iy0 + by;
                bool okr = true;
                if (g.mode == 0) { iz = min(max(iz, 0), g.In[0] - 1); iy = min(max(iy, 0), g.In[1] - 1); }
                else { okr = iz >= 0 && iz < g.In[0] && iy >= 0 && iy < g.In[1]; if (!okr) { iz = 0; iy = 0; } }
                const float *src_row = g.x + (in_rows_b + ((long)iz * g.In[1] + iy) * g.In[2]) * g.ldx + ch0;
                unsigned char *dst_row = s_box + (long)rowi * g.box[2] * rowb;
                for (int e = lane; e < row_elems; e += 64) {
                    const int bx = e >> per_log, k4 = e & (per - 1);
                    int ix = ix0 + bx;
                    bool ok = okr;
                    if (g.mode == 0) ix = min(max(ix, 0), g.In[2] - 1);
                    else if (ix < 0 || ix >= g.In[2]) { ok = false; ix = 0; }
                    f32x4 val = dense_xf(*(const f32x4 *)(src_row + (long)ix * g.ldx + 4 * k4), g.xf_scale, g.xf_shift, ch0 + 4 * k4);
                    if (!ok) val = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if constexpr (PREC) {
                        uint2 pk;
                        pk.x = f2bf(val[0]) | ((unsigned)f2bf(val[1]) << 16); pk.y = f2bf(val[2]) | ((unsigned)f2bf(val[3]) << 16);
                        *(uint2 *)(dst_row + bx * rowb + 8 * k4) = pk;
                    } else {
                        *(f32x4 *)(dst_row + bx * rowb + 16 * k4) = val;
                    }
                }
            }
        }
        // Weight tiles are staged per GROUP of taps (all 27 at once for the narrow layers: their tap loop then runs
        // without a barrier -- with one barrier per tap a 16-channel layer had 2 MFMAs per wave between barriers)
        for (int tap0 = tap_lo; tap0 < tap_hi; tap0 += g.tg) {
            const int tend = min(tap_hi, tap0 + g.tg);
            if (tap0 != tap_lo) __syncthreads();           // the previous group's readers are done with the tiles
            for (int tap = tap0; tap < tend; ++tap) stage_w(tap, ch0, tap - tap0);
            __syncthreads();                               // (also covers the box staged above)
            for (int tap = tap0; tap < tend; ++tap) {
                const int tbox = __builtin_amdgcn_readlane(v_tap_box, tap);
                const unsigned char *wb = s_w + (tap - tap0) * wtile;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    // row r of the block = output (uz0 + rz, uy0 + ry, ux0 + r): box voxel (s rz + dz, s ry + dy, s r + dx)
                    const unsigned char *arow = s_box + (long)(rbase[i] + tbox) * rowb;
                    for (int k = 0; k < KC; ++k) {
                        if constexpr (PREC) {
                            const s16x4 a = *(const s16x4 *)(arow + 32 * k + 8 * q);
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                if (c >= cb_lo && c < cb_hi) {
                                    const s16x4 bf = *(const s16x4 *)(wb + (16 * c + r) * rowb + 32 * k + 8 * q);
                                    acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, bf, acc[i][c], 0, 0, 0);
                                }
                            }
                        } else {
                            const f32x4 a = *(const f32x4 *)(arow + 64 * k + 16 * q);
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                if (c >= cb_lo && c < cb_hi) {
                                    const f32x4 bf = *(const f32x4 *)(wb + (16 * c + r) * rowb + 64 * k + 16 * q);
#pragma unroll
                                    for (int tt = 0; tt < 4; ++tt) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt], bf[tt], acc[i][c], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    // epilogue: D[row 4q + i][col r]; row = output u_x = ux0 + 4q + i of block (ry, rz)
    const bool split = g.slab != nullptr;
    const bool stats = g.stats != nullptr && !split;
    double d0[4] = {0.0, 0.0, 0.0, 0.0}, d1[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int uz = uz0 + rz[i], uy = uy0 + ry[i];
        if (uz >= g.Sub[0] || uy >= g.Sub[1]) continue;
        const int oz = g.p[0] + g.os[0] * uz, oy = g.p[1] + g.os[1] * uy;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < cb_lo || c >= cb_hi) continue;
            const int col = col_w0 + 16 * c + r;
            const float bv = (g.bias && !split) ? g.bias[col] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ux = ux0 + 4 * q + k;
                if (ux >= g.Sub[2]) continue;
                if (split) {
                    const long srow = (((long)b * g.Sub[0] + uz) * g.Sub[1] + uy) * g.Sub[2] + ux;
                    const long nsub = (long)g.B * g.Sub[0] * g.Sub[1] * g.Sub[2];
                    g.slab[((long)blockIdx.z * nsub + srow) * g.cout + col] = acc[i][c][k];
                } else {
                    const int ox = g.p[2] + g.os[2] * ux;
                    const long row = (long)b * g.Out[0] * g.Out[1] * g.Out[2] + ((long)oz * g.Out[1] + oy) * g.Out[2] + ox;
                    const float v = acc[i][c][k] + bv;
                    g.y[row * g.ldy + col] = v;
                    if (stats) { d0[c] += (double)v; d1[c] += (double)v * (double)v; }
                }
            }
        }
    }
    if (stats) dense_stats_flush<4>(d0, d1, cb_lo, cb_hi, ncols, col_w0, g.cout, g.stats, g.stat_slots, blockIdx.x, smem_raw);
}

// ---- fast path: 3 x 3 x 3 taps, unit strides, 16 row blocks (tile 16 x 4 x 4, box 18 x 6 x 6) ----------------------------
// The big levels of the dense model are k3 s1 convolutions (forward: e = j - 1, clamp; input gradient on the padded volume:
// e = -j, zero).  With the tile fixed, every LDS offset of the tap loop is a compile-time constant: the loop is 27 x (A
// fragments of the wave's two row blocks + NCB weight fragments + MFMAs) and nothing else.  The generic kernel spent
// ~1450 scalar and ~1900 vector instructions per wave on loop bookkeeping for 54 MFMAs at 16 channels -- the scalar unit of
// a CU alone was busy for half of the kernel's time (PMC: SQ_INSTS_SALU).  REV: tap j reads box offset 2 - j (the input
// gradient's e = -j) instead of j.  Weight tiles are staged one z-slice (9 taps) at a time.
// Registers decide how many workgroups share a CU (a workgroup is 2 waves per SIMD).  Without a bound hipcc took up to 256
// VGPRs for the wide chunks (one resident workgroup, nothing to hide its barriers behind): the launch bounds ask for 4 waves
// per SIMD (<= 128 VGPRs, 2 workgroups per CU) where the LDS image allows two workgroups.  Measured at 128^3 x 16 -> 16,
// bf16: 131 us at 94 VGPRs / 2 workgroups; forced to 80 VGPRs (3 workgroups) 174 us, to 64 (4 workgroups) slower still --
// the spills cost more than the occupancy gives.
template <int PREC, int KC, int NCB, int REV>
__global__ __launch_bounds__(512, (KC * NCB <= 4 && KC < 4) ? 4 : 2) void k_dense_conv3(DenseArgs g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int ES = PREC ? 2 : 4, CH = 16 * KC, ROWB = CH * ES + 16;
    constexpr int BX = 18, BY = 6, BZ = 6, NBOX = BX * BY * BZ;
    constexpr int NCOLS = 16 * NCB, WTILE = NCOLS * ROWB;
    constexpr int PER_LOG = KC == 4 ? 4 : (KC == 2 ? 3 : 2), PER = 1 << PER_LOG;
    unsigned char *s_box = smem_raw;
    unsigned char *s_w = smem_raw + ((NBOX * ROWB + 15) & ~15);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef URN_DENSE_STAMP
    long long *stamp = g.stamps ? g.stamps + (((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8 : nullptr;
#define URN_DSTAMP(i) do { if (stamp && lane == 0) stamp[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define URN_DSTAMP(i) do { } while (0)
#endif
    URN_DSTAMP(0);
    const int r = lane & 15, q = lane >> 4;
    const int tiles_x = (g.Sub[2] + 15) / 16, tiles_y = (g.Sub[1] + 3) / 4, tiles_z = (g.Sub[0] + 3) / 4;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    const int tz = t % tiles_z; const int b = t / tiles_z;
    const int ux0 = tx * 16, uy0 = ty * 4, uz0 = tz * 4;
    const int col_w0 = blockIdx.y * NCOLS;
    const int iz0 = uz0 + g.emin[0], iy0 = uy0 + g.emin[1], ix0 = ux0 + g.emin[2];
    const long in_rows_b = (long)b * g.In[0] * g.In[1] * g.In[2];
    f32x4 acc[2][NCB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < NCB; ++c) acc[i][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // wave w owns row blocks 2w, 2w+1: (ry, rz) = (rb & 3, rb >> 2); this lane's box row before the tap offset
    const unsigned char *abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int rb = 2 * wave + i; abase[i] = s_box + (((rb >> 2) * BY + (rb & 3)) * BX + r) * ROWB + (PREC ? 8 : 16) * q; }
    const unsigned char *bbase = s_w + r * ROWB + (PREC ? 8 : 16) * q;

    // split contraction (see k_dense_conv): blockIdx.z = chunk slice * zt + z-slice of the taps (zt = 1 or 3)
    const int nchunks = g.cin / CH;
    const int zci = blockIdx.z / g.zt, zti = blockIdx.z - zci * g.zt;
    const int chunk_lo = nchunks * zci / g.zc, chunk_hi = nchunks * (zci + 1) / g.zc;
    const int jz_lo = g.zt == 3 ? zti : 0, jz_hi = g.zt == 3 ? zti + 1 : 3;
    for (int ch0 = chunk_lo * CH; ch0 < chunk_hi * CH; ch0 += CH) {
        __syncthreads();
        // box: 36 (bz, by) rows x 18 voxels x PER 16-byte pieces, dealt flat over the 512 threads; every division below is by
        // a compile-time constant.  ALL loads of a thread are issued before the first conversion: one round trip per
        // chunk (the row-per-wave loop this replaces waited for each row's loads before it asked for the next row: five
        // dependent round trips per workgroup, 155 us for 128^3 x 16 -> 16 where the HBM traffic needs 45).
        {
            constexpr int ROW_E = BX * PER, TOT_E = BZ * BY * ROW_E, NIT = (TOT_E + 511) / 512;
            constexpr int HALF = NIT > 6 ? 6 : NIT;        // six 16-byte loads in flight per thread and batch (registers)
#pragma unroll
            for (int h0 = 0; h0 < NIT; h0 += HALF) {
                f32x4 val[HALF];
                bool okv[HALF];
#pragma unroll
                for (int it = 0; it < HALF; ++it) {
                    const int f = (h0 + it) * 512 + tid;
                    okv[it] = false;
                    val[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (h0 + it < NIT && f < TOT_E) {
                        const int rowi = f / ROW_E, e = f - rowi * ROW_E;
                        const int bz = rowi / BY, by = rowi - bz * BY;
                        const int bx = e >> PER_LOG, k4 = e & (PER - 1);
                        int iz = iz0 + bz, iy = iy0 + by, ix = ix0 + bx;
                        bool ok = true;
                        if (g.mode == 0) {
                            iz = min(max(iz, 0), g.In[0] - 1); iy = min(max(iy, 0), g.In[1] - 1); ix = min(max(ix, 0), g.In[2] - 1);
                        } else {
                            ok = iz >= 0 && iz < g.In[0] && iy >= 0 && iy < g.In[1] && ix >= 0 && ix < g.In[2];
                            if (!ok) { iz = 0; iy = 0; ix = 0; }
                        }
                        okv[it] = ok;
                        val[it] = *(const f32x4 *)(g.x + (in_rows_b + ((long)iz * g.In[1] + iy) * g.In[2] + ix) * g.ldx + ch0 + 4 * k4);
                    }
                }
#pragma unroll
                for (int it = 0; it < HALF; ++it) {
                    const int f = (h0 + it) * 512 + tid;
                    if (h0 + it < NIT && f < TOT_E) {
                        const int rowi = f / ROW_E, e = f - rowi * ROW_E;
                        const int bx = e >> PER_LOG, k4 = e & (PER - 1);
                        f32x4 v = dense_xf(val[it], g.xf_scale, g.xf_shift, ch0 + 4 * k4);
                        if (!okv[it]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                        unsigned char *dst = s_box + (rowi * BX + bx) * ROWB;
                        if constexpr (PREC) {
                            uint2 pk;
                            pk.x = f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16); pk.y = f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                            *(uint2 *)(dst + 8 * k4) = pk;
                        } else {
                            *(f32x4 *)(dst + 16 * k4) = v;
                        }
                    }
                }
            }
        }
        URN_DSTAMP(1);
#pragma unroll
        for (int jz = 0; jz < 3; ++jz) {
            if (jz < jz_lo || jz >= jz_hi) continue;       // workgroup-uniform
            if (jz > jz_lo) __syncthreads();
            // the 9 weight tiles of this z-slice: wt[(jz * 3 + jy) * 3 + jx][col_w0 + c][ch0 ..]; one flat loop
            {
                const float *src = g.wt + ((long)(jz * 9) * g.cout + col_w0) * g.cin + ch0;
                // (a plain loop waits for every load before it asks for the next: nine dependent round trips to HBM per slice
                // at 32 x 64 -- the weights of a deep level are read exactly once; batches of WB loads per thread instead)
                constexpr int TOTAL = 9 * NCOLS * PER, NITW = (TOTAL + 511) / 512, WB = NITW > 5 ? 5 : NITW;
#pragma unroll
                for (int h0 = 0; h0 < NITW; h0 += WB) {
                    f32x4 wv[WB];
#pragma unroll
                    for (int it = 0; it < WB; ++it) {
                        const int e = (h0 + it) * 512 + tid;
                        wv[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (h0 + it < NITW && e < TOTAL) {
                            const int tap = e / (NCOLS * PER), rem = e - tap * (NCOLS * PER);
                            const int c = rem >> PER_LOG, k4 = rem & (PER - 1);
                            wv[it] = *(const f32x4 *)(src + ((long)tap * g.cout + c) * g.cin + 4 * k4);
                        }
                    }
#pragma unroll
                    for (int it = 0; it < WB; ++it) {
                        const int e = (h0 + it) * 512 + tid;
                        if (h0 + it < NITW && e < TOTAL) {
                            const int tap = e / (NCOLS * PER), rem = e - tap * (NCOLS * PER);
                            const int c = rem >> PER_LOG, k4 = rem & (PER - 1);
                            const f32x4 v = wv[it];
                            if constexpr (PREC) {
                                uint2 pk;
                                pk.x = f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16); pk.y = f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                                *(uint2 *)(s_w + tap * WTILE + c * ROWB + 8 * k4) = pk;
                            } else {
                                *(f32x4 *)(s_w + tap * WTILE + c * ROWB + 16 * k4) = v;
                            }
                        }
                    }
                }
            }
            __syncthreads();
            if (jz == 0) URN_DSTAMP(2);
#pragma unroll
            for (int jy = 0; jy < 3; ++jy)
#pragma unroll
                for (int jx = 0; jx < 3; ++jx) {
                    constexpr int dummy = 0; (void)dummy;
                    const int boxoff = ((REV ? 2 - jz : jz) * BY + (REV ? 2 - jy : jy)) * BX + (REV ? 2 - jx : jx);
                    const int wslot = jy * 3 + jx;
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        if constexpr (PREC) {
                            s16x4 a[2], bf[NCB];
#pragma unroll
                            for (int i = 0; i < 2; ++i) a[i] = *(const s16x4 *)(abase[i] + boxoff * ROWB + 32 * k);
#pragma unroll
                            for (int c = 0; c < NCB; ++c) bf[c] = *(const s16x4 *)(bbase + wslot * WTILE + 16 * c * ROWB + 32 * k);
#pragma unroll
                            for (int i = 0; i < 2; ++i)
#pragma unroll
                                for (int c = 0; c < NCB; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[i], bf[c], acc[i][c], 0, 0, 0);
                        } else {
                            f32x4 a[2], bf[NCB];
#pragma unroll
                            for (int i = 0; i < 2; ++i) a[i] = *(const f32x4 *)(abase[i] + boxoff * ROWB + 64 * k);
#pragma unroll
                            for (int c = 0; c < NCB; ++c) bf[c] = *(const f32x4 *)(bbase + wslot * WTILE + 16 * c * ROWB + 64 * k);
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                                for (int i = 0; i < 2; ++i)
#pragma unroll
                                    for (int c = 0; c < NCB; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][tt], bf[c][tt], acc[i][c], 0, 0, 0);
                        }
                    }
                }
        }
    }
    URN_DSTAMP(3);
    const bool split = g.slab != nullptr;
    const bool stats = g.stats != nullptr && !split;
    double d0[NCB], d1[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c) { d0[c] = 0.0; d1[c] = 0.0; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rb = 2 * wave + i;
        const int uz = uz0 + (rb >> 2), uy = uy0 + (rb & 3);
        if (uz >= g.Sub[0] || uy >= g.Sub[1]) continue;
        const long rowz = (long)b * g.Out[0] * g.Out[1] * g.Out[2] + ((long)(g.p[0] + uz) * g.Out[1] + (g.p[1] + uy)) * g.Out[2] + g.p[2];
        const long nsub = (long)g.B * g.Sub[0] * g.Sub[1] * g.Sub[2];
        const long srow0 = (((long)b * g.Sub[0] + uz) * g.Sub[1] + uy) * g.Sub[2];
        // folded un-padding: is this row block inside the un-padded volume in z and y, and where does its row start there
        const int fz = g.p[0] + uz - g.fold_lo[0], fy = g.p[1] + uy - g.fold_lo[1];
        const bool fold_zy = g.fold_y && (unsigned)fz < (unsigned)g.fold_in[0] && (unsigned)fy < (unsigned)g.fold_in[1];
        const long frow = (((long)b * g.fold_in[0] + fz) * g.fold_in[1] + fy) * g.fold_in[2] + (g.p[2] - g.fold_lo[2]);
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const int col = col_w0 + 16 * c + r;
            const float bv = (g.bias && !split) ? g.bias[col] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ux = ux0 + 4 * q + k;
                if (ux >= g.Sub[2]) continue;
                if (split) g.slab[((long)blockIdx.z * nsub + srow0 + ux) * g.cout + col] = acc[i][c][k];
                else {
                    const float v = acc[i][c][k] + bv;
                    if (fold_zy && (unsigned)(g.p[2] + ux - g.fold_lo[2]) < (unsigned)g.fold_in[2]) g.fold_y[(frow + ux) * g.ldy + col] = v;
                    else g.y[(rowz + ux) * g.ldy + col] = v;
                    if (stats) { d0[c] += (double)v; d1[c] += (double)v * (double)v; }
                }
            }
        }
    }
    URN_DSTAMP(4);
    if (stats) dense_stats_flush<NCB>(d0, d1, 0, NCB, NCOLS, col_w0, g.cout, g.stats, g.stat_slots, blockIdx.x, smem_raw);
}

// y[out row of sub-grid row][col] = bias + slab[0] + slab[1] + ... (fixed order)
// With g.stats the launch is grid-strided (256 % (cout / 4) == 0: a thread keeps its 4 columns), the column sums of y stay in
// registers (fp64), are combined per workgroup through LDS and added to the slab like dense_stats_flush does.
__global__ __launch_bounds__(256) void k_dense_splitk_reduce(DenseArgs g, int Z)
{
    __shared__ double s_red[256 * 8];
    const int c4 = g.cout / 4;
    const long nsub = (long)g.B * g.Sub[0] * g.Sub[1] * g.Sub[2];
    const long total = nsub * c4;
    const bool stats = g.stats != nullptr;
    double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int k4 = (int)(e % c4);
        long v = e / c4;
        const long srow = v;
        const int ux = (int)(v % g.Sub[2]); v /= g.Sub[2];
        const int uy = (int)(v % g.Sub[1]); v /= g.Sub[1];
        const int uz = (int)(v % g.Sub[0]); const int b = (int)(v / g.Sub[0]);
        f32x4 s = g.bias ? *(const f32x4 *)(g.bias + 4 * k4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        {   // the slices are added in order, eight loads in flight (a plain loop waits for every load: Z round trips)
            const float *p = g.slab + srow * g.cout + 4 * k4;
            const long zs = nsub * g.cout;
            int z = 0;
            for (; z + 8 <= Z; z += 8) {
                f32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *(const f32x4 *)(p + (long)(z + k) * zs);
#pragma unroll
                for (int k = 0; k < 8; ++k) s += v[k];
            }
            if (z + 4 <= Z) {
                f32x4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = *(const f32x4 *)(p + (long)(z + k) * zs);
#pragma unroll
                for (int k = 0; k < 4; ++k) s += v[k];
                z += 4;
            }
            for (; z < Z; ++z) s += *(const f32x4 *)(p + (long)z * zs);
        }
        const long row = (long)b * g.Out[0] * g.Out[1] * g.Out[2] +
                         ((long)(g.p[0] + g.os[0] * uz) * g.Out[1] + (g.p[1] + g.os[1] * uy)) * g.Out[2] + (g.p[2] + g.os[2] * ux);
        float *dst = g.y + row * g.ldy + 4 * k4;
        dst[0] = s[0]; dst[1] = s[1]; dst[2] = s[2]; dst[3] = s[3];
        if (stats) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[k] += (double)s[k]; a[4 + k] += (double)s[k] * (double)s[k]; }
        }
    }
    if (!stats) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) s_red[k * 256 + threadIdx.x] = a[k];
    __syncthreads();
    if ((int)threadIdx.x < c4) {      // thread t < c4 owns columns 4t..4t+3: its peers are t + j * c4
        double r8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int j = threadIdx.x; j < 256; j += c4)
#pragma unroll
            for (int k = 0; k < 8; ++k) r8[k] += s_red[k * 256 + j];
        const long slot = blockIdx.x % (unsigned)g.stat_slots;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsafeAtomicAdd(&g.stats[(slot * 2 + 0) * g.cout + 4 * threadIdx.x + k], r8[k]);
            unsafeAtomicAdd(&g.stats[(slot * 2 + 1) * g.cout + 4 * threadIdx.x + k], r8[4 + k]);
        }
    }
}

int g_dense_strided_nrb = 16;          // row blocks per workgroup of the big strided launches (urn_set_option "dense_strided_nrb": 4 or 16; small launches drop to 4 below): cfg2 step 19.4 -> 19.2 ms
long long *g_dense_stamps = nullptr;   // diagnostics (urn_set_option "dense_stamp_ptr"; kernels built with -DURN_DENSE_STAMP)

// split factor of a launch: only when the output tiles alone leave most CUs idle
static void dense_plan_split(long wgs, int nchunks, int ntaps, int &zc, int &zt)
{
    zc = zt = 1;
    if (wgs >= 384) return;
    long want = (1024 + wgs - 1) / wgs;       // 64-column workgroups of 8 waves stay (their weight tiles are staged by all
    if (want > 32) want = 32;                  // threads); the contraction is what gets dealt out
    zt = (int)(want < ntaps ? want : ntaps);
    if (zt < 1) zt = 1;
    long rest = want / zt;
    zc = (int)(rest < nchunks ? rest : nchunks);
    if (zc < 1) zc = 1;
}

extern "C" int64_t urn_dense_conv_scratch_bytes(int cout, int batch, const urn_dense_geom *gm)
{
    if (!gm || cout <= 0 || batch <= 0) return -1;
    const long nsub = (long)batch * gm->Sub[0] * gm->Sub[1] * gm->Sub[2];
    // split only happens for launches with fewer than 256 workgroups of >= 64 outputs x 16 columns
    if (nsub * cout / (64 * 16) >= 4096) return 256;
    return 32 * nsub * cout * 4 + 256;
}

// armed by urn_dense_conv_dgrad_fold around its urn_dense_conv call: the un-padding target of an input gradient; `used` is set
// when the launch took it (3 x 3 x 3 fast path without a split contraction)
static thread_local struct { bool armed, used; float *y; int lo[3], in[3]; } g_fold = {false, false, nullptr, {0, 0, 0}, {0, 0, 0}};

extern "C" int urn_dense_conv(const float *x, int64_t ldx, int cin, const float *wt, const float *bias, float *y, int64_t ldy,
                              int cout, int batch, const urn_dense_geom *gm, int precision, double *stats, int stat_slots,
                              const float *xf_scale, const float *xf_shift, void *scratch, int64_t scratch_bytes, void *stream)
{
    URN_CHECK_ARG((xf_scale == nullptr) == (xf_shift == nullptr), "scale and shift go together");
    URN_CHECK_ARG(x && wt && y && gm, "null pointer");
    URN_CHECK_ARG(!stats || (stat_slots > 0 && 256 % (cout / 4) == 0), "statistics: slots > 0 and cout / 4 a divisor of 256");
    URN_CHECK_ARG(cin > 0 && cout > 0 && cin % 16 == 0 && cout % 16 == 0 && batch > 0, "channel counts must be multiples of 16");
    URN_CHECK_ARG(ldx >= cin && ldy >= cout && ldx % 4 == 0, "row strides");
    DenseArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.wt = wt; a.bias = bias; a.y = y; a.ldx = (long)ldx; a.ldy = (long)ldy; a.cin = cin; a.cout = cout; a.B = batch;
    a.mode = gm->mode; a.stats = stats; a.stat_slots = stat_slots; a.stamps = g_dense_stamps; a.xf_scale = xf_scale; a.xf_shift = xf_shift;
    for (int d = 0; d < 3; ++d) {
        a.In[d] = gm->In[d]; a.Out[d] = gm->Out[d]; a.Sub[d] = gm->Sub[d]; a.p[d] = gm->p[d]; a.os[d] = gm->os[d]; a.s[d] = gm->s[d];
        a.nt[d] = gm->nt[d]; a.kdim[d] = gm->kdim[d];
        URN_CHECK_ARG(a.nt[d] >= 1 && a.nt[d] <= 3 && a.Sub[d] >= 0 && a.s[d] >= 1 && a.os[d] >= 1, "bad geometry");
        int emin = 1 << 30, emax = -(1 << 30);
        for (int j = 0; j < a.nt[d]; ++j) {
            a.e[d][j] = gm->e[d][j]; a.wi[d][j] = gm->wi[d][j];
            emin = a.e[d][j] < emin ? a.e[d][j] : emin; emax = a.e[d][j] > emax ? a.e[d][j] : emax;
        }
        a.emin[d] = emin;
        a.box[d] = emax - emin + 1;   // + s * (tile - 1) below
    }
    if (a.Sub[0] == 0 || a.Sub[1] == 0 || a.Sub[2] == 0) return URN_OK;
    // Tile: 16 row blocks (256 outputs) x 64 columns for the big levels; the deep levels have few voxels and many channels
    // (16^3 x 128: 16 such tiles for 256 CUs), so the tile shrinks -- 4 row blocks, then 32 and 16 columns -- until the
    // launch has ~512 workgroups.  4 row blocks also when the input is read with a stride (the staged box grows with it).
    const bool strided = a.s[0] > 1 || a.s[1] > 1 || a.s[2] > 1;
    auto tiles_for = [&](int nrb, int &TY, int &TZ) {
        TY = a.Sub[0] > 1 ? (nrb == 16 ? 4 : 2) : nrb;
        if (a.Sub[1] < TY) { TY = 1; while (TY * 2 <= a.Sub[1] && TY < nrb) TY *= 2; }
        TZ = nrb / TY;
        return (long)((a.Sub[2] + 15) / 16) * ((a.Sub[1] + TY - 1) / TY) * ((a.Sub[0] + TZ - 1) / TZ) * batch;
    };
    a.NRB = 16; a.cw = 64;
    if (strided) {
        // 16 row blocks also for a strided input when the staged box (it grows with the stride) of ONE 16-channel chunk still fits
        int ty, tz;
        (void)tiles_for(16, ty, tz);
        const long nbox16 = (long)(a.box[0] + a.s[0] * (tz - 1)) * (a.box[1] + a.s[1] * (ty - 1)) * (a.box[2] + a.s[2] * 15);
        const long rowb1 = 16L * (precision ? 2 : 4) + 16;
        if (g_dense_strided_nrb != 16 || ((nbox16 * rowb1 + 15) & ~15L) + (cout < 64 ? cout : 64) * rowb1 > 150L * 1024) a.NRB = 4;
    }
    {
        int ty, tz;
        bool k333 = !strided;
        for (int d = 0; d < 3; ++d) k333 = k333 && a.nt[d] == 3 && a.os[d] == 1 && a.Sub[d] >= 4;
        // (the 3 x 3 x 3 fast path keeps its 256-output tile and splits the contraction instead; everything else shrinks the tile)
        if (a.NRB == 16 && !k333 && tiles_for(16, ty, tz) * ((cout + 63) / 64) < 512) a.NRB = 4;
        (void)tiles_for(a.NRB, a.TY, a.TZ);
    }
    a.box[2] += a.s[2] * 15; a.box[1] += a.s[1] * (a.TY - 1); a.box[0] += a.s[0] * (a.TZ - 1);
    const long nbox = (long)a.box[0] * a.box[1] * a.box[2];
    const int es = precision ? 2 : 4;
    // channels per chunk (kc) and taps per weight group (tg): the box plus the group's weight tiles, preferably under
    // 78 KB (two workgroups per CU), at most 150 KB
    const int ntaps = a.nt[0] * a.nt[1] * a.nt[2];
    const long ncols = cout < a.cw ? cout : a.cw;
    int kc = 0, tg = 0;
    for (long budget : {78L * 1024, 150L * 1024}) {
        for (int k = 4; k >= 1 && !kc; k >>= 1) {
            if ((cin / 16) % k) continue;
            const long rowb = 16L * k * es + 16;
            const long boxb = (nbox * rowb + 15) & ~15L;
            if (boxb + ncols * rowb > budget) continue;
            long t = (budget - boxb) / (ncols * rowb);
            kc = k; tg = (int)(t < ntaps ? t : ntaps);
        }
        if (kc) break;
    }
    if (!kc) { urn_set_error("urn_dense_conv: input box of %ld voxels does not fit LDS", nbox); return URN_EUNSUPPORTED; }
    a.kc = kc; a.tg = tg; a.ntaps = ntaps;
    for (int tap = 0; tap < ntaps; ++tap) {
        const int jx = tap % a.nt[2], jy = (tap / a.nt[2]) % a.nt[1], jz = tap / (a.nt[2] * a.nt[1]);
        a.tap_w[tap] = (a.wi[0][jz] * a.kdim[1] + a.wi[1][jy]) * a.kdim[2] + a.wi[2][jx];
        a.tap_box[tap] = ((a.e[0][jz] - a.emin[0]) * a.box[1] + (a.e[1][jy] - a.emin[1])) * a.box[2] + (a.e[2][jx] - a.emin[2]);
    }
    const long rowb = 16L * kc * es + 16;
    const size_t lds = (size_t)(((nbox * rowb + 15) & ~15L) + (long)tg * ncols * rowb);
    const long tiles = (long)((a.Sub[2] + 15) / 16) * ((a.Sub[1] + a.TY - 1) / a.TY) * ((a.Sub[0] + a.TZ - 1) / a.TZ) * batch;
    // waves: 16 row blocks = 8 waves (a pair of row blocks each, all column blocks); 4 row blocks = 2 waves per column block
    const int gy = (cout + a.cw - 1) / a.cw;
    dense_plan_split(tiles * gy, cin / (16 * kc), ntaps, a.zc, a.zt);
    const int Z = a.zc * a.zt;
    const long nsub = (long)batch * a.Sub[0] * a.Sub[1] * a.Sub[2];
    // the split reduce: one element per thread, or (statistics) grid-strided with at most 1024 workgroups
    long reduce_blocks = urn_cdiv(nsub * (cout / 4), 256);
    if (stats && reduce_blocks > 1024) reduce_blocks = 1024;
    if (Z > 1 && (!scratch || scratch_bytes < (int64_t)Z * nsub * cout * 4)) { a.zc = a.zt = 1; }   // no scratch: unsplit
    a.slab = a.zc * a.zt > 1 ? (float *)scratch : nullptr;
    const dim3 grid((unsigned)tiles, gy, a.zc * a.zt), block(a.NRB == 16 ? 512 : 128 * (int)(ncols / 16));
    hipStream_t st = (hipStream_t)stream;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)k_dense_conv<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_dense_conv<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    // fast path: full 3 x 3 x 3 taps, unit strides, forward order (e = j + emin) or reversed (e = -j), 16 row blocks of
    // 4 x 4, no split, a channel chunk whose box + 9 weight tiles fit
    {
        bool f3 = a.NRB == 16 && a.TY == 4 && a.TZ == 4 && ntaps == 27;
        int rev = -1;
        for (int d = 0; d < 3 && f3; ++d) {
            f3 = a.nt[d] == 3 && a.s[d] == 1 && a.os[d] == 1 && a.kdim[d] == 3;
            const bool fw = a.e[d][0] + 1 == a.e[d][1] && a.e[d][1] + 1 == a.e[d][2] && a.wi[d][0] == 0 && a.wi[d][1] == 1 && a.wi[d][2] == 2;
            const bool rv = a.e[d][0] - 1 == a.e[d][1] && a.e[d][1] - 1 == a.e[d][2] && a.wi[d][0] == 0 && a.wi[d][1] == 1 && a.wi[d][2] == 2;
            if (!fw && !rv) f3 = false;
            const int rd = rv ? 1 : 0;
            if (rev >= 0 && rev != rd) f3 = false;
            rev = rd;
        }
        if (f3) {
            int kc3 = 0, ncb = cout >= 64 ? 4 : cout / 16;
            if (cout % (16 * ncb)) ncb = 1;
            for (int k = 4; k >= 1 && !kc3; k >>= 1) {
                if ((cin / 16) % k) continue;
                const long rb3 = 16L * k * es + 16;
                if (((648 * rb3 + 15) & ~15L) + 9L * 16 * ncb * rb3 <= 150 * 1024) kc3 = k;
            }
            if (kc3) {
                const long rb3 = 16L * kc3 * es + 16;
                const size_t lds3 = (size_t)(((648 * rb3 + 15) & ~15L) + 9L * 16 * ncb * rb3);
                // split for the fast path: z-slices of the taps (3) x chunk slices
                const int gy3 = cout / (16 * ncb), nch3 = cin / (16 * kc3);
                a.zc = a.zt = 1;
                if (tiles * gy3 < 384 && scratch) {
                    long want = (1024 + tiles * gy3 - 1) / (tiles * gy3);
                    a.zt = want >= 3 ? 3 : 1;
                    want = (want + a.zt - 1) / a.zt;
                    a.zc = (int)(want < nch3 ? want : nch3);
                    if (a.zc < 1) a.zc = 1;
                    if (scratch_bytes < (int64_t)a.zc * a.zt * nsub * cout * 4) a.zc = a.zt = 1;
                }
                a.slab = a.zc * a.zt > 1 ? (float *)scratch : nullptr;
                if (g_fold.armed && !a.slab && rev == 1) {
                    a.fold_y = g_fold.y;
                    for (int d = 0; d < 3; ++d) { a.fold_lo[d] = g_fold.lo[d]; a.fold_in[d] = g_fold.in[d]; }
                    g_fold.used = true;
                }
                const dim3 grid3((unsigned)tiles, gy3, a.zc * a.zt);
                // (one flag per instantiation: the attribute call is a host round trip into the runtime, ~100 of them per dense step otherwise)
#define URN_D3(P, K, N, R) if (precision == P && kc3 == K && ncb == N && rev == R) { \
                    static bool attr_done = false; \
                    if (!attr_done) { (void)hipFuncSetAttribute((const void *)k_dense_conv3<P, K, N, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_done = true; } \
                    hipLaunchKernelGGL((k_dense_conv3<P, K, N, R>), grid3, dim3(512), lds3, st, a); \
                    if (a.slab) hipLaunchKernelGGL(k_dense_splitk_reduce, dim3(reduce_blocks), dim3(256), 0, st, a, a.zc * a.zt); \
                    URN_LAUNCH_CHECK(); return URN_OK; }
#define URN_D3K(P, R) URN_D3(P, 1, 1, R) URN_D3(P, 1, 2, R) URN_D3(P, 1, 3, R) URN_D3(P, 1, 4, R) URN_D3(P, 2, 1, R) URN_D3(P, 2, 2, R) URN_D3(P, 2, 3, R) URN_D3(P, 2, 4, R) \
                    URN_D3(P, 4, 1, R) URN_D3(P, 4, 2, R) URN_D3(P, 4, 3, R) URN_D3(P, 4, 4, R)
                URN_D3K(0, 0) URN_D3K(0, 1) URN_D3K(1, 0) URN_D3K(1, 1)
#undef URN_D3K
#undef URN_D3
            }
        }
    }
    if (precision) hipLaunchKernelGGL(k_dense_conv<1>, grid, block, lds, st, a);
    else hipLaunchKernelGGL(k_dense_conv<0>, grid, block, lds, st, a);
    if (a.slab)
        hipLaunchKernelGGL(k_dense_splitk_reduce, dim3(reduce_blocks), dim3(256), 0, st, a, a.zc * a.zt);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

// Gradient of the replicate padding: dx[i] = sum of dxp over the padded positions that clamp to i (per dimension the
// border voxels take their own position plus the pad_lo / pad_hi positions beyond it).  dxp rows: padded volume
// (In + lo + hi per dim), dx rows: the volume itself; c channels, dense rows.
__global__ void k_dense_fold(const float *__restrict__ dxp, float *__restrict__ dx, int B, int Z, int Y, int X, int lz, int hz, int ly,
                             int hy, int lx, int hx, int c)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = c / 4;
    const long total = (long)B * Z * Y * X * c4;
    if (e >= total) return;
    const int k4 = (int)(e % c4);
    long v = e / c4;
    const int x = (int)(v % X); v /= X;
    const int y = (int)(v % Y); v /= Y;
    const int z = (int)(v % Z); const int b = (int)(v / Z);
    const int PZ = Z + lz + hz, PY = Y + ly + hy, PX = X + lx + hx;
    const int z0 = z == 0 ? 0 : z + lz, z1 = z == Z - 1 ? PZ - 1 : z + lz;
    const int y0 = y == 0 ? 0 : y + ly, y1 = y == Y - 1 ? PY - 1 : y + ly;
    const int x0 = x == 0 ? 0 : x + lx, x1 = x == X - 1 ? PX - 1 : x + lx;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int pz = z0; pz <= z1; ++pz)
        for (int py = y0; py <= y1; ++py)
            for (int px = x0; px <= x1; ++px)
                s += *(const f32x4 *)(dxp + ((((long)b * PZ + pz) * PY + py) * PX + px) * c + 4 * k4);
    *(f32x4 *)(dx + ((((long)b * Z + z) * Y + y) * X + x) * c + 4 * k4) = s;
}

// the same for a dx whose voxels already hold their OWN padded position (written by the input-gradient launch itself,
// DenseArgs.fold_y): only the boundary voxels do anything -- they add the shell positions that clamp to them
__global__ void k_dense_fold_border(const float *__restrict__ dxp, float *__restrict__ dx, int B, int Z, int Y, int X, int lz, int hz,
                                    int ly, int hy, int lx, int hx, int c)
{
    // threads = the voxels of the boundary only (128^3: 97k of 2.1M; a thread per voxel of the volume that leaves at once was
    // bound by the launch rate of its 32,768 workgroups: 36 us), in three disjoint groups: the two z faces | the y faces of
    // the inner z planes | the x faces of the inner (z, y) rows
    const int c4 = c / 4;
    const int nzf = Z < 2 ? Z : 2, nyf = Y < 2 ? Y : 2, nxf = X < 2 ? X : 2, Zi = Z > 2 ? Z - 2 : 0, Yi = Y > 2 ? Y - 2 : 0;
    const long n1 = (long)nzf * Y * X, n2 = (long)Zi * nyf * X, n3 = (long)Zi * Yi * nxf, nb = n1 + n2 + n3;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)B * nb * c4) return;
    const int k4 = (int)(e % c4);
    const long v = e / c4;
    const int b = (int)(v / nb);
    long w = v - (long)b * nb;
    int x, y, z;
    if (w < n1) {
        const int zf = (int)(w / ((long)Y * X)); const int rem = (int)(w - (long)zf * Y * X);
        z = zf == 0 ? 0 : Z - 1; y = rem / X; x = rem - y * X;
    } else if (w < n1 + n2) {
        w -= n1;
        const int zi = (int)(w / (nyf * X)); const int rem = (int)(w - (long)zi * nyf * X);
        const int yf = rem / X;
        z = 1 + zi; y = yf == 0 ? 0 : Y - 1; x = rem - yf * X;
    } else {
        w -= n1 + n2;
        const int zi = (int)(w / (Yi * nxf)); const int rem = (int)(w - (long)zi * Yi * nxf);
        const int yi = rem / nxf, xf = rem - yi * nxf;
        z = 1 + zi; y = 1 + yi; x = xf == 0 ? 0 : X - 1;
    }
    const int PZ = Z + lz + hz, PY = Y + ly + hy, PX = X + lx + hx;
    const int z0 = z == 0 ? 0 : z + lz, z1 = z == Z - 1 ? PZ - 1 : z + lz;
    const int y0 = y == 0 ? 0 : y + ly, y1 = y == Y - 1 ? PY - 1 : y + ly;
    const int x0 = x == 0 ? 0 : x + lx, x1 = x == X - 1 ? PX - 1 : x + lx;
    if (z0 == z1 && y0 == y1 && x0 == x1) return;          // nothing clamps to it
    float *own = dx + ((((long)b * Z + z) * Y + y) * X + x) * c + 4 * k4;
    f32x4 s = *(const f32x4 *)own;
    for (int pz = z0; pz <= z1; ++pz)
        for (int py = y0; py <= y1; ++py)
            for (int px = x0; px <= x1; ++px)
                if (pz != z + lz || py != y + ly || px != x + lx)
                    s += *(const f32x4 *)(dxp + ((((long)b * PZ + pz) * PY + py) * PX + px) * c + 4 * k4);
    *(f32x4 *)own = s;
}

extern "C" int urn_dense_fold(const float *dxp, float *dx, int batch, const int *dims /* Z, Y, X */, const int *pad_lo,
                              const int *pad_hi, int c, void *stream)
{
    URN_CHECK_ARG(dxp && dx && dims && pad_lo && pad_hi && c > 0 && c % 4 == 0 && batch > 0, "bad argument");
    const long total = (long)batch * dims[0] * dims[1] * dims[2] * (c / 4);
    if (total == 0) return URN_OK;
    hipLaunchKernelGGL(k_dense_fold, dim3(urn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dxp, dx, batch, dims[0], dims[1],
                       dims[2], pad_lo[0], pad_hi[0], pad_lo[1], pad_hi[1], pad_lo[2], pad_hi[2], c);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

// Input gradient of a stride-1 convolution on the replicate-padded volume AND its un-padding: urn_dense_conv with the
// geometry of the padded volume (gm) into dxp, then urn_dense_fold into dx -- but when the launch is the 3 x 3 x 3 fast
// path the kernel writes the voxels inside the volume straight to dx and only the shell to dxp, and the fold touches the
// boundary voxels only (128^3 x 16: 55 us of fold -> 8).  Same results, same order of the sums at the boundary voxels
// except that a voxel's own position comes first.
extern "C" int urn_dense_conv_dgrad_fold(const float *dy, int64_t ld_dy, int cout, const float *wb, float *dxp, float *dx, int64_t ld_dx,
                                         int cin, int batch, const urn_dense_geom *gm, const int *dims, const int *pad_lo,
                                         const int *pad_hi, int precision, void *scratch, int64_t scratch_bytes, void *stream)
{
    URN_CHECK_ARG(dy && wb && dxp && dx && gm && dims && pad_lo && pad_hi, "null pointer");
    g_fold.armed = true; g_fold.used = false; g_fold.y = dx;
    for (int d = 0; d < 3; ++d) { g_fold.lo[d] = pad_lo[d]; g_fold.in[d] = dims[d]; }
    const int r = urn_dense_conv(dy, ld_dy, cout, wb, nullptr, dxp, ld_dx, cin, batch, gm, precision, nullptr, 0, nullptr, nullptr, scratch,
                                 scratch_bytes, stream);
    g_fold.armed = false;
    if (r != URN_OK) return r;
    if (!g_fold.used) return urn_dense_fold(dxp, dx, batch, dims, pad_lo, pad_hi, cin, stream);
    URN_CHECK_ARG(ld_dx == cin, "dense rows");
    const long Zd = dims[0], Yd = dims[1], Xd = dims[2];
    const long nbnd = (Zd < 2 ? Zd : 2) * Yd * Xd + (Zd > 2 ? Zd - 2 : 0) * ((Yd < 2 ? Yd : 2) * Xd + (Yd > 2 ? Yd - 2 : 0) * (Xd < 2 ? Xd : 2));
    const long total = (long)batch * nbnd * (cin / 4);
    hipLaunchKernelGGL(k_dense_fold_border, dim3(urn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float *)dxp, dx, batch, dims[0],
                       dims[1], dims[2], pad_lo[0], pad_hi[0], pad_lo[1], pad_hi[1], pad_lo[2], pad_hi[2], cin);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dw[tap][ci][co] = sum over outputs o of x[in(o, tap)][ci] * dy[o][co], in(o, tap) = s * o + tap - lo (clamped: mode 0,
// or skipped when out of range: mode 1).  The OUTPUT VOXELS are the contraction index of the MFMAs.  A workgroup walks a
// share of the output tiles (16 u_x * NRB row blocks); per tile it stages the input box (32 input channels) and the dy
// tile (32 output channels) in LDS once and every wave accumulates the (tap, 16x16 block) products of ITS taps over the
// tile's voxels in registers; the sums over a share stay in registers and are written once to slab[share] -- a second
// launch adds the shares in a fixed order (no atomics: bitwise reproducible).
struct DenseDwArgs {
    const float *x, *dy;
    const float *xf_scale, *xf_shift;   // optional (cin): x is used as x * scale + shift
    long ldx, ld_dy;
    int cin, cout, B;
    int In[3], Out[3], k[3], s[3], lo[3];
    int mode, TY, TZ, NRB;
    int box[3];
    int S;                 // shares (gridDim.x)
    float *slab;           // [S][ntap][cin][cout]
};

template <int PREC>
__global__ __launch_bounds__(512) void k_dense_dw(DenseDwArgs g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int ES = PREC ? 2 : 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int n_ci = (g.cin + 31) / 32;
    const int cic = blockIdx.y % n_ci, coc = blockIdx.y / n_ci;
    const int ci0 = cic * 32, co0 = coc * 32;
    const int NCI = min(2, (g.cin - ci0) / 16), NCO = min(2, (g.cout - co0) / 16);
    const int rowb = 32 * ES + 16;
    const int nbox = g.box[0] * g.box[1] * g.box[2];
    unsigned char *s_box = smem_raw;
    unsigned char *s_dy = smem_raw + (((long)nbox * rowb + 15) & ~15L);
    const int ntap = g.k[0] * g.k[1] * g.k[2];
    const int tiles_x = (g.Out[2] + 15) / 16, tiles_y = (g.Out[1] + g.TY - 1) / g.TY, tiles_z = (g.Out[0] + g.TZ - 1) / g.TZ;
    const long ntiles = (long)tiles_x * tiles_y * tiles_z * g.B;
    const long t_lo = ntiles * blockIdx.x / g.S, t_hi = ntiles * (blockIdx.x + 1) / g.S;
    // this wave's taps: wave, wave + 8, ... (at most 4 for 27 taps); per tap NCI x NCO accumulators
    f32x4 acc[4][2][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[a][i][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (long tile = t_lo; tile < t_hi; ++tile) {
        long t = tile;
        const int tx = (int)(t % tiles_x); t /= tiles_x;
        const int ty = (int)(t % tiles_y); t /= tiles_y;
        const int tz = (int)(t % tiles_z); const int b = (int)(t / tiles_z);
        const int ox0 = tx * 16, oy0 = ty * g.TY, oz0 = tz * g.TZ;
        const int iz0 = g.s[0] * oz0 - g.lo[0], iy0 = g.s[1] * oy0 - g.lo[1], ix0 = g.s[2] * ox0 - g.lo[2];
        __syncthreads();                                   // readers of the previous tile are done
        // one (bz, by) row of the box / one row block of the dy tile per wave and pass: scalar index arithmetic, no
        // integer division per element
        for (int rowi = wave; rowi < g.box[0] * g.box[1]; rowi += 8) {
            const int bz = rowi / g.box[1], by = rowi - bz * g.box[1];
            int iz = iz0 + bz, iy = iy0 + by;
            bool okr = true;
            if (g.mode == 0) { iz = min(max(iz, 0), g.In[0] - 1); iy = min(max(iy, 0), g.In[1] - 1); }
            else { okr = iz >= 0 && iz < g.In[0] && iy >= 0 && iy < g.In[1]; if (!okr) { iz = 0; iy = 0; } }
            const float *src_row = g.x + ((((long)b * g.In[0] + iz) * g.In[1] + iy) * g.In[2]) * g.ldx + ci0;
            unsigned char *dst_row = s_box + (long)rowi * g.box[2] * rowb;
            for (int e = lane; e < g.box[2] * 8; e += 64) {   // 8 float4 per voxel (32 channels)
                const int bx = e >> 3, k4 = e & 7;
                int ix = ix0 + bx;
                bool ok = okr && 4 * k4 < 16 * NCI;
                if (g.mode == 0) ix = min(max(ix, 0), g.In[2] - 1);
                else if (ix < 0 || ix >= g.In[2]) { ok = false; ix = 0; }
                f32x4 val = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (ok) val = dense_xf(*(const f32x4 *)(src_row + (long)ix * g.ldx + 4 * k4), g.xf_scale, g.xf_shift, ci0 + 4 * k4);
                if constexpr (PREC) {
                    uint2 pk;
                    pk.x = f2bf(val[0]) | ((unsigned)f2bf(val[1]) << 16); pk.y = f2bf(val[2]) | ((unsigned)f2bf(val[3]) << 16);
                    *(uint2 *)(dst_row + bx * rowb + 8 * k4) = pk;
                } else {
                    *(f32x4 *)(dst_row + bx * rowb + 16 * k4) = val;
                }
            }
        }
        for (int rb = wave; rb < g.NRB; rb += 8) {
            const int oz = oz0 + rb / g.TY, oy = oy0 + rb % g.TY;
            const bool okr = oz < g.Out[0] && oy < g.Out[1];
            const float *src_row = g.dy + ((((long)b * g.Out[0] + (okr ? oz : 0)) * g.Out[1] + (okr ? oy : 0)) * g.Out[2]) * g.ld_dy + co0;
            for (int e = lane; e < 128; e += 64) {
                const int vx = e >> 3, k4 = e & 7;
                const int ox = ox0 + vx;
                const bool ok = okr && ox < g.Out[2] && 4 * k4 < 16 * NCO;
                f32x4 val = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (ok) val = *(const f32x4 *)(src_row + (long)ox * g.ld_dy + 4 * k4);
                unsigned char *dst = s_dy + (long)(rb * 16 + vx) * rowb;
                if constexpr (PREC) {
                    uint2 pk;
                    pk.x = f2bf(val[0]) | ((unsigned)f2bf(val[1]) << 16); pk.y = f2bf(val[2]) | ((unsigned)f2bf(val[3]) << 16);
                    *(uint2 *)(dst + 8 * k4) = pk;
                } else {
                    *(f32x4 *)(dst + 16 * k4) = val;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int tap = wave + 8 * a;
            if (tap >= ntap) break;
            const int dx = tap % g.k[2], dy_ = (tap / g.k[2]) % g.k[1], dz = tap / (g.k[2] * g.k[1]);
            for (int rb = 0; rb < g.NRB; ++rb) {
                const int rz = rb / g.TY, ry = rb % g.TY;
                const int bv0 = ((g.s[0] * rz + dz) * g.box[1] + (g.s[1] * ry + dy_)) * g.box[2] + dx;   // + s * voxel
                if constexpr (PREC) {
                    // lane (r, q) wants voxels 4q..4q+3 of channel r: one transposing read per operand (lane addresses voxel
                    // 4q + (r >> 2), channels 4 (r & 3)..+3; see k_dense_dw3) instead of four ds_read_u16
                    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                    s16x4 af[2], bf[2];
                    {
                        const int vx = 4 * q + (r >> 2), cc = 4 * (r & 3);
                        const unsigned char *xr = s_box + (long)(bv0 + g.s[2] * vx) * rowb;
                        const unsigned char *dr = s_dy + (long)(rb * 16 + vx) * rowb;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            af[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(xr + 2 * (16 * i + cc)));
                            bf[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(dr + 2 * (16 * i + cc)));
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            if (i < NCI && c < NCO) acc[a][i][c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af[i], bf[c], acc[a][i][c], 0, 0, 0);
                } else {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int vx = 4 * m + q;
                        const unsigned char *xr = s_box + (long)(bv0 + g.s[2] * vx) * rowb;
                        const unsigned char *dr = s_dy + (long)(rb * 16 + vx) * rowb;
                        float af[2], bf[2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) { af[i] = *(const float *)(xr + 4 * (16 * i + r)); bf[i] = *(const float *)(dr + 4 * (16 * i + r)); }
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int c = 0; c < 2; ++c)
                                if (i < NCI && c < NCO) acc[a][i][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[c], acc[a][i][c], 0, 0, 0);
                    }
                }
            }
        }
    }
    // D[row 4q + k][col r] of (tap, i, c) = dw[tap][ci0 + 16 i + 4q + k][co0 + 16 c + r]
    float *out = g.slab + (long)blockIdx.x * ntap * g.cin * g.cout;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int tap = wave + 8 * a;
        if (tap >= ntap) break;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (i >= NCI || c >= NCO) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    out[((long)tap * g.cin + ci0 + 16 * i + 4 * q + k) * g.cout + co0 + 16 * c + r] = acc[a][i][c][k];
            }
    }
}

// ---- fast path of the weight gradient: 3 x 3 x 3 taps, unit strides, tile 16 x 4 x 4 (box 18 x 6 x 6) -----------------------
// Compile-time LDS offsets (see k_dense_conv3), the dy fragments of a row block loaded once for the wave's 3-4 taps, and for
// bf16 the voxel-major tiles are read with ds_read_b64_tr_b16: one transposing read delivers, to lane (channel r, voxel
// group q), the four voxels 4q..4q+3 of channel r -- the MFMA's k-contiguous operand -- where the generic kernel issues four
// ds_read_u16 (it was LDS-instruction bound: 8 reads per MFMA at 16 channels).
template <int PREC, int NCI, int NCO>
__global__ __launch_bounds__(512) void k_dense_dw3(DenseDwArgs g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int ES = PREC ? 2 : 4, ROWB = 32 * ES + 16;
    constexpr int BX = 18, BY = 6, BZ = 6, NBOX = BX * BY * BZ;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int n_ci = (g.cin + 31) / 32;
    const int cic = blockIdx.y % n_ci, coc = blockIdx.y / n_ci;
    const int ci0 = cic * 32, co0 = coc * 32;
    unsigned char *s_box = smem_raw;
    unsigned char *s_dy = smem_raw + ((NBOX * ROWB + 15) & ~15);
    const int tiles_x = (g.Out[2] + 15) / 16, tiles_y = (g.Out[1] + 3) / 4, tiles_z = (g.Out[0] + 3) / 4;
    const long ntiles = (long)tiles_x * tiles_y * tiles_z * g.B;
    const long t_lo = ntiles * blockIdx.x / g.S, t_hi = ntiles * (blockIdx.x + 1) / g.S;
    f32x4 acc[4][NCI][NCO];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < NCI; ++i)
#pragma unroll
            for (int c = 0; c < NCO; ++c) acc[a][i][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this wave's taps: wave + 8a; their box offsets (dz * BY + dy) * BX + dx
    int toff[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { const int tap = min(wave + 8 * a, 26); toff[a] = ((tap / 9) * BY + (tap / 3) % 3) * BX + tap % 3; }
    // per-lane byte offsets inside a row block: transposing read (bf16): lane i of a 16-lane group supplies row 4q + (i >> 2),
    // columns 4 (i & 3); plain reads (fp32): voxel 4m + q, channel r
    const int tr_row = 4 * q + (r >> 2), tr_col = 4 * (r & 3);

    // Staging: the box (18 x 6 x 6 voxels, 16 NCI channels) and the dy tile (256 voxels, 16 NCO channels) of a tile as 16-byte
    // pieces dealt flat over the 512 threads, ALL of a thread's loads requested before the first is parked (the row-per-wave
    // loop this replaces waited for each box row before it asked for the next: five dependent round trips per tile, 13 us per
    // tile at 128^3 x 16 -> 16 where the tile's MFMAs need 3).  With one channel block per side (PF) the requests of tile
    // t + 1 go out BEFORE the MFMAs of tile t and are parked behind them: the round trip runs beside the arithmetic.
    constexpr int PVX = 4 * NCI, PVY = 4 * NCO;                       // 16-byte pieces per voxel (fp32 source)
    constexpr int BOX_E = NBOX * PVX, NITB = (BOX_E + 511) / 512;
    constexpr int DY_E = 256 * PVY, NITD = (DY_E + 511) / 512;
    constexpr bool PF = NCI * NCO <= 2;
    f32x4 vb[NITB], vd[NITD];
    bool okb[NITB], okd[NITD];
    auto issue = [&](long tile, int b_lo, int b_hi, bool with_dy) {   // (literal ranges at the call sites: resolved when unrolled)
        long t = tile;
        const int tx = (int)(t % tiles_x); t /= tiles_x;
        const int ty = (int)(t % tiles_y); t /= tiles_y;
        const int tz = (int)(t % tiles_z); const int b = (int)(t / tiles_z);
        const int ox0 = tx * 16, oy0 = ty * 4, oz0 = tz * 4;
        const int iz0 = oz0 - g.lo[0], iy0 = oy0 - g.lo[1], ix0 = ox0 - g.lo[2];
#pragma unroll
        for (int it = 0; it < NITB; ++it) {
            if (it < b_lo || it >= b_hi) continue;
            const int e = it * 512 + tid;
            const int vox = e / PVX, k4 = e - vox * PVX;
            const int rowi = vox / BX, bx = vox - rowi * BX;
            const int bz = rowi / BY, by = rowi - bz * BY;
            int iz = iz0 + bz, iy = iy0 + by, ix = ix0 + bx;
            bool ok = e < BOX_E;
            if (g.mode == 0) { iz = min(max(iz, 0), g.In[0] - 1); iy = min(max(iy, 0), g.In[1] - 1); ix = min(max(ix, 0), g.In[2] - 1); }
            else if (iz < 0 || iz >= g.In[0] || iy < 0 || iy >= g.In[1] || ix < 0 || ix >= g.In[2]) { ok = false; }
            if (!ok) { iz = 0; iy = 0; ix = 0; }
            okb[it] = ok;
            vb[it] = *(const f32x4 *)(g.x + ((((long)b * g.In[0] + iz) * g.In[1] + iy) * g.In[2] + ix) * g.ldx + ci0 + (ok ? 4 * k4 : 0));
        }
#pragma unroll
        for (int it = 0; it < NITD; ++it) {
            if (!with_dy) continue;
            const int e = it * 512 + tid;
            const int vox = e / PVY, k4 = e - vox * PVY;       // vox = rb * 16 + vx
            const int rb = vox >> 4, vx = vox & 15;
            const int oz = oz0 + (rb >> 2), oy = oy0 + (rb & 3), ox = ox0 + vx;
            const bool ok = e < DY_E && oz < g.Out[0] && oy < g.Out[1] && ox < g.Out[2];
            okd[it] = ok;
            vd[it] = *(const f32x4 *)(g.dy + ((((long)b * g.Out[0] + (ok ? oz : 0)) * g.Out[1] + (ok ? oy : 0)) * g.Out[2] + (ok ? ox : 0)) * g.ld_dy + co0 + (ok ? 4 * k4 : 0));
        }
    };
    auto park = [&](int b_lo, int b_hi, bool with_dy) {
#pragma unroll
        for (int it = 0; it < NITB; ++it) {
            if (it < b_lo || it >= b_hi) continue;
            const int e = it * 512 + tid;
            if (e >= BOX_E) continue;
            const int vox = e / PVX, k4 = e - vox * PVX;
            f32x4 val = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (okb[it]) val = dense_xf(vb[it], g.xf_scale, g.xf_shift, ci0 + 4 * k4);
            unsigned char *dst = s_box + vox * ROWB;
            if constexpr (PREC) {
                uint2 pk;
                pk.x = f2bf(val[0]) | ((unsigned)f2bf(val[1]) << 16); pk.y = f2bf(val[2]) | ((unsigned)f2bf(val[3]) << 16);
                *(uint2 *)(dst + 8 * k4) = pk;
            } else {
                *(f32x4 *)(dst + 16 * k4) = val;
            }
        }
#pragma unroll
        for (int it = 0; it < NITD; ++it) {
            if (!with_dy) continue;
            const int e = it * 512 + tid;
            if (e >= DY_E) continue;
            const int vox = e / PVY, k4 = e - vox * PVY;
            const f32x4 val = okd[it] ? vd[it] : (f32x4){0.f, 0.f, 0.f, 0.f};
            unsigned char *dst = s_dy + vox * ROWB;
            if constexpr (PREC) {
                uint2 pk;
                pk.x = f2bf(val[0]) | ((unsigned)f2bf(val[1]) << 16); pk.y = f2bf(val[2]) | ((unsigned)f2bf(val[3]) << 16);
                *(uint2 *)(dst + 8 * k4) = pk;
            } else {
                *(f32x4 *)(dst + 16 * k4) = val;
            }
        }
    };
    constexpr int HB = NITB > 8 ? 8 : NITB;   // wide tiles (two channel blocks on both sides): two batches per tile instead of 15 loads held
    if (PF && t_lo < t_hi) issue(t_lo, 0, NITB, true);
    for (long tile = t_lo; tile < t_hi; ++tile) {
        __syncthreads();                      // the previous tile's readers are done with the box and the dy tile
        if constexpr (PF) park(0, NITB, true);
        else {
            issue(tile, 0, HB, HB == NITB); park(0, HB, HB == NITB);
            if constexpr (HB < NITB) { issue(tile, HB, NITB, true); park(HB, NITB, true); }
        }
        __syncthreads();
        if (PF && tile + 1 < t_hi) issue(tile + 1, 0, NITB, true);
        __builtin_amdgcn_sched_barrier(0);    // (keep the requests in front of the MFMA loop)
#pragma unroll 4
        for (int rb = 0; rb < 16; ++rb) {
            const int rbase = ((rb >> 2) * BY + (rb & 3)) * BX;
            if constexpr (PREC) {
                s16x4 bf[NCO];
#pragma unroll
                for (int c = 0; c < NCO; ++c)
                    bf[c] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(s_dy + (rb * 16 + tr_row) * ROWB + 2 * (16 * c + tr_col)));
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (a == 3 && wave + 24 >= 27) break;          // waves 3..7 have three taps
                    s16x4 af[NCI];
#pragma unroll
                    for (int i = 0; i < NCI; ++i)
                        af[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(s_box + (rbase + toff[a] + tr_row) * ROWB + 2 * (16 * i + tr_col)));
#pragma unroll
                    for (int i = 0; i < NCI; ++i)
#pragma unroll
                        for (int c = 0; c < NCO; ++c) acc[a][i][c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af[i], bf[c], acc[a][i][c], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int vx = 4 * m + q;
                    float bf[NCO];
#pragma unroll
                    for (int c = 0; c < NCO; ++c) bf[c] = *(const float *)(s_dy + (rb * 16 + vx) * ROWB + 4 * (16 * c + r));
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        if (a == 3 && wave + 24 >= 27) break;
#pragma unroll
                        for (int i = 0; i < NCI; ++i) {
                            const float af = *(const float *)(s_box + (rbase + toff[a] + vx) * ROWB + 4 * (16 * i + r));
#pragma unroll
                            for (int c = 0; c < NCO; ++c) acc[a][i][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[c], acc[a][i][c], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    float *out = g.slab + (long)blockIdx.x * 27 * g.cin * g.cout;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int tap = wave + 8 * a;
        if (tap >= 27) break;
#pragma unroll
        for (int i = 0; i < NCI; ++i)
#pragma unroll
            for (int c = 0; c < NCO; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    out[((long)tap * g.cin + ci0 + 16 * i + 4 * q + k) * g.cout + co0 + 16 * c + r] = acc[a][i][c][k];
    }
}

// Sum of the S partial slabs, 4 consecutive elements per thread.  The narrow layers have few elements and many partials
// (128^3 x 16 -> 16: 6,912 weights, 512 slabs = 14 MB): with one thread per 4 elements summing all the slabs that was 7
// workgroups each walking 512 dependent-latency steps, 25 us per convolution and 1.2 ms per cfg2 step.  So the slabs are
// dealt to SL slices of the workgroup as well (thread = (element quad, slice); slice j takes slabs j, j + SL, ...: eight
// loads in flight), and the SL slice sums are combined through LDS in the fixed order 0 .. SL-1: the result depends on
// the launch shape only, bitwise reproducible.  SL is a power of two <= 64 chosen by the host from n (1 for the big layers).
__device__ __forceinline__ f32x4 dense_dw_sum(const float *__restrict__ slab, int S, long n, long e, int SL, bool ok, f32x4 *s_part)
{
    const int EQ = 256 / SL, eq = threadIdx.x % EQ, sl = threadIdx.x / EQ;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (ok) {
        int s = sl;
        for (; s + 7 * SL < S; s += 8 * SL) {
            f32x4 p[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) p[k] = *(const f32x4 *)(slab + (long)(s + k * SL) * n + e);
#pragma unroll
            for (int k = 0; k < 8; ++k) v += p[k];
        }
        for (; s < S; s += SL) v += *(const f32x4 *)(slab + (long)s * n + e);
    }
    if (SL == 1) return v;
    s_part[threadIdx.x] = v;
    __syncthreads();
    if (sl == 0)
        for (int j = 1; j < SL; ++j) v += s_part[j * EQ + eq];
    return v;
}

__global__ __launch_bounds__(256) void k_dense_dw_reduce(const float *__restrict__ slab, int S, long n, float *__restrict__ dw, int SL)
{
    __shared__ f32x4 s_part[256];
    const int EQ = 256 / SL;
    const long e = ((long)blockIdx.x * EQ + threadIdx.x % EQ) * 4;
    const f32x4 v = dense_dw_sum(slab, S, n, e, SL, e < n, s_part);
    if (e < n && threadIdx.x / EQ == 0) *(f32x4 *)(dw + e) += v;
}

// the same sum written (not accumulated) in torch's parameter layout: element [tap][ix][iy] of the slab layout goes to
// out[(iy * vx + ix) * ntap + tap] for ix < vx, iy < vy (the zero-padded channels are dropped) -- nn.Conv weight
// (cout, cin, taps) with x = input, y = output channels, nn.ConvTranspose weight (cin, cout, taps) with the roles swapped.
__global__ __launch_bounds__(256) void k_dense_dw_reduce_t(const float *__restrict__ slab, int S, long n, int cx, int cy, int vx, int vy,
                                                           int ntap, float *__restrict__ out, int SL)
{
    __shared__ f32x4 s_part[256];
    const int EQ = 256 / SL;
    const long e = ((long)blockIdx.x * EQ + threadIdx.x % EQ) * 4;
    const int iy = (int)(e % cy);
    const long r = e / cy;
    const int ix = (int)(r % cx), tap = (int)(r / cx);
    const bool ok = e < n && ix < vx && iy < vy;
    const f32x4 v = dense_dw_sum(slab, S, n, e, SL, ok, s_part);
    if (!ok || threadIdx.x / EQ != 0) return;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (iy + k < vy) out[((long)(iy + k) * vx + ix) * ntap + tap] = v[k];
}

// The same for the wide layers (few slabs, many weights: 256 x 256 x 27 at 8^3), where the sum is cheap and the WRITE was the
// cost: a thread's four iy values land 27 vx floats apart in torch's layout (78 us for 1.8 M weights).  A workgroup takes a
// 16 (ix) x 16 (iy) block of all taps: reads 64-byte runs of iy, transposes through LDS, writes runs of 16 x ntap floats.
__global__ __launch_bounds__(256) void k_dense_dw_reduce_tt(const float *__restrict__ slab, int S, long n, int cx, int cy, int vx, int vy,
                                                            int ntap, float *__restrict__ out)
{
    __shared__ float s_t[27][16][17];
    const int nby = cy / 16;
    const int ix0 = (blockIdx.x / nby) * 16, iy0 = (blockIdx.x % nby) * 16;
    const int tid = threadIdx.x, ixl = tid >> 4, iyl = tid & 15;
    for (int tap = 0; tap < ntap; ++tap) {
        const long e = ((long)tap * cx + ix0 + ixl) * cy + iy0 + iyl;
        float v = 0.f;
        int s = 0;
        for (; s + 4 <= S; s += 4) {
            const float a = slab[(long)s * n + e], b = slab[(long)(s + 1) * n + e], c = slab[(long)(s + 2) * n + e], d = slab[(long)(s + 3) * n + e];
            v += a; v += b; v += c; v += d;
        }
        for (; s < S; ++s) v += slab[(long)s * n + e];
        s_t[tap][ixl][iyl] = v;
    }
    __syncthreads();
    const int row_e = 16 * ntap;                       // floats of one iy row of the block in the output: [ix 16][tap]
    for (int e = tid; e < 16 * row_e; e += 256) {
        const int yl = e / row_e, rem = e - yl * row_e;
        const int xl = rem / ntap, tap = rem - xl * ntap;
        const int ix = ix0 + xl, iy = iy0 + yl;
        if (ix < vx && iy < vy) out[((long)iy * vx + ix) * ntap + tap] = s_t[tap][xl][yl];
    }
}

extern "C" int64_t urn_dense_dw_scratch_bytes(int batch, const int *out_dims, const int *k, int cin, int cout)
{
    if (!out_dims || !k || cin <= 0 || cout <= 0) return -1;
    const long ntap = (long)k[0] * k[1] * k[2];
    const long tiles = (long)((out_dims[2] + 15) / 16) * out_dims[1] * out_dims[0] * batch;   // upper bound on the tile count
    long S = tiles < 512 ? tiles : 512;   // two workgroups per CU (72 KB of LDS each): one stages while the other multiplies
    if (S < 1) S = 1;
    return S * ntap * cin * cout * 4 + 256;
}

extern "C" int urn_dense_dw(const float *x, int64_t ldx, int cin, const float *dy, int64_t ld_dy, int cout, int batch,
                            const int *in_dims, const int *out_dims, const int *k, const int *s, const int *lo, int mode,
                            float *dw, int dw_layout, int cin_valid, int cout_valid, const float *xf_scale, const float *xf_shift,
                            void *scratch, int64_t scratch_bytes, int precision, void *stream)
{
    URN_CHECK_ARG((xf_scale == nullptr) == (xf_shift == nullptr), "scale and shift go together");
    URN_CHECK_ARG(x && dy && dw && scratch && in_dims && out_dims && k && s && lo, "null pointer");
    URN_CHECK_ARG(dw_layout == 0 || (dw_layout == 1 && cin_valid > 0 && cin_valid <= cin && cout_valid > 0 && cout_valid <= cout),
                  "dw_layout 0 ([tap][cin][cout], accumulated) or 1 (torch layout, written) with the valid channel counts");
    URN_CHECK_ARG(cin > 0 && cout > 0 && cin % 16 == 0 && cout % 16 == 0 && batch > 0 && ldx >= cin && ld_dy >= cout && ldx % 4 == 0 && ld_dy % 4 == 0,
                  "channel counts must be multiples of 16");
    DenseDwArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.dy = dy; a.ldx = (long)ldx; a.ld_dy = (long)ld_dy; a.cin = cin; a.cout = cout; a.B = batch; a.mode = mode;
    a.xf_scale = xf_scale; a.xf_shift = xf_shift;
    int ntap = 1;
    for (int d = 0; d < 3; ++d) {
        a.In[d] = in_dims[d]; a.Out[d] = out_dims[d]; a.k[d] = k[d]; a.s[d] = s[d]; a.lo[d] = lo[d];
        URN_CHECK_ARG(k[d] >= 1 && k[d] <= 3 && s[d] >= 1 && s[d] <= 2 && out_dims[d] >= 1 && in_dims[d] >= 1, "bad geometry");
        ntap *= k[d];
    }
    // row blocks per tile: 16 for stride 1, 4 for stride 2 (the staged box grows with the stride)
    const int stride = (s[0] > 1 || s[1] > 1 || s[2] > 1) ? 2 : 1;
    a.NRB = stride == 1 ? 16 : 4;
    a.TY = a.Out[0] > 1 ? (a.NRB == 16 ? 4 : 2) : a.NRB;
    if (a.Out[1] < a.TY) { a.TY = 1; while (a.TY * 2 <= a.Out[1] && a.TY < a.NRB) a.TY *= 2; }
    a.TZ = a.NRB / a.TY;
    a.box[2] = a.s[2] * 15 + a.k[2]; a.box[1] = a.s[1] * (a.TY - 1) + a.k[1]; a.box[0] = a.s[0] * (a.TZ - 1) + a.k[0];
    const long nbox = (long)a.box[0] * a.box[1] * a.box[2];
    const int es = precision ? 2 : 4;
    const long rowb = 32L * es + 16;
    const size_t lds = (size_t)(((nbox * rowb + 15) & ~15L) + (long)a.NRB * 16 * rowb);
    if (lds > 160 * 1024) { urn_set_error("urn_dense_dw: staged box of %ld voxels does not fit LDS", nbox); return URN_EUNSUPPORTED; }
    const long ntiles = (long)((a.Out[2] + 15) / 16) * ((a.Out[1] + a.TY - 1) / a.TY) * ((a.Out[0] + a.TZ - 1) / a.TZ) * batch;
    long S = ntiles < 512 ? ntiles : 512;
    if (S < 1) S = 1;
    const long n = (long)ntap * cin * cout;
    URN_CHECK_ARG(scratch_bytes >= S * n * 4, "scratch smaller than urn_dense_dw_scratch_bytes");
    a.S = (int)S; a.slab = (float *)scratch;
    hipStream_t st = (hipStream_t)stream;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)k_dense_dw<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_dense_dw<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const dim3 grid((unsigned)S, ((cin + 31) / 32) * ((cout + 31) / 32)), block(512);
    bool done = false;
    {   // fast path: every channel block of the launch is a full 32 (or the layer has exactly 16 channels on that side)
        const int nci = cin >= 32 ? 2 : 1, nco = cout >= 32 ? 2 : 1;
        const bool full = (cin % 32 == 0 || cin == 16) && (cout % 32 == 0 || cout == 16);
        if (ntap == 27 && stride == 1 && a.NRB == 16 && a.TY == 4 && a.TZ == 4 && full) {
#define URN_DW3(P, I, C) if (precision == P && nci == I && nco == C) { \
                static bool attr_done = false; \
                if (!attr_done) { (void)hipFuncSetAttribute((const void *)k_dense_dw3<P, I, C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_done = true; } \
                hipLaunchKernelGGL((k_dense_dw3<P, I, C>), grid, block, lds, st, a); done = true; }
            URN_DW3(0, 1, 1) URN_DW3(0, 1, 2) URN_DW3(0, 2, 1) URN_DW3(0, 2, 2) URN_DW3(1, 1, 1) URN_DW3(1, 1, 2) URN_DW3(1, 2, 1) URN_DW3(1, 2, 2)
#undef URN_DW3
        }
    }
    if (!done) {
        if (precision) hipLaunchKernelGGL(k_dense_dw<1>, grid, block, lds, st, a);
        else hipLaunchKernelGGL(k_dense_dw<0>, grid, block, lds, st, a);
    }
    // slices of the partial slabs per workgroup (see dense_dw_sum): enough threads for the chip, at most one slice per two slabs
    int SL = 1;
    while (SL < 64 && (n / 4) * SL < 131072 && 2 * SL * 2 <= S) SL *= 2;
    const unsigned rgrid = (unsigned)urn_cdiv((n + 3) / 4, 256 / SL);
    if (dw_layout == 1 && SL == 1 && ntap <= 27 && (long)cin * cout >= 64 * 64)
        hipLaunchKernelGGL(k_dense_dw_reduce_tt, dim3((cin / 16) * (cout / 16)), dim3(256), 0, st, (const float *)scratch, (int)S, n, cin,
                           cout, cin_valid, cout_valid, ntap, dw);
    else if (dw_layout == 1)
        hipLaunchKernelGGL(k_dense_dw_reduce_t, dim3(rgrid), dim3(256), 0, st, (const float *)scratch, (int)S, n, cin,
                           cout, cin_valid, cout_valid, ntap, dw, SL);
    else
        hipLaunchKernelGGL(k_dense_dw_reduce, dim3(rgrid), dim3(256), 0, st, (const float *)scratch, (int)S, n, dw, SL);
    URN_LAUNCH_CHECK();
    return URN_OK;
}
