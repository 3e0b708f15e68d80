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

// Second form of the weight gradient (default): same tiling and arithmetic, restructured after the findings on the
// forward kernel (tools/ablate_gconv.py): (1) the valid pairs of up to 1024 candidate rows are compacted in ONE pass
// (four table loads in flight per thread, two block barriers) instead of one pass per 256 rows with a carried remainder;
// (2) the batch loop is straight-line code -- loads of batch +2 into a two-slot register ring, parking of batch +1 and
// the MFMAs of the current batch in one basic block, every load unconditional (clamped pair index) so that the
// compiler counts them (s_waitcnt vmcnt(N)) and interleaves everything with the MFMAs; one LDS-only barrier per batch.
// S = output blocks per wave (ceil(blocks / 4)), XF = the conv input was relu(x*scale + shift).
#define DW2_LIST 1024
#include <type_traits>
// KT = pairs per batch: 32 for tiles of >= 3 output blocks; narrow tiles take bigger batches and split the batch's
// pairs (the contraction dimension) over the waves, which would otherwise all compute the same block: one block
// (16 x 16 channels): 128 pairs, 4 waves x 8 k-steps; two blocks (16 x 32, 32 x 16): 64 pairs, 2 waves per block.
// PREC: MFMA operand precision, 0 fp32, 1 bf16, 2 fp16 (as in the forward kernel): the staged rows are rounded while they
//       are parked; a fragment = four consecutive PAIRS of one channel, assembled from four 16-bit LDS reads
//       (v_mfma_f32_16x16x16: one MFMA per 16 pairs instead of four).
template <int S, int XF, int KT, int PREC = 0>
__global__ __launch_bounds__(256) void k_gconv_dw2(const float *__restrict__ x, const float *__restrict__ xf_scale,
                                                   const float *__restrict__ xf_shift, const float *__restrict__ dy,
                                                   const int *__restrict__ tbl, long ld, long n_out, int cin,
                                                   int cout, long chunk, int n_ci_tiles,
                                                   float *__restrict__ dw, long ld_dy, float *__restrict__ slab, int row_ok)
{
    __shared__ int s_in[DW2_LIST], s_out[DW2_LIST];
    __shared__ int s_cnt[16];
    // leading dimensions = 16 mod 32 floats, so that the four pair rows of a fragment read fall on different banks
    constexpr int A_LD = KT == 32 ? DW_MAXI * 16 + 16 : (KT == 64 ? 48 : 16);
    constexpr int B_LD = KT == 32 ? DW_MAXN * 16 + 32 : (KT == 64 ? 48 : 16);
    // 16-bit operands: the same row length in elements, i.e. half the floats per row
    __shared__ __attribute__((aligned(16))) float s_a[2][KT][PREC ? A_LD / 2 : A_LD];
    __shared__ __attribute__((aligned(16))) float s_b[2][KT][PREC ? B_LD / 2 : B_LD];
    auto lds_store = [&](float *row_ptr, int c4, f32x4 v) {
        if constexpr (PREC == 0) {
            *(f32x4 *)(row_ptr + 4 * c4) = v;
        } else if constexpr (PREC == 1) {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            const bf16x4 h = __builtin_convertvector(v, bf16x4);
            *(uint2 *)((unsigned short *)row_ptr + 4 * c4) = __builtin_bit_cast(uint2, h);
        } else {
            typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
            const h16x4 h = __builtin_convertvector(v, h16x4);
            *(uint2 *)((unsigned short *)row_ptr + 4 * c4) = __builtin_bit_cast(uint2, h);
        }
    };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    // XCD-aware work assignment (workgroups are dealt round-robin to the 8 XCDs in launch order, each with its own L2): XCD x
    // takes a contiguous range of the (row chunk, offset, channel tile) items, chunk-major -- the K offsets and the channel
    // tiles of a row chunk read the same dy rows and neighbouring x rows, now through one L2 instead of up to eight
    unsigned chunk_id, o_id, z_id;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z, total = gx * gy * gz;
        const unsigned lin = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x;
        const unsigned xq = total >> 3, xr = total & 7u, xcd = lin & 7u, slot = lin >> 3;
        const unsigned v = xcd * xq + (xcd < xr ? xcd : xr) + slot;
        chunk_id = v / (gy * gz);
        const unsigned rem = v - chunk_id * (gy * gz);
        z_id = rem / gy; o_id = rem - z_id * gy;
    }
    const int o = (int)o_id;
    const int ci_tile = (int)z_id % n_ci_tiles, co_tile = (int)z_id / n_ci_tiles;
    const int ci0 = ci_tile * DW_MAXI * 16, co0 = co_tile * DW_MAXN * 16;
    const int ci_w = min(DW_MAXI * 16, cin - ci0), co_w = min(DW_MAXN * 16, cout - co0);
    const int mi_n = ci_w / 16, ni_n = co_w / 16, nblk = mi_n * ni_n;

    f32x4 acc[S];
    int mi_[S], ni_[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // KT == 32: block b -> wave b % 4, slot b / 4; past the last block: recompute it (its result is not flushed).
        // KT > 32 (one or two blocks, S == 1): wave -> block wave % nblk, and the batch's k-steps are split
        const int blk = KT == 32 ? min(wave + 4 * s, nblk - 1) : wave % nblk;
        mi_[s] = blk / ni_n; ni_[s] = blk - mi_[s] * ni_n;
    }
    // full-width tiles (four input-channel blocks): wave w owns input block w and ALL output blocks -- its x fragment of a
    // k-step is read from LDS once for the S MFMAs instead of once per MFMA (6 transposing reads per 5 MFMAs instead of 10)
    const bool rowmode = row_ok && KT == 32 && mi_n == 4;
    if (rowmode) {
#pragma unroll
        for (int s = 0; s < S; ++s) { mi_[s] = wave; ni_[s] = min(s, ni_n - 1); }
    }
    constexpr int WPB = KT == 128 ? 4 : (KT == 64 ? 2 : 1);   // waves per block
    constexpr int KPM = PREC ? 16 : 4;                          // pairs contracted by one MFMA
    constexpr int KSW = KT / KPM / WPB;                         // k-steps per wave and batch
    const int ks0 = KT == 32 ? 0 : (wave / (4 / WPB)) * KSW;
    // what this thread moves per batch: two 16-byte pieces of the x rows, three of the dy rows
    const int a_tot = KT * (ci_w / 4), b_tot = KT * (co_w / 4);
    int a_rr[2], a_c4[2], b_rr[3], b_c4[3];
    f32x4 xsc[2], xsh[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = min(j * 256 + tid, a_tot - 1);
        a_rr[j] = e / (ci_w / 4); a_c4[j] = e - a_rr[j] * (ci_w / 4);
        if (XF) { xsc[j] = *(const f32x4 *)(xf_scale + ci0 + 4 * a_c4[j]); xsh[j] = *(const f32x4 *)(xf_shift + ci0 + 4 * a_c4[j]); }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int e = min(j * 256 + tid, b_tot - 1);
        b_rr[j] = e / (co_w / 4); b_c4[j] = e - b_rr[j] * (co_w / 4);
    }
    f32x4 ra[2][2], rb[2][3];   // register ring: two batches in flight

    const long row_begin = (long)chunk_id * chunk;
    const long row_end = min(n_out, row_begin + chunk);
    for (long sub = row_begin; sub < row_end; sub += DW2_LIST) {
        // ---- phase A: compact the valid (in, out) pairs of up to 1024 rows, row order preserved ----
        const long sub_end = min(row_end, sub + DW2_LIST);
        int idx[4];
        unsigned long long bal[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = sub + r * 256 + tid;
            idx[r] = row < sub_end ? tbl[(long)o * ld + row] : -1;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bal[r] = __ballot(idx[r] >= 0);
            if (lane == 0) s_cnt[r * 4 + wave] = __popcll(bal[r]);
        }
        __syncthreads();
        int cnt = 0;
        int base[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int v = s_cnt[i];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (i < r * 4 + wave) base[r] += v;
            cnt += v;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (idx[r] >= 0) {
                const int p = base[r] + __popcll(bal[r] & ((1ull << lane) - 1ull));
                s_in[p] = idx[r];
                s_out[p] = (int)(sub + r * 256 + tid);
            }
        __syncthreads();
        if (cnt == 0) continue;   // block-uniform
        const int nbatch = (cnt + KT - 1) / KT;

        // ---- phase B: batches of 32 pairs ----
        auto fetch_a = [&](int b, auto slot, auto jj) {
            constexpr int u = decltype(slot)::value, j = decltype(jj)::value;
            const int p = min(b * KT + a_rr[j], cnt - 1);   // past the end: the last pair again (zero-filled when parked)
            ra[u][j] = *(const f32x4 *)(x + (long)s_in[p] * cin + ci0 + 4 * a_c4[j]);
        };
        auto fetch_b = [&](int b, auto slot, auto jj) {
            constexpr int u = decltype(slot)::value, j = decltype(jj)::value;
            const int p = min(b * KT + b_rr[j], cnt - 1);
            rb[u][j] = *(const f32x4 *)(dy + (long)s_out[p] * ld_dy + co0 + 4 * b_c4[j]);
        };
        auto park_a = [&](int b, int buf, auto slot, auto jj) {
            constexpr int u = decltype(slot)::value, j = decltype(jj)::value;
            const int nb = cnt - b * KT;
            f32x4 v = ra[u][j];
            if (XF) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaf(v[k], xsc[j][k], xsh[j][k]), 0.f);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = a_rr[j] < nb ? v[k] : 0.f;   // zero-fill the tail of the last batch
            lds_store(&s_a[buf][a_rr[j]][0], a_c4[j], v);
        };
        auto park_b = [&](int b, int buf, auto slot, auto jj) {
            constexpr int u = decltype(slot)::value, j = decltype(jj)::value;
            const int nb = cnt - b * KT;
            f32x4 v = rb[u][j];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = b_rr[j] < nb ? v[k] : 0.f;
            lds_store(&s_b[buf][b_rr[j]][0], b_c4[j], v);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        fetch_a(0, I0(), I0()); fetch_a(0, I0(), I1()); fetch_b(0, I0(), I0()); fetch_b(0, I0(), I1()); fetch_b(0, I0(), I2());
        fetch_a(1, I1(), I0()); fetch_a(1, I1(), I1()); fetch_b(1, I1(), I0()); fetch_b(1, I1(), I1()); fetch_b(1, I1(), I2());
        park_a(0, 0, I0(), I0()); park_a(0, 0, I0(), I1()); park_b(0, 0, I0(), I0()); park_b(0, 0, I0(), I1()); park_b(0, 0, I0(), I2());
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // one batch: MFMAs of batch b (LDS image b & 1), loads of batch b+2 into slot U, parking of batch b+1 from slot U^1
        auto step = [&](int b, auto slot) {
            constexpr int U = decltype(slot)::value;
            using SU = std::integral_constant<int, U>;
            using SN = std::integral_constant<int, U ^ 1>;
            const int buf = b & 1;
            auto piece = [&](auto pi) {
                constexpr int i = decltype(pi)::value;
                if constexpr (i == 0) fetch_a(b + 2, SU(), I0());
                else if constexpr (i == 1) fetch_a(b + 2, SU(), I1());
                else if constexpr (i == 2) fetch_b(b + 2, SU(), I0());
                else if constexpr (i == 3) fetch_b(b + 2, SU(), I1());
                else if constexpr (i == 4) fetch_b(b + 2, SU(), I2());
                else if constexpr (i == 5) park_a(b + 1, buf ^ 1, SN(), I0());
                else if constexpr (i == 6) park_a(b + 1, buf ^ 1, SN(), I1());
                else if constexpr (i == 7) park_b(b + 1, buf ^ 1, SN(), I0());
                else if constexpr (i == 8) park_b(b + 1, buf ^ 1, SN(), I1());
                else if constexpr (i == 9) park_b(b + 1, buf ^ 1, SN(), I2());
            };
            constexpr int PER = (10 + S - 1) / S;
            typedef short s16x4k __attribute__((ext_vector_type(4)));
            s16x4k av_keep[2] = {};
            auto group = [&](auto ss) {
                constexpr int sl = decltype(ss)::value;
                if constexpr (PREC == 0) {
#pragma unroll
                    for (int kk = 0; kk < KSW; ++kk) {
                        const int ks = ks0 + kk;
                        const float av = s_a[buf][4 * ks + q][mi_[sl] * 16 + m];
                        const float bv = s_b[buf][4 * ks + q][ni_[sl] * 16 + m];
                        acc[sl] = MFMA16(av, bv, acc[sl]);
                    }
                } else {
                    // lane (m, q): pairs 16 ks + 4 q + i (i < 4) of channel m of the block = one transposing read of the
                    // pair-major tile (ds_read_b64_tr_b16; four ds_read_u16 per operand made the 16-bit kernel
                    // LDS-instruction bound: 8 reads per MFMA).  The wave's two k-steps of a batch (KSW == 2 for every KT) are the
                    // eight contraction slots of ONE v_mfma_f32_16x16x32_*: slots 0-3 = pairs 4 q + i of the first step, 4-7 of the second
                    static_assert(KSW == 2, "16-bit operands: two k-steps of 16 pairs per wave and batch");
                    typedef short s16x4 __attribute__((ext_vector_type(4)));
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                    const int tr_row = 16 * ks0 + 4 * q + (m >> 2), tr_col = 4 * (m & 3);
                    if (sl == 0 || !rowmode) {
                        av_keep[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)((unsigned short *)&s_a[buf][tr_row][0] + mi_[sl] * 16 + tr_col));
                        av_keep[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)((unsigned short *)&s_a[buf][tr_row + 16][0] + mi_[sl] * 16 + tr_col));
                    }
                    const s16x4 bv0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)((unsigned short *)&s_b[buf][tr_row][0] + ni_[sl] * 16 + tr_col));
                    const s16x4 bv1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)((unsigned short *)&s_b[buf][tr_row + 16][0] + ni_[sl] * 16 + tr_col));
                    const s16x8 a8 = __builtin_shufflevector(av_keep[0], av_keep[1], 0, 1, 2, 3, 4, 5, 6, 7);
                    const s16x8 b8 = __builtin_shufflevector(bv0, bv1, 0, 1, 2, 3, 4, 5, 6, 7);
                    if constexpr (PREC == 1) {
                        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
                        acc[sl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), acc[sl], 0, 0, 0);
                    } else {
                        typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
                        acc[sl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a8), __builtin_bit_cast(h16x8, b8), acc[sl], 0, 0, 0);
                    }
                }
                if constexpr (sl * PER + 0 < 10) piece(std::integral_constant<int, sl * PER + 0>());
                if constexpr (PER > 1 && sl * PER + 1 < 10) piece(std::integral_constant<int, sl * PER + 1>());
                if constexpr (PER > 2 && sl * PER + 2 < 10) piece(std::integral_constant<int, sl * PER + 2>());
                if constexpr (PER > 3 && sl * PER + 3 < 10) piece(std::integral_constant<int, sl * PER + 3>());
                if constexpr (PER > 4 && sl * PER + 4 < 10) piece(std::integral_constant<int, sl * PER + 4>());
                if constexpr (PER > 5 && sl * PER + 5 < 10) piece(std::integral_constant<int, sl * PER + 5>());
                if constexpr (PER > 6 && sl * PER + 6 < 10) piece(std::integral_constant<int, sl * PER + 6>());
                if constexpr (PER > 7 && sl * PER + 7 < 10) piece(std::integral_constant<int, sl * PER + 7>());
                if constexpr (PER > 8 && sl * PER + 8 < 10) piece(std::integral_constant<int, sl * PER + 8>());
                if constexpr (PER > 9 && sl * PER + 9 < 10) piece(std::integral_constant<int, sl * PER + 9>());
            };
            group(I0());
            if constexpr (S > 1) group(I1());
            if constexpr (S > 2) group(I2());
            if constexpr (S > 3) group(std::integral_constant<int, 3>());
            if constexpr (S > 4) group(std::integral_constant<int, 4>());
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        for (int b = 0; b < nbatch; b += 2) {
            step(b, I0());
            if (b + 1 < nbatch) step(b + 1, I1());
        }
    }
    // accumulate this block's partial into dw[o][ci][co]; C layout: col = lane&15, row = q*4+i.  Two-stage mode (slab):
    // the partial is STORED into slab[part][o][ci][co], part = (row chunk, wave-partial of a split batch) -- every element
    // of a part is written by exactly one wave of the grid; k_dw2_reduce adds the parts in a fixed order (no atomics:
    // bitwise reproducible)
    const long wn = (long)gridDim.y * cin * cout;
    float *dst = slab ? slab + ((long)chunk_id * WPB + (KT > 32 ? wave / nblk : 0)) * wn : dw;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (KT > 32 || (rowmode ? s < ni_n : wave + 4 * s < nblk)) {   // split batches: every wave holds a partial of its block
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ci = ci0 + mi_[s] * 16 + q * 4 + i, co = co0 + ni_[s] * 16 + m;
                if (slab) dst[((long)o * cin + ci) * cout + co] = acc[s][i];
                else atomicAdd(&dst[((long)o * cin + ci) * cout + co], acc[s][i]);
            }
        }
    }
}

// dw[e] += slab[0][e] + slab[1][e] + ... (fixed order), eight independent loads in flight per thread
__global__ void k_dw2_reduce(const float *__restrict__ slab, int P, long n, float *__restrict__ dw)
{
    const long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e >= n) return;             // n = K * cin * cout with cin, cout multiples of 16
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= P; s += 8) {
        f32x4 p[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) p[k] = *(const f32x4 *)(slab + (long)(s + k) * n + e);
#pragma unroll
        for (int k = 0; k < 8; ++k) v += p[k];
    }
    for (; s < P; ++s) v += *(const f32x4 *)(slab + (long)s * n + e);
    *(f32x4 *)(dw + e) += v;
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
        if (rl < R) {
            // four rows per pass with their table words, then their operands, requested together (one dependent chain per row was
            // 24 round trips per thread: 26 us for the 50k-row stem, the last weight gradient the optimizer waits for)
            long j = row_begin + rl;
            for (; j + 3 * R < row_end; j += 4 * R) {
                int i4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) i4[u] = tbl[(long)o * ld + j + (long)u * R];
                float xv[4], gv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xv[u] = x[(long)(i4[u] >= 0 ? i4[u] : 0) * cin + ci];
                    gv[u] = dy[(j + (long)u * R) * cout + co];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i4[u] >= 0) acc = fmaf(xv[u], gv[u], acc);
            }
            for (; j < row_end; j += R) {
                const int i = tbl[(long)o * ld + j];
                if (i >= 0) acc = fmaf(x[(long)i * cin + ci], dy[j * cout + co], acc);
            }
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

extern int g_opt_precision;
int g_dw_split = 2;       // split-batch variants of k_gconv_dw2 for one- and two-block tiles: 0 never, 1 automatic, 2 always (urn_set_option "dw_split")
int g_dw_rowmode = 1;     // full-width tiles: wave w owns input-channel block w and all output blocks (urn_set_option "dw_rowmode": 0 never, 1 with 16-bit operands, 2 always)
int g_dw_kernel = 2;      // 2 = k_gconv_dw2 (default), 1 = k_gconv_dw (urn_set_option "dw_kernel")
int g_dw_blocks = 768;    // target number of workgroups of the weight-gradient kernel (urn_set_option "dw_blocks"); alone 2048 is the optimum; beside the dX chain fewer are better (round 1: 3.44 ms per cfg3 step at 1024-1280 against 3.50 at 1536 and 3.57 at 2048; round 2, pair-list dX kernels: 3.20 at 768 = three per CU, 3.22 at 640, 3.23 at 896, 3.24 at 1024-1152, 3.33 at 512, 4.07 at 256 where the side stream becomes the critical path)

extern "C" int urn_gconv_bwd_dw(const float *x, const float *dy, const int32_t *tbl, int64_t ld, int K,
                                int64_t n_out, int cin, int cout, float *dw, void *stream)
{
    return urn_gconv_bwd_dw_ex(x, nullptr, nullptr, dy, tbl, ld, K, n_out, cin, cout, dw, stream);
}

extern "C" int urn_gconv_bwd_dw_ex(const float *x, const float *xf_scale, const float *xf_shift, const float *dy,
                                   const int32_t *tbl, int64_t ld, int K, int64_t n_out, int cin, int cout,
                                   float *dw, void *stream)
{
    return urn_gconv_bwd_dw_strided(x, xf_scale, xf_shift, dy, cout, tbl, ld, K, n_out, cin, cout, dw, stream);
}

// launch plan of k_gconv_dw2: row chunks, rows per chunk, partials per chunk (the wave-partials of the split-batch variants)
static void dw2_plan(int K, int64_t n_out, int cin, int cout, int &chunks, long &chunk, int &wpb)
{
    const int n_ci_tiles = urn_cdiv(cin, DW_MAXI * 16), n_co_tiles = urn_cdiv(cout, DW_MAXN * 16);
    // aim for ~2048 blocks, chunks of at least 256 rows (multiple of 256)
    // (the tuned target holds up to ~100k rows; launches over larger levels own more of the chip: 4 events per GPU
    // measured 9.5 ms per step with 2048 against 10.0 with 1152)
    // ... and more of them for bigger launches (rows x cin x cout; every cfg3 launch is under 40 M): 2 and 4 events per GPU and
    // the cfg5 shapes (768^3, 200k voxels, uf 32) like 1152-1536 (cfg5 fp16 step 22.0 -> 21.0 ms, fp32 29.6 -> 27.9)
    const double work = (double)n_out * cin * cout;
    int dw_target = g_dw_blocks;
    if (g_dw_blocks == 768) dw_target = work < 40e6 ? 768 : (work < 160e6 ? 1152 : 1536);
    if (n_out >= 150000 && dw_target < 2048) dw_target = 2048;
    chunks = dw_target / (K * n_ci_tiles * n_co_tiles);
    if (chunks < 1) chunks = 1;
    chunk = (n_out + chunks - 1) / chunks;
    if (chunk < 256) chunk = 256;
    chunk = ((chunk + 255) / 256) * 256;
    chunks = (int)((n_out + chunk - 1) / chunk);
    const int ci_w = cin < DW_MAXI * 16 ? cin : DW_MAXI * 16, co_w = cout < DW_MAXN * 16 ? cout : DW_MAXN * 16;
    const int nblk_max = (ci_w / 16) * (co_w / 16);
    const bool split = g_dw_split == 2 || (g_dw_split == 1 && n_out >= 131072);
    wpb = (nblk_max == 1 && split) ? 4 : ((nblk_max == 2 && split) ? 2 : 1);
}

static int dw_launch(const float *x, const float *xf_scale, const float *xf_shift, const float *dy, int64_t ld_dy, const int32_t *tbl,
                     int64_t ld, int K, int64_t n_out, int cin, int cout, float *dw, float *slab, int64_t slab_bytes, void *stream);

extern "C" int64_t urn_gconv_dw_2stage_scratch_bytes(int K, int64_t n_out, int cin, int cout)
{
    if (K <= 0 || n_out < 0 || cin <= 0 || cout <= 0 || cin % 16 || cout % 16) return -1;
    int chunks, wpb; long chunk;
    dw2_plan(K, n_out > 0 ? n_out : 1, cin, cout, chunks, chunk, wpb);
    // (the plan depends on the options dw_blocks / dw_split: sized for the largest split either way)
    return (int64_t)chunks * 4 * K * cin * cout * 4;
}

// upper bound over every shape: parts x K x cin x cout = workgroups x tile floats x wave-partials; the plan launches at most
// max(dw_blocks, 2048) + K * tiles workgroups of at most DW_MAXI * 16 x DW_MAXN * 16 floats (x 4 only for 16 x 16 tiles)
extern "C" int64_t urn_gconv_dw_2stage_scratch_max(void)
{
    const int64_t wgs = (g_dw_blocks > 2048 ? g_dw_blocks : 2048) + 27 * 16;
    return wgs * (DW_MAXI * 16) * (DW_MAXN * 16) * 4;
}

extern "C" int urn_gconv_bwd_dw_2stage(const float *x, const float *xf_scale, const float *xf_shift, const float *dy, int64_t ld_dy,
                                       const int32_t *tbl, int64_t ld, int K, int64_t n_out, int cin, int cout, float *dw,
                                       void *scratch, int64_t scratch_bytes, void *stream)
{
    URN_CHECK_ARG(scratch && cin % 16 == 0 && cout % 16 == 0, "two-stage weight gradient: scratch and channel counts that are multiples of 16");
    return dw_launch(x, xf_scale, xf_shift, dy, ld_dy, tbl, ld, K, n_out, cin, cout, dw, (float *)scratch, scratch_bytes, stream);
}

extern "C" int urn_gconv_bwd_dw_strided(const float *x, const float *xf_scale, const float *xf_shift, const float *dy,
                                        int64_t ld_dy, const int32_t *tbl, int64_t ld, int K, int64_t n_out, int cin,
                                        int cout, float *dw, void *stream)
{
    return dw_launch(x, xf_scale, xf_shift, dy, ld_dy, tbl, ld, K, n_out, cin, cout, dw, nullptr, 0, stream);
}

static int dw_launch(const float *x, const float *xf_scale, const float *xf_shift, const float *dy, int64_t ld_dy, const int32_t *tbl,
                     int64_t ld, int K, int64_t n_out, int cin, int cout, float *dw, float *slab, int64_t slab_bytes, void *stream)
{
    URN_CHECK_ARG(ld_dy >= cout && ld_dy % 4 == 0, "ld_dy smaller than the row or not a multiple of 4");
    if (n_out <= 0) return URN_OK;
    URN_CHECK_ARG(x && dy && tbl && dw, "null pointer");
    URN_CHECK_ARG(K > 0 && cin > 0 && cout > 0 && ld >= n_out, "bad shape");
    URN_CHECK_ARG((xf_scale == nullptr) == (xf_shift == nullptr), "scale and shift go together");
    hipStream_t st = (hipStream_t)stream;
    if (ld_dy != cout && (g_dw_kernel != 2 || (cin % 16) || (cout % 16))) {
        urn_set_error("urn_gconv_bwd_dw_strided: a strided dy needs k_gconv_dw2 (channel counts that are multiples of 16)");
        return URN_EUNSUPPORTED;
    }
    if ((cin % 16) || (cout % 16)) {
        if (xf_scale) { urn_set_error("urn_gconv_bwd_dw_ex: input transform needs channel counts that are multiples of 16"); return URN_EUNSUPPORTED; }
        int chunks = (int)((n_out + 255) / 256);
        if (chunks > 256) chunks = 256;
        long chunk = (n_out + chunks - 1) / chunks;
        hipLaunchKernelGGL(k_gconv_dw_small, dim3(chunks, K), dim3(256), 0, st, x, dy, tbl, (long)ld, (long)n_out,
                           cin, cout, chunk, dw);
        URN_LAUNCH_CHECK();
        return URN_OK;
    }
    const int n_ci_tiles = urn_cdiv(cin, DW_MAXI * 16), n_co_tiles = urn_cdiv(cout, DW_MAXN * 16);
    int chunks, wpb; long chunk;
    dw2_plan(K, n_out, cin, cout, chunks, chunk, wpb);
    const long wn = (long)K * cin * cout;
    if (slab && (g_dw_kernel != 2 || slab_bytes < (int64_t)chunks * wpb * wn * 4)) {
        urn_set_error("urn_gconv_bwd_dw_2stage: scratch smaller than urn_gconv_dw_2stage_scratch_bytes");
        return URN_EINVAL;
    }
    const bool prof = urn_prof_on();
    if (prof) urn_prof_begin(URN_PROF_DW, st);
    const dim3 grid(chunks, K, n_ci_tiles * n_co_tiles);
    const int prec = g_opt_precision;   // 0 fp32, 1 bf16, 2 fp16 (urn_set_option "gconv_precision")
    const int row_ok = g_dw_rowmode == 2 || (g_dw_rowmode == 1 && prec != 0);
    if (g_dw_kernel == 2) {
        const int ci_w = cin < DW_MAXI * 16 ? cin : DW_MAXI * 16, co_w = cout < DW_MAXN * 16 ? cout : DW_MAXN * 16;
        const int nblk_max = (ci_w / 16) * (co_w / 16);   // of the widest tile
        const int slots = (nblk_max + 3) / 4;
        // one- and two-block tiles exist only when the whole layer is that narrow (cin, cout <= 32): then every tile
        // of the grid has the same shape and the split-batch variants apply
#define URN_DW2P(Sv, XFv, KTv)                                                                                                       \
        do {                                                                                                                         \
            if (prec == 1) hipLaunchKernelGGL((k_gconv_dw2<Sv, XFv, KTv, 1>), grid, dim3(256), 0, st, x, xf_scale, xf_shift, dy, tbl, \
                                              (long)ld, (long)n_out, cin, cout, chunk, n_ci_tiles, dw, (long)ld_dy, slab, row_ok);        \
            else if (prec == 2) hipLaunchKernelGGL((k_gconv_dw2<Sv, XFv, KTv, 2>), grid, dim3(256), 0, st, x, xf_scale, xf_shift, dy, \
                                                   tbl, (long)ld, (long)n_out, cin, cout, chunk, n_ci_tiles, dw, (long)ld_dy, slab, row_ok); \
            else hipLaunchKernelGGL((k_gconv_dw2<Sv, XFv, KTv, 0>), grid, dim3(256), 0, st, x, xf_scale, xf_shift, dy, tbl, (long)ld, \
                                    (long)n_out, cin, cout, chunk, n_ci_tiles, dw, (long)ld_dy, slab, row_ok);                            \
        } while (0)
#define URN_DW2K(Sv, KTv)                                                                                                            \
        if (xf_scale) URN_DW2P(Sv, 1, KTv);                                                                                          \
        else URN_DW2P(Sv, 0, KTv);
#define URN_DW2(Sv) case Sv: URN_DW2K(Sv, 32) break;
        // The split variants keep all four matrix pipes of a CU busy: faster alone (16 x 16 at 50k rows: 29 -> 24 us;
        // dense 128^3 model: 60.6 -> 49.6 ms per step).  Beside the dX chain of the sparse executor they were first
        // 1.4 % slower per step (hence the automatic mode: only launches of >= 128k rows), with the final conv
        // kernels 0.5 % faster (3.510 vs 3.527 ms) -> default: always.
        const bool split = g_dw_split == 2 || (g_dw_split == 1 && n_out >= 131072);
        if (nblk_max == 1 && split) { URN_DW2K(1, 128) }
        else if (nblk_max == 2 && split) { URN_DW2K(1, 64) }
        else switch (slots) { URN_DW2(1) URN_DW2(2) URN_DW2(3) URN_DW2(4) URN_DW2(5) default: break; }
#undef URN_DW2K
#undef URN_DW2P
#undef URN_DW2
    } else {
        hipLaunchKernelGGL(k_gconv_dw, grid, dim3(256), 0, st, x, xf_scale, xf_shift, dy, tbl, (long)ld,
                           (long)n_out, cin, cout, chunk, n_ci_tiles, dw);
    }
    if (slab) hipLaunchKernelGGL(k_dw2_reduce, dim3(urn_cdiv(wn / 4, 256)), dim3(256), 0, st, (const float *)slab, chunks * wpb, wn, dw);
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

// wt (K, cout, cin) -> fragment order (urn_gconv_args.wt_frag): thread = one 16-byte piece of the destination (coalesced
// writes; the source rows are L2-resident weights)
__global__ void k_weight_fragments(const float *__restrict__ wt, long total4, int cout, int cin, float *__restrict__ wf)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total4) return;
    const int lane = (int)(e & 63), r = lane & 15, q = lane >> 4;
    long f = e >> 6;                         // fragment index = (o * (cout / 16) + cb) * (cin / 16) + kb
    const int kbn = cin / 16, cbn = cout / 16;
    const int kb = (int)(f % kbn); f /= kbn;
    const int cb = (int)(f % cbn); const long o = f / cbn;
    *(f32x4 *)(wf + e * 4) = *(const f32x4 *)(wt + ((long)o * cout + 16 * cb + r) * cin + 16 * kb + 4 * q);
}

extern "C" int urn_weight_fragments(const float *wt, int K, int cout, int cin, float *wt_frag, void *stream)
{
    URN_CHECK_ARG(wt && wt_frag && K > 0 && cout > 0 && cin > 0 && cout % 16 == 0 && cin % 16 == 0, "channel counts must be multiples of 16");
    const long total4 = (long)K * cout * cin / 4;
    hipLaunchKernelGGL(k_weight_fragments, dim3(urn_cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, wt, total4, cout, cin, wt_frag);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

// the same with 16-bit elements: thread = the 8 bytes of one lane
template <int PREC>
__global__ void k_weight_fragments16(const float *__restrict__ wt, long total4, int cout, int cin, uint2 *__restrict__ wf)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total4) return;
    const int lane = (int)(e & 63), r = lane & 15, q = lane >> 4;
    long f = e >> 6;
    const int kbn = cin / 16, cbn = cout / 16;
    const int kb = (int)(f % kbn); f /= kbn;
    const int cb = (int)(f % cbn); const long o = f / cbn;
    wf[urn_frag16_slot(e, kbn)] = urn_round16x4<PREC>(*(const f32x4 *)(wt + ((long)o * cout + 16 * cb + r) * cin + 16 * kb + 4 * q));
}

extern "C" int urn_weight_fragments16(const float *wt, int K, int cout, int cin, int precision, void *wt_frag16, void *stream)
{
    URN_CHECK_ARG(wt && wt_frag16 && K > 0 && cout > 0 && cin > 0 && cout % 16 == 0 && cin % 16 == 0, "channel counts must be multiples of 16");
    URN_CHECK_ARG(precision == 1 || precision == 2, "precision: 1 bf16, 2 fp16");
    const long total4 = (long)K * cout * cin / 4;
    if (precision == 1) hipLaunchKernelGGL(k_weight_fragments16<1>, dim3(urn_cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, wt, total4, cout, cin, (uint2 *)wt_frag16);
    else hipLaunchKernelGGL(k_weight_fragments16<2>, dim3(urn_cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, wt, total4, cout, cin, (uint2 *)wt_frag16);
    URN_LAUNCH_CHECK();
    return URN_OK;
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
