// Gather convolution, 2-D workgroup tile: RB row blocks x CB column blocks of 16x16 outputs, one wave each.
//
// In-kernel cycle stamps on the 64x16 tile kernel (urn_gconv_lds.hip) showed ~80% of an offset step in
// issuing the gathers and waiting for them: the step is bound by the per-CU gather rate (~23 B/clk/CU
// measured, MI355X_MICROARCH.md "Indexed rows: gather into LDS" gives 14-30 B/clk/CU), and that tile moves
// A + B/4 operand bytes per wave-step.  Here the RB*CB waves of a workgroup share BOTH operands through
// LDS: the 16*RB gathered rows are fetched once for the CB column blocks, the CB*16-column weight tile once
// for the RB row blocks, i.e. A/CB + B/RB bytes per wave-step, with the same number of waves on the chip.
// Arithmetic, determinism, epilogues and the partial-slab layout (one row per workgroup) are unchanged.
#include "urn_common.h"
#include "urn_gconv_int.h"
// timing-only ablation switches exist in -DURN_DIAG builds only (the product kernels carry no diagnostic branch)
#ifdef URN_DIAG
#define URN_TILE_DBG(g, bits) ((g).dbg & (bits))
#else
#define URN_TILE_DBG(g, bits) 0
#endif
#include <type_traits>

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// f(integral_constant<I>) for I in [B, E): loop bodies that index register arrays need compile-time indices
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>());
        static_for<B + 1, E>(f);
    }
}

// ABL: timing-only ablation of the probe instantiations (1 no MFMA, 2 no global loads, 4 no LDS parking, 8 no barrier)
// IL: straight-line step (no branches) with the loads of step +2 and the parking of step +1 interleaved between the
//     MFMAs of the current step; 1 = plain input rows, 2 = relu(x*scale+shift) on the input rows.  Needs D == 2.
// PREC: operand precision of the MFMAs (interleaved variant only): 0 = fp32 (v_mfma_f32_16x16x4_f32), 1 = bf16, 2 = fp16
//       (one v_mfma_f32_16x16x16 per 16-channel group): rows and weights are read as fp32 from HBM and rounded (RNE)
//       while they are parked in LDS, accumulation stays fp32 -- BASELINE configs[1] (bf16) and configs[4] (fp16).
template <int KS, int RB, int CB, int D, int STAMP = 0, int ABL = 0, int IL = 0, int PREC = 0>
__global__ __launch_bounds__(64 * RB * CB) void k_gconv_tile(GArgs g)
{
    constexpr int CIN = KS * 16;                                    // channels per step (a chunk of g.cin)
    constexpr int LDA = CIN + 4;
    constexpr int T = 64 * RB * CB;                                 // threads
    constexpr int A_TOT = RB * 16 * CIN / 4, B_TOT = CB * 16 * CIN / 4;   // float4 per operand tile
    constexpr int A_F4 = (A_TOT + T - 1) / T, B_F4 = (B_TOT + T - 1) / T;
    __shared__ int s_idx[RB][28 * 16];
    __shared__ unsigned s_mask[RB];
    __shared__ float s_xf[2][512];                                  // folded BatchNorm affine of all input channels
    static_assert(PREC == 0 || IL != 0, "reduced precision is implemented in the interleaved variant");
    constexpr int LDS_LD = PREC ? CIN / 2 + 4 : LDA;                // floats per LDS row (16-bit rows: CIN + 8 halves)
    __shared__ __attribute__((aligned(16))) float s_a[2][RB * 16][LDS_LD];
    __shared__ __attribute__((aligned(16))) float s_b[2][CB * 16][LDS_LD];
    // store four consecutive channels of an operand row in the LDS image (fp32, or rounded to 16 bits)
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
    __shared__ double s_p[2][RB][CB * 16];

    const long n_out = g.n_dev ? (long)*g.n_dev : g.n_cap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rb = wave / CB, cb = wave - rb * CB;
    const int r = lane & 15, q = lane >> 4;
    // XCD-aware row-tile assignment: workgroups are dealt to the 8 XCDs round-robin (linear id % 8), so consecutive
    // row tiles would land on 8 different L2s and every XCD would fetch (nearly) the whole input.  Give XCD x a
    // contiguous range of tiles instead: rows that are neighbours in space are mostly neighbours in the site order.
    unsigned tile_x = blockIdx.x;
    if (!URN_TILE_DBG(g, 64) && gridDim.y == 1) {
        const unsigned nb = gridDim.x, xq = nb >> 3, xr = nb & 7u, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        tile_x = xcd * xq + (xcd < xr ? xcd : xr) + slot;
    }
    const long row_base = ((long)tile_x * RB + rb) * 16;
    const long tile_row0 = (long)tile_x * RB * 16;
    const int col_base = (blockIdx.y * CB + cb) * 16;
    const int tile_col0 = blockIdx.y * CB * 16;
    const int K = g.K, cout = g.cout;
    const bool xs = g.xs_sums[0] != nullptr;                        // folded affine derived here from accumulated sums
    const bool xf = g.xf_scale != nullptr || xs;

    // 1. table fetch (the cb == 0 wave of every row block), masks.  The table loads are issued first and consumed
    //    after the BatchNorm coefficients below, so that the two global round trips of the prologue overlap.
    int tv[7];
    if (cb == 0) {
        const long row = row_base + r;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int t = 4 * i + q;
            tv[i] = -1;
            if (t < K && row < n_out) tv[i] = g.tbl[(long)t * g.ld + row];
        }
    }
    if (xs) {
        // BatchNorm statistics of the input rows: the producers accumulated (sum, sum of squares) into xs_slots rows;
        // same arithmetic as k_bn_finalize_fwd_f.  Workgroup 0 keeps the results for the backward pass.
        const bool keep = blockIdx.x == 0 && blockIdx.y == 0;
        const double inv_n = g.xs_n > 0 ? 1.0 / (double)g.xs_n : 0.0;
        for (int e = tid; e < g.cin; e += T) {
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
            s_xf[0][e] = sc; s_xf[1][e] = sh;
            if (keep) {
                g.xs_mean[e] = (float)mu; g.xs_invstd[e] = (float)is; g.xs_scale[e] = sc; g.xs_shift[e] = sh;
                if (g.xs_rm) g.xs_rm[e] = (float)(g.fin_momentum * g.xs_rm[e] + (1.0 - g.fin_momentum) * mu);
                if (g.xs_rv) g.xs_rv[e] = (float)(g.fin_momentum * g.xs_rv[e] + (1.0 - g.fin_momentum) * var);
            }
        }
    } else if (xf) {
        for (int e = tid; e < g.cin; e += T) { s_xf[0][e] = g.xf_scale[e]; s_xf[1][e] = g.xf_shift[e]; }
    }
    if (cb == 0) {
        unsigned amask = 0u;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int v = tv[i];
            s_idx[rb][i * 64 + lane] = v;  // == [t][r]
            const unsigned long long b = __ballot(v >= 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((b >> (16 * j)) & 0xFFFFull) amask |= 1u << (4 * i + j);
        }
        if (lane == 0) s_mask[rb] = amask;
    }
    __syncthreads();
    unsigned m = 0u;
#pragma unroll
    for (int i = 0; i < RB; ++i) m |= s_mask[i];
    const unsigned my_mask = s_mask[rb];
    if (URN_TILE_DBG(g, 16)) m = 0u;

    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
    // operands of the epilogue (residual, BatchNorm input of the backward reduce) are requested now: their round trip
    // is hidden behind the offset loop instead of being exposed after it
    float pre_res[4] = {0.f, 0.f, 0.f, 0.f}, pre_ex[4] = {0.f, 0.f, 0.f, 0.f};
    float esc = 0.f, esh = 0.f, emu = 0.f, eis = 0.f;
    {
        const int col = col_base + r;
        if (g.epi == 2) { esc = g.e_scale[col]; esh = g.e_shift[col]; emu = g.e_mean[col]; eis = g.e_invstd[col]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = min(row_base + q * 4 + i, n_out - 1);
            const long off = row * cout + col;
            if (g.res && !STAMP) pre_res[i] = g.res[off];
            if (g.epi == 2) pre_ex[i] = g.e_x[off];
        }
    }
    f32x4 ra[D][A_F4], rb_[D][B_F4];   // register ring: operands of the next D steps, in flight
    const int cin = g.cin, nch = cin / CIN;   // input-channel chunks per offset
    auto fetch = [&](int t, int ch, int gi) {
        if constexpr (ABL & 2) {
#pragma unroll
            for (int j = 0; j < A_F4; ++j) ra[gi][j] = (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int j = 0; j < B_F4; ++j) rb_[gi][j] = (f32x4){1.f, 1.f, 1.f, 1.f};
            return;
        }
        const int o = g.flip ? (K - 1 - t) : t;
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            // threads past the end of a tile repeat its last element: every load is unconditional, so that the
            // compiler can count the loads in flight (s_waitcnt vmcnt(N)) instead of draining them
            const int e = (A_TOT % T == 0) ? j * T + tid : min(j * T + tid, A_TOT - 1);
            const int row = e / (CIN / 4), c4 = e - row * (CIN / 4);
            const int idx = s_idx[row >> 4][t * 16 + (row & 15)];
            ra[gi][j] = *(const f32x4 *)(g.x + (long)(idx < 0 ? 0 : idx) * g.ldx + ch * CIN + 4 * c4);
        }
        const float *wo = g.wt + ((long)o * cout + tile_col0) * cin + ch * CIN;
#pragma unroll
        for (int j = 0; j < B_F4; ++j) {
            const int e = (B_TOT % T == 0) ? j * T + tid : min(j * T + tid, B_TOT - 1);
            const int col = e / (CIN / 4), c4 = e - col * (CIN / 4);
            rb_[gi][j] = *(const f32x4 *)(wo + (long)col * cin + 4 * c4);
        }
    };
    auto park = [&](int t, int ch, int buf, int gi) {
        if constexpr (ABL & 4) {   // wait for the data, write nothing
#pragma unroll
            for (int j = 0; j < A_F4; ++j) asm volatile("" ::"v"(ra[gi][j]));
#pragma unroll
            for (int j = 0; j < B_F4; ++j) asm volatile("" ::"v"(rb_[gi][j]));
            return;
        }
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            const int e = j * T + tid;
            if (A_TOT % T == 0 || e < A_TOT) {
                const int row = e / (CIN / 4), c4 = e - row * (CIN / 4);
                const bool have = s_idx[row >> 4][t * 16 + (row & 15)] >= 0;
                f32x4 v = ra[gi][j];
                if (xf) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaf(v[k], s_xf[0][ch * CIN + 4 * c4 + k], s_xf[1][ch * CIN + 4 * c4 + k]), 0.f);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = have ? v[k] : 0.f;
                lds_store(&s_a[buf][row][0], c4, v);
            }
        }
#pragma unroll
        for (int j = 0; j < B_F4; ++j) {
            const int e = j * T + tid;
            if (B_TOT % T == 0 || e < B_TOT) {
                const int col = e / (CIN / 4), c4 = e - col * (CIN / 4);
                lds_store(&s_b[buf][col][0], c4, rb_[gi][j]);
            }
        }
    };

    // 2. offset loop.  A step is one (active offset, channel chunk) pair; nch == 1 for cin <= 112.  A step with one
    //    16..64-channel offset is too thin to cover a memory round trip (4*KS MFMAs per wave), so the operands of
    //    the next D steps are kept in flight in a register ring (slot u is refilled as soon as its step has been
    //    parked in LDS); LDS holds the current and the next step; one LDS-only barrier per step.
    int t_run = -1, c_run = 0;   // walker over (active offset, chunk)
    auto take = [&](int &ts, int &cs) {
        ts = -1; cs = 0;
        if (t_run >= 0 && c_run + 1 < nch) { ++c_run; ts = t_run; cs = c_run; }
        else if (m) { t_run = __builtin_ctz(m); m &= m - 1u; c_run = 0; ts = t_run; }
        else t_run = -1;
    };
    int rt[D], rc[D];
    int buf = 0;
    int ct = -1;   // the step whose operands are in LDS image `buf`
    if constexpr (IL == 0) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            take(rt[i], rc[i]);
            fetch(rt[i] < 0 ? 0 : rt[i], rc[i], i);   // unconditional (a dummy step past the end): see step()
        }
        park(rt[0] < 0 ? 0 : rt[0], rc[0], 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ct = rt[0];
    }
    unsigned long long st_fetch = 0, st_mfma = 0, st_park = 0, st_bar = 0, st_n = 0;   // STAMP build only
#define URN_STAMP(v)                                                                          \
    unsigned long long v = 0;                                                                 \
    if (STAMP) {                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    }
    // one step with the ring slot as a compile-time constant (register arrays must be indexed statically)
    auto step = [&](auto slot) {
        constexpr int u = decltype(slot)::value;
        constexpr int nx = (u + 1) % D;   // slot of the next step
        URN_STAMP(c0)
        // slot u is free: its step is the one in LDS.  Loads and parks are unconditional -- past the last step they
        // re-fetch offset 0 into a buffer nobody reads -- because a branch around a load makes the compiler drain
        // vmcnt to 0 instead of counting the D-1 younger steps that may stay in flight.
        take(rt[u], rc[u]);
        fetch(rt[u] < 0 ? 0 : rt[u], rc[u], u);
        URN_STAMP(c1)   // global loads of step +D issued
        if ((my_mask >> ct) & 1u) {   // wave-uniform
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f32x4 a = *(const f32x4 *)&s_a[buf][rb * 16 + r][ks * 16 + 4 * q];
                const f32x4 b = *(const f32x4 *)&s_b[buf][cb * 16 + r][ks * 16 + 4 * q];
                if constexpr (ABL & 1) {
                    asm volatile("" ::"v"(a), "v"(b));
                } else if (ks & 1) {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc2 = MFMA16(a[tt], b[tt], acc2);
                } else {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc = MFMA16(a[tt], b[tt], acc);
                }
            }
        }
        URN_STAMP(c2)   // fragment reads + MFMAs issued
        park(rt[nx] < 0 ? 0 : rt[nx], rc[nx], buf ^ 1, nx);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        URN_STAMP(c3)   // next step parked (includes the wait for its global data)
        if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        URN_STAMP(c4)   // barrier passed
        if (STAMP) { st_fetch += c1 - c0; st_mfma += c2 - c1; st_park += c3 - c2; st_bar += c4 - c3; st_n += 1; }
        ct = rt[nx];
        buf ^= 1;
    };
    // --- interleaved variant -------------------------------------------------------------------------------------
    // Everything of a step that is not an MFMA (address arithmetic and global loads of step +2, BatchNorm/ReLU and
    // zero-fill plus the LDS writes of step +1, fragment reads of this step) is cut into pieces and placed between
    // the MFMAs of this step, in one basic block: the matrix pipe of a SIMD is then fed while its waves do the
    // bookkeeping, instead of all waves of the workgroup alternating between a bookkeeping phase and an MFMA phase.
    // Per-thread constants of the pieces: the offset loop of a narrow layer is bound by instruction issue (a wave64 VALU
    // instruction occupies the 16-lane SIMD for four cycles, and the waves of up to three workgroups share it), so
    // everything that does not change from step to step is computed once: LDS positions, 32-bit byte offsets into
    // the rows and into the weight tile (global loads in scalar-base + 32-bit-offset form), the folded BatchNorm
    // affine of the thread's four channels, and the validity of a gathered row travels with it in the ring.
    int a_idx[A_F4];                       // position of the piece's row in s_idx (offset 0)
    unsigned a_off[A_F4], b_off[B_F4];     // byte offsets inside a row / inside the weight tile of an offset
    float *a_dst[A_F4], *b_dst[B_F4];      // LDS row of the piece in image 0
    int a_c4[A_F4], b_c4[B_F4];
    f32x4 a_sc[A_F4], a_sh[A_F4];
    bool rv[D][A_F4];
    const bool xf_regs = nch == 1;         // one chunk: the affine of the thread's channels stays in registers
#pragma unroll
    for (int j = 0; j < A_F4; ++j) {
        const int e = (A_TOT % T == 0) ? j * T + tid : min(j * T + tid, A_TOT - 1);   // duplicates rewrite the same value
        const int row = e / (CIN / 4), c4 = e - row * (CIN / 4);
        a_idx[j] = (row >> 4) * (28 * 16) + (row & 15);
        a_off[j] = (unsigned)c4 * 16u;
        a_dst[j] = &s_a[0][row][0];
        a_c4[j] = c4;
        if (IL == 2 && xf_regs) { a_sc[j] = *(const f32x4 *)&s_xf[0][4 * c4]; a_sh[j] = *(const f32x4 *)&s_xf[1][4 * c4]; }
    }
#pragma unroll
    for (int j = 0; j < B_F4; ++j) {
        const int e = (B_TOT % T == 0) ? j * T + tid : min(j * T + tid, B_TOT - 1);
        const int col = e / (CIN / 4), c4 = e - col * (CIN / 4);
        b_off[j] = (unsigned)(col * cin + 4 * c4) * 4u;
        b_dst[j] = &s_b[0][col][0];
        b_c4[j] = c4;
    }
    const unsigned ldx4 = (unsigned)g.ldx * 4u;                                   // host checks rows * ldx * 4 < 2^32
    const char *const wt_tile = (const char *)(g.wt + (long)tile_col0 * cin);    // this workgroup's columns, offset 0
    const int wstride = cout * cin;                                             // floats per offset (< 2^31 / 27)
    constexpr int A_IMG = RB * 16 * LDS_LD, B_IMG = CB * 16 * LDS_LD;           // floats per LDS image
    auto piece_fetch_a = [&](int t, int ch, auto slot, auto jj) {
        constexpr int gi = decltype(slot)::value, j = decltype(jj)::value;
        const int idx = (&s_idx[0][0])[a_idx[j] + t * 16];
        rv[gi][j] = idx >= 0;
        const unsigned off = (unsigned)max(idx, 0) * ldx4 + a_off[j];
        ra[gi][j] = *(const f32x4 *)((const char *)(g.x + ch * CIN) + off);
    };
    auto piece_fetch_b = [&](int t, int ch, auto slot, auto jj) {
        constexpr int gi = decltype(slot)::value, j = decltype(jj)::value;
        const int o = g.flip ? (K - 1 - t) : t;
        const char *wo = wt_tile + (long)(o * wstride + ch * CIN) * 4;
        rb_[gi][j] = *(const f32x4 *)(wo + b_off[j]);
    };
    auto piece_park_a = [&](int t, int ch, int pbuf, auto slot, auto jj) {
        constexpr int gi = decltype(slot)::value, j = decltype(jj)::value;
        f32x4 v = ra[gi][j];
        if constexpr (IL == 2) {
            f32x4 sc, sh;
            if (xf_regs) { sc = a_sc[j]; sh = a_sh[j]; }
            else { sc = *(const f32x4 *)&s_xf[0][ch * CIN + 4 * a_c4[j]]; sh = *(const f32x4 *)&s_xf[1][ch * CIN + 4 * a_c4[j]]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaf(v[k], sc[k], sh[k]), 0.f);
        }
        const bool have = rv[gi][j];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = have ? v[k] : 0.f;
        lds_store(a_dst[j] + pbuf * A_IMG, a_c4[j], v);
    };
    auto piece_park_b = [&](int pbuf, auto slot, auto jj) {
        constexpr int gi = decltype(slot)::value, j = decltype(jj)::value;
        lds_store(b_dst[j] + pbuf * B_IMG, b_c4[j], rb_[gi][j]);
    };
    auto step_il = [&](auto slot) {
        constexpr int u = decltype(slot)::value;
        constexpr int nx = (u + 1) % D;
        using SU = std::integral_constant<int, u>;
        using SN = std::integral_constant<int, nx>;
        // next-next step (branch-free walker; past the end it repeats offset 0, whose operands nobody reads)
        {
            const bool stay = t_run >= 0 && c_run + 1 < nch;
            const bool pop = !stay && m != 0u;
            const int nt = pop ? __builtin_ctz(m | 0x80000000u) : (stay ? t_run : -1);
            m = pop ? (m & (m - 1u)) : m;
            c_run = stay ? c_run + 1 : 0;
            t_run = nt;
            rt[u] = nt; rc[u] = c_run;
        }
        const int tf = rt[u] < 0 ? 0 : rt[u], cf = rc[u];         // to fetch into slot u
        const int tp = rt[nx] < 0 ? 0 : rt[nx], cp = rc[nx];      // to park from slot nx
        const int pbuf = buf ^ 1;
        constexpr int NP = 2 * (A_F4 + B_F4);                      // pieces: fetch A.., fetch B.., park A.., park B..
        auto piece = [&](auto pi) {
            constexpr int i = decltype(pi)::value;
            if constexpr (i < A_F4) piece_fetch_a(tf, cf, SU(), std::integral_constant<int, i>());
            else if constexpr (i < A_F4 + B_F4) piece_fetch_b(tf, cf, SU(), std::integral_constant<int, i - A_F4>());
            else if constexpr (i < 2 * A_F4 + B_F4) piece_park_a(tp, cp, pbuf, SN(), std::integral_constant<int, i - A_F4 - B_F4>());
            else if constexpr (i < NP) piece_park_b(pbuf, SN(), std::integral_constant<int, i - 2 * A_F4 - B_F4>());
        };
        // (issuing all loads first behind a sched_barrier, or pinning the pieces to their group, measured 5-15 % slower
        // than leaving the order inside the block to the compiler)
        constexpr int P0 = 0;
        constexpr int PER = (NP - P0 + KS - 1) / KS;               // pieces after each group of four MFMAs
        auto group = [&](auto kk) {
            constexpr int ks = decltype(kk)::value;
            if constexpr (PREC == 0) {
                const f32x4 a = *(const f32x4 *)&s_a[buf][rb * 16 + r][ks * 16 + 4 * q];
                const f32x4 b = *(const f32x4 *)&s_b[buf][cb * 16 + r][ks * 16 + 4 * q];
                // two accumulators, alternating per MFMA: consecutive MFMAs are independent (measured 22 vs 24 us at
                // level 3, 64 -> 64, against one accumulator per group of four)
                acc = MFMA16(a[0], b[0], acc); acc2 = MFMA16(a[1], b[1], acc2);
                acc = MFMA16(a[2], b[2], acc); acc2 = MFMA16(a[3], b[3], acc2);
            } else {
                // 16-bit operands: lane (r, q) holds the four k-values 4q..4q+3 of the 16-channel group, one MFMA per group
                typedef short s16x4 __attribute__((ext_vector_type(4)));
                const s16x4 a = *(const s16x4 *)((const unsigned short *)&s_a[buf][rb * 16 + r][0] + ks * 16 + 4 * q);
                const s16x4 b = *(const s16x4 *)((const unsigned short *)&s_b[buf][cb * 16 + r][0] + ks * 16 + 4 * q);
                if constexpr (PREC == 1) {
                    if constexpr (ks & 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc2, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
                } else {
                    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
                    const h16x4 ah = __builtin_bit_cast(h16x4, a), bh = __builtin_bit_cast(h16x4, b);
                    if constexpr (ks & 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, acc2, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, acc, 0, 0, 0);
                }
            }
            static_for<P0 + ks * PER, (P0 + (ks + 1) * PER < NP ? P0 + (ks + 1) * PER : NP)>(piece);
        };
        static_for<0, KS>(group);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ct = rt[nx];
        buf ^= 1;
    };
    if constexpr (IL != 0 && D == 2) {
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        take(rt[0], rc[0]);
        static_for<0, A_F4>([&](auto jj) { piece_fetch_a(max(rt[0], 0), rc[0], I0(), jj); });
        static_for<0, B_F4>([&](auto jj) { piece_fetch_b(max(rt[0], 0), rc[0], I0(), jj); });
        take(rt[1], rc[1]);
        static_for<0, A_F4>([&](auto jj) { piece_fetch_a(max(rt[1], 0), rc[1], I1(), jj); });
        static_for<0, B_F4>([&](auto jj) { piece_fetch_b(max(rt[1], 0), rc[1], I1(), jj); });
        static_for<0, A_F4>([&](auto jj) { piece_park_a(max(rt[0], 0), rc[0], 0, I0(), jj); });
        static_for<0, B_F4>([&](auto jj) { piece_park_b(0, I0(), jj); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ct = rt[0];
        while (ct >= 0) {
            step_il(std::integral_constant<int, 0>());
            if (ct < 0) break;
            step_il(std::integral_constant<int, 1>());
        }
    }
    while (IL == 0 && ct >= 0) {   // ct is workgroup-uniform
        step(std::integral_constant<int, 0>());
        if constexpr (D > 1) { if (ct < 0) break; step(std::integral_constant<int, 1>()); }
        if constexpr (D > 2) { if (ct < 0) break; step(std::integral_constant<int, 2>()); }
        if constexpr (D > 3) { if (ct < 0) break; step(std::integral_constant<int, 3>()); }
    }
#undef URN_STAMP
    if (STAMP && lane == 0 && g.e_x == nullptr && g.res != nullptr) {
        // diagnostic build: cycle sums of every wave go to the buffer passed as `res` (results are still written)
        float *o = const_cast<float *>(g.res) + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * (RB * CB) + wave) * 8;
        o[0] = (float)st_fetch; o[1] = (float)st_mfma; o[2] = (float)st_park; o[3] = (float)st_bar; o[4] = (float)st_n;
    }

    // 3. epilogue: one 16x16 block per wave.  C layout: col = lane&15, row = (lane>>4)*4 + reg
    double s0 = 0.0, s1 = 0.0;
    {
        const int col = col_base + r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = row_base + q * 4 + i;
            if (row >= n_out) continue;
            float v = acc[i] + acc2[i];
            if (g.res && !STAMP) v += pre_res[i];
            if (g.epi == 1) {
                s0 += (double)v;
                s1 += (double)v * (double)v;
            } else if (g.epi == 2) {
                const float xv = pre_ex[i];
                if (!(fmaf(xv, esc, esh) > 0.f)) v = 0.f;
                const double xh = ((double)xv - (double)emu) * (double)eis;
                s0 += (double)v;
                s1 += (double)v * xh;
            }
            g.y[row * g.ldy + col] = v;   // ldy > cout: the output is a column block of a wider matrix (channel concat)
        }
    }
    if (g.epi == 0) return;
    s0 += __shfl_xor(s0, 16); s1 += __shfl_xor(s1, 16);
    s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32);
    if (q == 0) { s_p[0][rb][cb * 16 + r] = s0; s_p[1][rb][cb * 16 + r] = s1; }
    __syncthreads();
    if (tid < CB * 16) {
        double v0 = 0.0, v1 = 0.0;
#pragma unroll
        for (int i = 0; i < RB; ++i) { v0 += s_p[0][i][tid]; v1 += s_p[1][i][tid]; }
        if (g.part_slots > 0) {   // accumulate: hardware fp64 add at the L2, no return value
            const long slot = blockIdx.x % (unsigned)g.part_slots;
            unsafeAtomicAdd(&g.part[(slot * 2 + 0) * cout + tile_col0 + tid], v0);
            unsafeAtomicAdd(&g.part[(slot * 2 + 1) * cout + tile_col0 + tid], v1);
        } else {
            g.part[((long)blockIdx.x * 2 + 0) * cout + tile_col0 + tid] = v0;
            g.part[((long)blockIdx.x * 2 + 1) * cout + tile_col0 + tid] = v1;
        }
    }
    (void)tile_row0;
}

// Ring depth of the plain (not interleaved) step: measured on the cfg3 shapes, depth 1, 2 and 4 were within 1 us of each
// other for every layer -- the offset step is not bound by load latency but by the sum of its phases
// (tools/ablate_gconv.py) -- so only depth 1 is instantiated; the interleaved step uses a two-slot ring.
// The file is compiled three times (Makefile: -DURN_TILE_PART=0/1/2), each part instantiating the kernels of some
// channel-step widths KS, so that the build parallelises; part 0 also holds the globals and the dispatcher.
#ifndef URN_TILE_PART
#define URN_TILE_PART 0
#endif
#if URN_TILE_PART == 0
#define URN_TILE_GLOBAL(decl, init) decl = init
#else
#define URN_TILE_GLOBAL(decl, init) extern decl
#endif
URN_TILE_GLOBAL(int g_tile_depth, 0);   // kept for the option table; no effect
URN_TILE_GLOBAL(int g_tile_il, 1);            // interleaved step (urn_set_option("tile_il", 0) = plain step)
URN_TILE_GLOBAL(int g_tile_min_wgs, 100);      // a tile with several row blocks must leave this many workgroups (urn_set_option "tile_min_wgs")
URN_TILE_GLOBAL(int g_tile_il_min_ks, 1);     // narrowest channel step that takes the interleaved variant (urn_set_option "tile_il_min_ks")

template <int KS, int RB, int CB>
static int launch_tile2(const GArgs &a, long n_out, hipStream_t st)
{
    const long bx = (n_out + 16 * RB - 1) / (16 * RB);
    const dim3 grid((unsigned)bx, a.cout / (16 * CB)), block(64 * RB * CB);
    {
        const bool xfm = a.xf_scale != nullptr || a.xs_sums[0] != nullptr;
        if (a.prec == 1) {
            if (xfm) hipLaunchKernelGGL((k_gconv_tile<KS, RB, CB, 2, 0, 0, 2, 1>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_gconv_tile<KS, RB, CB, 2, 0, 0, 1, 1>), grid, block, 0, st, a);
            return (int)bx;
        }
        if (a.prec == 2) {
            if (xfm) hipLaunchKernelGGL((k_gconv_tile<KS, RB, CB, 2, 0, 0, 2, 2>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_gconv_tile<KS, RB, CB, 2, 0, 0, 1, 2>), grid, block, 0, st, a);
            return (int)bx;
        }
    }
    {
        if (g_tile_il && KS >= g_tile_il_min_ks) {
            const bool xfm = a.xf_scale != nullptr || a.xs_sums[0] != nullptr;
            if (xfm) hipLaunchKernelGGL((k_gconv_tile<KS, RB, CB, 2, 0, 0, 2>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_gconv_tile<KS, RB, CB, 2, 0, 0, 1>), grid, block, 0, st, a);
            return (int)bx;
        }
    }
    hipLaunchKernelGGL((k_gconv_tile<KS, RB, CB, 1>), grid, block, 0, st, a);
    return (int)bx;
}

URN_TILE_GLOBAL(int g_tile_rb, 0); URN_TILE_GLOBAL(int g_tile_cb, 0); URN_TILE_GLOBAL(int g_tile_kc, 0);   // tuning knobs (urn_set_option "tile_rb" / "tile_cb" / "tile_kc"), 0 = automatic

// returns the number of partial rows (workgroups along the rows), 0 when the shape has no instantiation
template <int KS>
int launch_tile_ks(const GArgs &a, long n_out, int nblk, hipStream_t st)
{
    const long blocks16 = (n_out + 15) / 16;
    // Tile choice (measured sweep on MI355X, tools/bench_gconv.py): take all column blocks when there are <= 5 (the
    // gathered rows are then fetched once), as many row blocks as still leave >= ~100 workgroups, at most 16 waves,
    // and an LDS image (2 A + 2 B buffers) under 96 KiB.
    auto lds_ok = [&](int rb_, int cb_) { return (long)(rb_ + cb_) * 128 * (KS * 16 + 4) <= 98304 && rb_ * cb_ <= 12; };   // 16-wave (1024-thread) workgroups measured 1.5x slower
    int rb = 0, cb = 0;
    for (int c = (nblk <= 5 ? nblk : 4); c >= 1 && !rb; --c) {
        if (nblk % c) continue;
        // narrow inputs prefer few row blocks (sweep with the trimmed interleaved kernel, tools/bench_gconv.py narrow:
        // 16 -> 16 one row block 14 us, two 16, four 17-18; 16 -> 32 22 vs 27; 32 -> 64 one row block 35 us vs 42 with two;
        // 32 -> 32 and 32 -> 16 two row blocks)
        const int first = KS == 1 ? 1 : (KS == 2 ? (c >= 4 ? 1 : 2) : 4);
        for (int cand : {first, first == 4 ? 2 : (first == 2 ? 4 : 2), first == 1 ? 4 : 1}) {
            if (!lds_ok(cand, c)) continue;
            if (cand > 1 && (blocks16 / cand) * (nblk / c) < g_tile_min_wgs) continue;
            rb = cand; cb = c;
            break;
        }
    }
    if (!rb) return 0;
    if (g_tile_rb > 0) rb = g_tile_rb;
    if (g_tile_cb > 0 && nblk % g_tile_cb == 0) cb = g_tile_cb;
    if (!lds_ok(rb, cb)) return 0;
#ifdef URN_DIAG   // stamp builds and timing-only ablations of two shapes: diagnostic library only (tools/build_diag_lib.sh)
    if ((a.dbg & 32) && KS == 4 && a.cout == 64) {   // diagnostic stamp build of one shape
        if constexpr (KS == 4) {
            const long bx = (n_out + 31) / 32;
            hipLaunchKernelGGL((k_gconv_tile<4, 2, 4, 1, 1>), dim3((unsigned)bx, 1), dim3(512), 0, st, a);
            return (int)bx;
        }
    }
    if ((a.dbg & 32) && KS == 1 && a.cout == 16) {   // ... and of the 16 -> 16 shape (level 0)
        if constexpr (KS == 1) {
            const long bx = (n_out + 63) / 64;
            hipLaunchKernelGGL((k_gconv_tile<1, 4, 1, 1, 1>), dim3((unsigned)bx, 1), dim3(256), 0, st, a);
            return (int)bx;
        }
    }
    if ((a.dbg >> 8) && ((KS == 4 && a.cout == 64) || (KS == 1 && a.cout == 16))) {   // timing-only ablations of two shapes
        constexpr int PRB = KS == 4 ? 2 : 4, PCB = KS == 4 ? 4 : 1;
        if constexpr (KS == 4 || KS == 1) {
            const long bx = (n_out + 16 * PRB - 1) / (16 * PRB);
            const dim3 grid((unsigned)bx, 1), block(64 * PRB * PCB);
#define URN_ABL(v) case v: hipLaunchKernelGGL((k_gconv_tile<KS, PRB, PCB, 2, 0, v>), grid, block, 0, st, a); return (int)bx;
            switch (a.dbg >> 8) {
                URN_ABL(1) URN_ABL(2) URN_ABL(3) URN_ABL(4) URN_ABL(6) URN_ABL(7) URN_ABL(8) URN_ABL(14) URN_ABL(15)
            default: break;
            }
#undef URN_ABL
        }
    }
#endif
#define URN_TL(RBv, CBv) if (rb == RBv && cb == CBv) return launch_tile2<KS, RBv, CBv>(a, n_out, st);
    URN_TL(1, 1) URN_TL(2, 1) URN_TL(4, 1) URN_TL(1, 2) URN_TL(2, 2) URN_TL(4, 2) URN_TL(1, 3) URN_TL(2, 3) URN_TL(4, 3)
    URN_TL(1, 4) URN_TL(2, 4) URN_TL(4, 4) URN_TL(1, 5) URN_TL(2, 5)
#undef URN_TL
    return 0;
}

// explicit instantiations of this part; the others are declared for the dispatcher
#define URN_TILE_INST(KSv) template int launch_tile_ks<KSv>(const GArgs &, long, int, hipStream_t);
#define URN_TILE_DECL(KSv) extern template int launch_tile_ks<KSv>(const GArgs &, long, int, hipStream_t);
#if URN_TILE_PART == 0
URN_TILE_INST(1) URN_TILE_INST(2) URN_TILE_INST(3) URN_TILE_DECL(4) URN_TILE_DECL(5) URN_TILE_DECL(6) URN_TILE_DECL(7)
#elif URN_TILE_PART == 1
URN_TILE_INST(4) URN_TILE_INST(5)
#else
URN_TILE_INST(6) URN_TILE_INST(7)
#endif

#if URN_TILE_PART == 0
int urn_gconv_tile_launch(const GArgs &a, int ks, long n_out, hipStream_t st)
{
    const int nblk = a.cout / 16;
    if (a.cin > 512) return 0;   // s_xf holds 512 channels
    // channels per step: the whole row up to 112 channels, else the largest chunk (in 16s, at most 7) that divides it
    // (128 channels in one step leave LDS room for one row block only and re-read the weights per 16 rows: measured
    // 72-93 us against 57 us in two 64-channel chunks with two row blocks at level 3, 128 -> 64)
    int kc = ks;
    if (ks > 7) {
        kc = 1;
        for (int d : {7, 6, 5, 4, 3, 2})
            if (ks % d == 0) { kc = d; break; }
    }
    if (g_tile_kc > 0 && g_tile_kc <= 7 && ks % g_tile_kc == 0) kc = g_tile_kc;
    switch (kc) {
    case 1: return launch_tile_ks<1>(a, n_out, nblk, st);
    case 2: return launch_tile_ks<2>(a, n_out, nblk, st);
    case 3: return launch_tile_ks<3>(a, n_out, nblk, st);
    case 4: return launch_tile_ks<4>(a, n_out, nblk, st);
    case 5: return launch_tile_ks<5>(a, n_out, nblk, st);
    case 6: return launch_tile_ks<6>(a, n_out, nblk, st);
    case 7: return launch_tile_ks<7>(a, n_out, nblk, st);
    default: return 0;
    }
}
#endif   // URN_TILE_PART == 0
