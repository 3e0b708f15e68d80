// Gather convolution, LDS-staged tile kernel (the default forward / input-gradient kernel).
//
// Measured on MI355X, the register-gather kernel (urn_gconv_fwd.hip) is bound by the per-CU vector
// memory pipeline: every MFMA operand went through L1 as 16 separate 64-byte segments per
// wave-instruction, and the weight tile was re-fetched by every 16-row block.  This kernel keeps
// the same arithmetic (v_mfma_f32_16x16x4_f32, output stationary, no atomics, bit-reproducible) but
// moves both operands through LDS:
//   * a workgroup = 4 waves = 64 output rows x NB*16 output columns; wave w owns rows [16w, 16w+16);
//   * the union of the waves' active-offset masks drives ONE offset loop for the workgroup;
//   * per active offset the 256 threads fetch the weight tile W[o] (NB*16 x cin, contiguous rows =>
//     full 128-byte lines) ONCE into LDS, double buffered; it is shared by the 4 row blocks;
//   * each wave fetches its 16 gathered rows row-contiguously (4 full rows per wave-instruction for
//     cin = 64), applies the optional BatchNorm+ReLU transform, and parks them in its private LDS
//     block; MFMA fragments are then 16-byte LDS reads (padded rows: conflict-free up to 2-way);
//   * global loads of offset i+1 are in flight while the MFMAs of offset i issue; one barrier per
//     offset.
// Epilogues (residual add, column statistics, BatchNorm-backward reduce) are those of
// urn_gconv_fwd.hip; partial slabs are per 16-row block, n_part = ceil(n/16).
#include "urn_common.h"
#include "urn_gconv_int.h"

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int KS, int NB, int STAMP = 0>
__global__ __launch_bounds__(256) void k_gconv_lds(GArgs g)
{
    constexpr int CIN = KS * 16;
    constexpr int LDA = CIN + 4;                 // padded row (floats), keeps 16-byte alignment
    constexpr int A_F4 = KS;                     // float4 loads per lane for a wave's 16 x CIN block
    constexpr int B_F4 = (NB * 16 * CIN / 4 + 255) / 256;   // float4 loads per thread for the weight tile
    __shared__ int s_idx[4][28 * 16];
    __shared__ unsigned s_mask[4];
    __shared__ float s_xf[2][CIN];
    __shared__ __attribute__((aligned(16))) float s_a[4][16][LDA];
    __shared__ __attribute__((aligned(16))) float s_b[2][NB * 16][LDA];

    const long n_out = g.n_dev ? (long)*g.n_dev : g.n_cap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const long row_base = ((long)blockIdx.x * 4 + wave) * 16;
    const int col_base = blockIdx.y * (NB * 16);
    const int K = g.K, cout = g.cout;
    const bool xf = g.xf_scale != nullptr;

    // 1. table fetch for this wave's 16 rows, active-offset mask, union over the workgroup
    unsigned amask = 0u;
    {
        const long row = row_base + r;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int t = 4 * i + q;
            int v = -1;
            if (t < K && row < n_out) v = g.tbl[(long)t * g.ld + row];
            s_idx[wave][i * 64 + lane] = v;  // == [t][r]
            const unsigned long long b = __ballot(v >= 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((b >> (16 * j)) & 0xFFFFull) amask |= 1u << (4 * i + j);
        }
    }
    if (lane == 0) s_mask[wave] = amask;
    if (xf)
        for (int e = tid; e < CIN; e += 256) { s_xf[0][e] = g.xf_scale[e]; s_xf[1][e] = g.xf_shift[e]; }
    __syncthreads();
    unsigned m = s_mask[0] | s_mask[1] | s_mask[2] | s_mask[3];
    if (g.dbg & 16) m = 0u;

    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging registers: three offsets in flight (fetched two iterations before they are parked in LDS)
    f32x4 ra0[A_F4], ra1[A_F4], ra2[A_F4], rb0[B_F4], rb1[B_F4], rb2[B_F4];
    auto fetch = [&](f32x4 (&ra)[A_F4], f32x4 (&rb)[B_F4], int t) {
        if (t < 0) return;
        const int o = g.flip ? (K - 1 - t) : t;
        if (!(g.dbg & 2))
        // unconditional loads (a missing neighbour reads row 0 and is zeroed when parked): a branch around a
        // load makes hipcc wait for each load separately and serialises the gather
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            const int e = j * 64 + lane, row = e / (CIN / 4), c4 = e - row * (CIN / 4);
            const int idx = s_idx[wave][t * 16 + row];
            ra[j] = *(const f32x4 *)(g.x + (long)(idx < 0 ? 0 : idx) * CIN + 4 * c4);
        }
        const float *wo = g.wt + ((long)o * cout + col_base) * CIN;   // NB*16 contiguous rows of CIN floats
        if (!(g.dbg & 4))
#pragma unroll
        for (int j = 0; j < B_F4; ++j) {
            const int e = j * 256 + tid;
            if (e < NB * 16 * CIN / 4) rb[j] = *(const f32x4 *)(wo + 4 * (long)e);
        }
    };
    auto park = [&](f32x4 (&ra)[A_F4], f32x4 (&rb)[B_F4], int t, int buf) {
        if (t < 0) return;
        if ((amask >> t) & 1u) {
#pragma unroll
            for (int j = 0; j < A_F4; ++j) {
                const int e = j * 64 + lane, row = e / (CIN / 4), c4 = e - row * (CIN / 4);
                f32x4 v = ra[j];
                const bool have = s_idx[wave][t * 16 + row] >= 0;
                if (xf) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaf(v[k], s_xf[0][4 * c4 + k], s_xf[1][4 * c4 + k]), 0.f);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = have ? v[k] : 0.f;   // missing neighbours are exactly zero
                *(f32x4 *)&s_a[wave][row][4 * c4] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < B_F4; ++j) {
            const int e = j * 256 + tid;
            if (e < NB * 16 * CIN / 4) {
                const int col = e / (CIN / 4), c4 = e - col * (CIN / 4);
                *(f32x4 *)&s_b[buf][col][4 * c4] = rb[j];
            }
        }
    };
    auto next_t = [&]() -> int {
        if (!m) return -1;
        const int t = __builtin_ctz(m);
        m &= m - 1u;
        return t;
    };
    unsigned long long st_frag = 0, st_fetch = 0, st_mfma = 0, st_park = 0, st_bar = 0, st_n = 0;   // STAMP build only
    // one pipeline step: fragments of offset tc out of LDS, global fetch of the offset three ahead, MFMAs of tc,
    // then park the next offset (fetched two steps ago) into the free LDS buffers; one barrier per step
    auto step = [&](int tc, int buf, f32x4 (&ra_n)[A_F4], f32x4 (&rb_n)[B_F4], int tn, f32x4 (&ra_f)[A_F4],
                    f32x4 (&rb_f)[B_F4], int tf) {
        const bool act = (amask >> tc) & 1u;  // wave-uniform
        f32x4 fa[KS], fb[KS][NB];
        unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
#define URN_STAMP(v)                                                                          \
    if (STAMP) {                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    }
        URN_STAMP(c0)
        if (act) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                fa[ks] = *(const f32x4 *)&s_a[wave][r][ks * 16 + 4 * q];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) fb[ks][nb] = *(const f32x4 *)&s_b[buf][nb * 16 + r][ks * 16 + 4 * q];
            }
        }
        URN_STAMP(c1)     // fragment reads landed (the stamp waits lgkmcnt(0))
        fetch(ra_f, rb_f, tf);
        URN_STAMP(c2)     // global fetch of the offset three ahead issued
        if (act && !(g.dbg & 1)) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc[nb] = MFMA16(fa[ks][tt], fb[ks][nb][tt], acc[nb]);
        }
        URN_STAMP(c3)     // MFMAs issued
        park(ra_n, rb_n, tn, buf ^ 1);   // s_a[wave] is private and its fragments are already in registers
        // The barrier only orders LDS traffic.  __syncthreads() would also wait for vmcnt(0) -- on gfx950 loads and
        // stores share that counter -- and drain the two offsets of global prefetch every step.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        URN_STAMP(c4)     // next offset parked (includes the wait for its global data)
        if (!(g.dbg & 8)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        URN_STAMP(c5)     // barrier passed
#undef URN_STAMP
        if (STAMP) { st_frag += c1 - c0; st_fetch += c2 - c1; st_mfma += c3 - c2; st_park += c4 - c3; st_bar += c5 - c4; st_n += 1; }
    };

    // 2. offset loop, unrolled by three so that every register set has a static name
    int t0 = next_t(), t1 = next_t(), t2 = next_t();
    fetch(ra0, rb0, t0); fetch(ra1, rb1, t1); fetch(ra2, rb2, t2);
    park(ra0, rb0, t0, 0);
    __syncthreads();
    int buf = 0;
    while (t0 >= 0) {
        int t3 = next_t();
        step(t0, buf, ra1, rb1, t1, ra0, rb0, t3); buf ^= 1;      // fetch t3 -> set 0
        if (t1 < 0) break;
        int t4 = next_t();
        step(t1, buf, ra2, rb2, t2, ra1, rb1, t4); buf ^= 1;      // fetch t4 -> set 1
        if (t2 < 0) break;
        int t5 = next_t();
        step(t2, buf, ra0, rb0, t3, ra2, rb2, t5); buf ^= 1;      // fetch t5 -> set 2
        t0 = t3; t1 = t4; t2 = t5;
    }

    if (STAMP) {   // diagnostic build: per-workgroup cycle sums of wave 0 into the (otherwise unused) partial slab
        if (tid == 0 && g.part) {
            double *o = g.part + (long)blockIdx.x * 8;
            o[0] = (double)st_frag; o[1] = (double)st_fetch; o[2] = (double)st_mfma; o[3] = (double)st_park;
            o[4] = (double)st_bar; o[5] = (double)st_n;
        }
        return;
    }
    // 3. epilogue (per 16-row block).  C layout of 16x16x4: col = lane&15, row = (lane>>4)*4 + reg
    double s0[NB], s1[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { s0[nb] = 0.0; s1[nb] = 0.0; }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = col_base + nb * 16 + r;
        float esc = 0.f, esh = 0.f, emu = 0.f, eis = 0.f;
        if (g.epi == 2) { esc = g.e_scale[col]; esh = g.e_shift[col]; emu = g.e_mean[col]; eis = g.e_invstd[col]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = row_base + q * 4 + i;
            if (row >= n_out) continue;
            const long off = row * cout + col;
            float v = acc[nb][i];
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
    if (g.epi == 0) return;   // kernel-uniform

    // 4. column partials: wave -> workgroup (LDS, wave order) -> slab row blockIdx.x
    __shared__ double s_p[2][4][NB * 16];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        double a0 = s0[nb], a1 = s1[nb];
        a0 += __shfl_xor(a0, 16); a1 += __shfl_xor(a1, 16);
        a0 += __shfl_xor(a0, 32); a1 += __shfl_xor(a1, 32);
        if (q == 0) { s_p[0][wave][nb * 16 + r] = a0; s_p[1][wave][nb * 16 + r] = a1; }
    }
    __syncthreads();
    if (tid < NB * 16) {
        const double v0 = ((s_p[0][0][tid] + s_p[0][1][tid]) + s_p[0][2][tid]) + s_p[0][3][tid];
        const double v1 = ((s_p[1][0][tid] + s_p[1][1][tid]) + s_p[1][2][tid]) + s_p[1][3][tid];
        // write-through (sc1) 8-byte stores: visible to the reducing workgroup without an agent-scope release
        // fence, i.e. without a write-back of this XCD's whole L2 at the end of every workgroup
        __hip_atomic_store(&g.part[((long)blockIdx.x * 2 + 0) * cout + col_base + tid], v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&g.part[((long)blockIdx.x * 2 + 1) * cout + col_base + tid], v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!g.sync_word) return;   // kernel-uniform: the caller finalizes with a separate launch

    // 5. the last workgroup to arrive reduces the slab (sc1 slab stores -> every wave's vmcnt(0) -> barrier ->
    //    relaxed agent ticket; reducer: agent acquire -> plain loads; cdna_hip_programming.md, in-launch
    //    split-K reduction recipe) and finalizes the BatchNorm quantities
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(g.sync_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (ticket == gridDim.x * gridDim.y - 1u);
    }
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    {
        __shared__ double s_r[2][256];
        const int n_part = gridDim.x;
        const int cl = tid & 15, gi = tid >> 4;
        for (int cb = 0; cb < cout; cb += 16) {
            const int col = cb + cl;
            double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
            int p = gi;
            for (; p + 16 < n_part; p += 32) {
                a0 += g.part[((long)p * 2) * cout + col];
                a1 += g.part[((long)p * 2 + 1) * cout + col];
                b0 += g.part[((long)(p + 16) * 2) * cout + col];
                b1 += g.part[((long)(p + 16) * 2 + 1) * cout + col];
            }
            if (p < n_part) { a0 += g.part[((long)p * 2) * cout + col]; a1 += g.part[((long)p * 2 + 1) * cout + col]; }
            s_r[0][tid] = a0 + b0; s_r[1][tid] = a1 + b1;
            __syncthreads();
            if (tid < 16) {
                double v0 = 0.0, v1 = 0.0;
                for (int j = 0; j < 16; ++j) { v0 += s_r[0][j * 16 + tid]; v1 += s_r[1][j * 16 + tid]; }
                const long n = g.fin_n;
                if (g.epi == 1) {
                    const double mu = n > 0 ? v0 / (double)n : 0.0;
                    double var = n > 0 ? v1 / (double)n - mu * mu : 0.0;
                    if (var < 0.0) var = 0.0;
                    const double is = 1.0 / sqrt(var + g.fin_eps);
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const GArgs::FinBN &b = g.fin_bn[f];
                        if (!b.mean) continue;
                        b.mean[col] = (float)mu;
                        b.invstd[col] = (float)is;
                        const float sc = b.gamma[col] * (float)is;
                        b.scale[col] = sc;
                        b.shift[col] = fmaf(-(float)mu, sc, b.beta[col]);
                        if (b.running_mean) b.running_mean[col] = (float)(g.fin_momentum * b.running_mean[col] + (1.0 - g.fin_momentum) * mu);
                        if (b.running_var) b.running_var[col] = (float)(g.fin_momentum * b.running_var[col] + (1.0 - g.fin_momentum) * var);
                    }
                } else {
                    const double invn = n > 0 ? 1.0 / (double)n : 0.0;
                    g.fin_dbeta[col] += (float)v0;
                    g.fin_dgamma[col] += (float)v1;
                    g.fin_coef0[col] = (float)(v0 * invn);
                    g.fin_coef1[col] = (float)(v1 * invn);
                }
            }
            __syncthreads();
        }
        if (tid == 0) *g.sync_word = 0u;   // ready for the next launch on this stream
    }
}

template <int KS, int NB>
static void launch_lds(const GArgs &a, long n_out, hipStream_t st)
{
    const long blocks = (n_out + 63) / 64;
    hipLaunchKernelGGL((k_gconv_lds<KS, NB>), dim3((unsigned)blocks, a.cout / (NB * 16)), dim3(256), 0, st, a);
}

long g_lds_min_wgs = 1024;   // tuning knob (urn_set_option "gconv_lds_min_wgs")

// LDS per workgroup: idx 7 KB + A 4*16*LDA*4 + B 2*NB*16*LDA*4  (LDA = cin+4); keep it <= 64 KB
template <int KS>
static bool launch_lds_ks(const GArgs &a, long n_out, int nblk, hipStream_t st)
{
    const long blocks = (n_out + 63) / 64;
    int nb = 1;
    // widen the column tile (fewer re-gathers of the A rows) while the launch still has >= 1024 workgroups
    for (int d = 4; d >= 2; --d) {
        if (nblk % d) continue;
        const long lds = 7168 + 4L * 16 * (KS * 16 + 4) * 4 + 2L * d * 16 * (KS * 16 + 4) * 4;
        if (lds > 65536) continue;
        if (blocks * (nblk / d) >= g_lds_min_wgs) { nb = d; break; }
    }
    if ((a.dbg & 32) && a.part) {   // diagnostic stamp build (KS = 4 only)
        if constexpr (KS == 4) {
            hipLaunchKernelGGL((k_gconv_lds<4, 1, 1>), dim3((unsigned)blocks, a.cout / 16), dim3(256), 0, st, a);
            return true;
        }
    }
    if (nb == 4) { if constexpr (KS <= 6) { launch_lds<KS, 4>(a, n_out, st); return true; } nb = 2; }
    if (nb == 3) { if constexpr (KS <= 8) { launch_lds<KS, 3>(a, n_out, st); return true; } nb = 1; }
    if (nb == 2 && nblk % 2 == 0) { if constexpr (KS <= 10) { launch_lds<KS, 2>(a, n_out, st); return true; } }
    launch_lds<KS, 1>(a, n_out, st);
    return true;
}

bool urn_gconv_lds_launch(const GArgs &a, int ks, long n_out, hipStream_t st)
{
    const int nblk = a.cout / 16;
    switch (ks) {
    case 1: return launch_lds_ks<1>(a, n_out, nblk, st);
    case 2: return launch_lds_ks<2>(a, n_out, nblk, st);
    case 3: return launch_lds_ks<3>(a, n_out, nblk, st);
    case 4: return launch_lds_ks<4>(a, n_out, nblk, st);
    case 5: return launch_lds_ks<5>(a, n_out, nblk, st);
    case 6: return launch_lds_ks<6>(a, n_out, nblk, st);
    case 8: return launch_lds_ks<8>(a, n_out, nblk, st);
    case 10: return launch_lds_ks<10>(a, n_out, nblk, st);
    case 12: return launch_lds_ks<12>(a, n_out, nblk, st);
    case 14: return launch_lds_ks<14>(a, n_out, nblk, st);
    default: return false;
    }
}
