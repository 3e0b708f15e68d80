// Gather convolution with LDS-DMA operand rings (global_load_lds_dwordx4), gfx950.
//
// Same arithmetic and tile as urn_gconv_lds.hip (workgroup = 4 waves = 64 rows x NB*16 columns,
// v_mfma_f32_16x16x4_f32, output stationary, bit-reproducible), but neither operand passes through
// VGPRs on its way to LDS.  Register staging lost to the compiler: hipcc re-used staging registers
// of in-flight loads and inserted s_waitcnt vmcnt(0) inside the offset loop, so every step paid a full
// memory round trip (measured: ~2800 cycles per offset step for 16 MFMAs).  Here:
//   * per filter offset each wave issues KS row-gather DMAs for its own 16 rows (per-lane source
//     address = a gathered row, or a zero page for a missing neighbour; the LDS destination is
//     lane-linear) and its share of the NB*KS weight-tile DMAs;
//   * D offsets are in flight in an LDS ring; a step waits with a COUNTED s_waitcnt vmcnt for
//     exactly its own offset, then one raw s_barrier (no vmcnt(0) drain);
//   * LDS images are linear (a DMA cannot pad), bank conflicts are avoided by permuting the
//     16-byte chunks of every row on the SOURCE address and applying the same involution on the
//     fragment read (cdna_hip_programming.md, rule 21);
//   * the BatchNorm+ReLU input transform is applied to the A fragments after they are read.
#include "urn_common.h"
#include "urn_gconv_int.h"

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ float g_zero_page[256];   // source of missing-neighbour rows (zero-initialised)

typedef __attribute__((address_space(3))) void *lds_ptr_t;

// chunk permutation of row r (chunks are 16 bytes; NCH = chunks per row), an XOR involution
template <int NCH>
__device__ __forceinline__ int chunk_swz(int r)
{
    if (NCH == 4) return (r >> 2) & 3;
    if (NCH == 8) return (r >> 1) & 7;
    if (NCH == 16 || NCH == 32) return r & 15;
    return (r >> 2) & 3;   // row lengths that are not a power of two: permute inside groups of four chunks
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int KS, int NB, int D>
__global__ __launch_bounds__(256) void k_gconv_dma(GArgs g)
{
    constexpr int CIN = KS * 16;
    constexpr int NCH = 4 * KS;                 // 16-byte chunks per row
    constexpr int ROWB = CIN * 4;               // bytes per row
    constexpr int A_SLOT = 16 * ROWB;           // one wave's 16 gathered rows = KS KiB
    constexpr int B_SLOT = NB * 16 * ROWB;      // weight tile = NB*KS KiB
    constexpr int BI = (NB * KS + 3) / 4;       // weight DMAs per wave and offset
    constexpr int PER = KS + BI;                // DMAs per wave and offset
    static_assert((D - 1) * PER <= 63, "vmcnt is a 6-bit counter");
    // ONE __shared__ object: with several, hipcc orders every ds_read behind the LDS-DMAs in flight with an
    // s_waitcnt vmcnt(0) (cdna_hip_programming.md, "second __shared__ object" trap) and the ring never overlaps
    constexpr int OFF_A = 0;                                  // [4 waves][D][A_SLOT]
    constexpr int OFF_B = OFF_A + 4 * D * A_SLOT;             // [D][B_SLOT]
    constexpr int OFF_DUMMY = OFF_B + D * B_SLOT;             // [4][1024]
    constexpr int OFF_IDX = OFF_DUMMY + 4 * 1024;             // int [4][28*16]
    constexpr int OFF_MASK = OFF_IDX + 4 * 28 * 16 * 4;       // unsigned [4]
    constexpr int OFF_LIST = OFF_MASK + 16;                   // int [32]
    constexpr int OFF_XF = OFF_LIST + 128;                    // float [2][CIN]
    constexpr int SMEM = OFF_XF + 2 * CIN * 4;
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];
    int (*s_idx)[28 * 16] = (int (*)[28 * 16])(smem + OFF_IDX);
    unsigned *s_mask = (unsigned *)(smem + OFF_MASK);
    int *s_list = (int *)(smem + OFF_LIST);
    float (*s_xf)[CIN] = (float (*)[CIN])(smem + OFF_XF);
    auto a_slot = [&](int w, int slot) -> char * { return smem + OFF_A + (w * D + slot) * A_SLOT; };
    auto b_slot = [&](int slot) -> char * { return smem + OFF_B + slot * B_SLOT; };

    const long n_out = g.n_dev ? (long)*g.n_dev : g.n_cap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const long row_base = ((long)blockIdx.x * 4 + wave) * 16;
    const int col_base = blockIdx.y * (NB * 16);
    const int K = g.K, cout = g.cout;
    const bool xf = g.xf_scale != nullptr;

    // 1. table fetch, active-offset masks, list of the workgroup's active offsets
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
    const unsigned m_all = s_mask[0] | s_mask[1] | s_mask[2] | s_mask[3];
    const int n_act = __popc(m_all);
    if (tid < 32) {
        // tid-th set bit of m_all
        unsigned mm = m_all;
        int t = -1;
        for (int i = 0; i <= tid && mm; ++i) { t = __builtin_ctz(mm); mm &= mm - 1u; if (i < tid) t = -1; }
        s_list[tid] = (tid < n_act) ? t : -1;
    }
    __syncthreads();   // last use of plain global loads (table) before the DMA loop: they are drained here

    // per-lane constants of the DMA images: byte offset of this lane in instruction j of a slot
    // A image: rows of ROWB bytes, lane covers [j*1024 + lane*16, +16)
    auto issue = [&](int t, int slot) {
        const int o = g.flip ? (K - 1 - t) : t;
        // this wave's 16 gathered rows
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const int boff = j * 1024 + lane * 16;
            const int row = boff / ROWB, p = (boff - row * ROWB) >> 4;
            const int c = p ^ chunk_swz<NCH>(row);           // global chunk that lands at LDS position p
            const int idx = s_idx[wave][t * 16 + row];
            const float *src = idx >= 0 ? g.x + (long)idx * CIN + 4 * c : g_zero_page + 4 * c;
            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(a_slot(wave, slot) + j * 1024), 16, 0, 0);
        }
        // this wave's share of the weight tile W[o][col_base .. +NB*16][0..CIN)
        const float *wo = g.wt + ((long)o * cout + col_base) * CIN;
#pragma unroll
        for (int jj = 0; jj < BI; ++jj) {
            const int i = wave + 4 * jj;                      // DMA instruction index inside the tile
            if (i < NB * KS) {
                const int boff = i * 1024 + lane * 16;
                const int row = boff / ROWB, p = (boff - row * ROWB) >> 4;
                const int c = p ^ chunk_swz<NCH>(row & 15);
                __builtin_amdgcn_global_load_lds(wo + (long)row * CIN + 4 * c, (lds_ptr_t)(b_slot(slot) + i * 1024), 16, 0, 0);
            } else {   // keep the per-wave DMA count uniform so the counted wait below stays exact
                __builtin_amdgcn_global_load_lds(g_zero_page + 4 * (lane & 31), (lds_ptr_t)(smem + OFF_DUMMY + wave * 1024), 16, 0, 0);
            }
        }
    };

    f32x4 acc0[NB], acc1[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { acc0[nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // 2. prologue: D-1 offsets in flight
#pragma unroll
    for (int k = 0; k < D - 1; ++k)
        if (k < n_act) issue(s_list[k], k);

    // 3. offset loop
    const int sw_r = chunk_swz<NCH>(r);
    for (int k = 0; k < n_act; ++k) {
        const int t = s_list[k];
        const int slot = k % D;
        // my DMAs of offset k have landed when at most the younger (D-2) offsets' are outstanding
        const int younger = min(n_act - 1 - k, D - 2);
        if (younger >= D - 2 && D >= 2) wait_vmcnt<(D - 2 > 0 ? (D - 2) * PER : 0)>();
        else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // everybody's DMAs of offset k are visible; slot (k-1)%D is free
        asm volatile("" ::: "memory");
        if (k + D - 1 < n_act) issue(s_list[k + D - 1], (k + D - 1) % D);
        if ((amask >> t) & 1u) {  // wave-uniform
            const bool have = s_idx[wave][t * 16 + r] >= 0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int pa = (ks * 4 + q) ^ sw_r;
                f32x4 a = *(const f32x4 *)(a_slot(wave, slot) + r * ROWB + pa * 16);
                if (xf) {
                    const f32x4 sc = *(const f32x4 *)&s_xf[0][ks * 16 + 4 * q];
                    const f32x4 sh = *(const f32x4 *)&s_xf[1][ks * 16 + 4 * q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = have ? fmaxf(fmaf(a[e], sc[e], sh[e]), 0.f) : 0.f;
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const f32x4 b = *(const f32x4 *)(b_slot(slot) + (nb * 16 + r) * ROWB + pa * 16);
                    if (ks & 1) {
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) acc1[nb] = MFMA16(a[tt], b[tt], acc1[nb]);
                    } else {
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) acc0[nb] = MFMA16(a[tt], b[tt], acc0[nb]);
                    }
                }
            }
        }
    }

    // 4. epilogue (per 16-row block).  C layout of 16x16x4: col = lane&15, row = (lane>>4)*4 + reg
    if (row_base >= n_out) return;
    const long tile = (long)blockIdx.x * 4 + wave;
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
            float v = acc0[nb][i] + acc1[nb][i];
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

template <int KS, int NB, int D>
static void launch_dma(const GArgs &a, long n_out, hipStream_t st)
{
    const long blocks = (n_out + 63) / 64;
    hipLaunchKernelGGL((k_gconv_dma<KS, NB, D>), dim3((unsigned)blocks, a.cout / (NB * 16)), dim3(256), 0, st, a);
}

// ring depth by LDS budget: A ring 4*D*KS KiB + B ring D*NB*KS KiB (+ ~12 KiB) must fit 160 KiB
bool urn_gconv_dma_launch(const GArgs &a, int ks, long n_out, hipStream_t st)
{
    switch (ks) {
    case 1: launch_dma<1, 1, 4>(a, n_out, st); return true;
    case 2: launch_dma<2, 1, 4>(a, n_out, st); return true;
    case 3: launch_dma<3, 1, 4>(a, n_out, st); return true;
    case 4: launch_dma<4, 1, 4>(a, n_out, st); return true;
    case 5: launch_dma<5, 1, 4>(a, n_out, st); return true;
    case 6: launch_dma<6, 1, 3>(a, n_out, st); return true;
    case 8: launch_dma<8, 1, 3>(a, n_out, st); return true;
    case 10: launch_dma<10, 1, 2>(a, n_out, st); return true;
    case 12: launch_dma<12, 1, 2>(a, n_out, st); return true;
    default: return false;
    }
}
