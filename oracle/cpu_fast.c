/* CPU baseline kernels for bench.py's `cpu_baseline` leg (oracle/cpu_port.py): the gather convolution of
 * SparseConvNet's CPU path -- per filter offset, gather the input rows of the offset's rules, multiply with W[offset],
 * add into the output rows (reference call sites uresnet/models/uresnet_sparse.py:21-22; the library itself is absent) --
 * in fp32 on ALL host cores with OpenMP.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY: nothing under uresnet_pytorch_amd/ links or loads this file.
 *
 * Why it exists beside sparse_ref.c (the fp64-accumulating checker) and the torch-op port: the torch-op port issues ~30
 * index_select / mm / index_add_ calls per convolution, which do not scale past 8-16 threads on a 50k-voxel event (bench
 * round 2: 0.78 s per step on 8 of 256 host CPUs).  Here a convolution is one parallel loop over tiles of 64 output rows
 * (output stationary over the dense [K][ld] neighbour table: no scatter conflicts, no atomics); inside a tile the loop
 * runs offset by offset so that W[offset] stays in L1/L2 for the tile's rules (the sgemm of the gather-GEMM-scatter form,
 * with M = the rules of the offset in the tile).  The weight gradient keeps one private dW per thread and reduces them.
 * Results are checked against the oracle in tests/test_oracle_sparse.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 64

void cpuf_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int cpuf_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* y[j,:] (+)= sum_o x[tbl[o*ld + j],:] @ W[o]   W: (K, cin, cout) row-major; tbl entry -1 = no rule.
 * flip: table row o carries the weights of offset K-1-o (input gradient of a submanifold convolution on its own table).
 * wt_transposed: W is used as (K, cout_eff = cin of W, ...) i.e. y = x @ W[o]^T with W (K, cout, cin) -- the input gradient. */
__attribute__((target_clones("avx512f", "avx2", "default")))
void cpuf_gconv(const float *x, const float *W, const int32_t *tbl, int64_t ld, int K, int flip, int64_t n_out,
                int cin, int cout, int w_is_transposed, float *y)
{
    const int64_t ntiles = (n_out + TILE - 1) / TILE;
#pragma omp parallel
    {
        float *acc = (float *)malloc(sizeof(float) * TILE * (size_t)cout);
        float *wbuf = w_is_transposed ? (float *)malloc(sizeof(float) * (size_t)cin * cout) : NULL;
#pragma omp for schedule(dynamic, 4)
        for (int64_t t = 0; t < ntiles; ++t) {
            const int64_t j0 = t * TILE, j1 = j0 + TILE < n_out ? j0 + TILE : n_out;
            memset(acc, 0, sizeof(float) * TILE * (size_t)cout);
            for (int o = 0; o < K; ++o) {
                const int32_t *row = tbl + (int64_t)o * ld;
                int any = 0;
                for (int64_t j = j0; j < j1; ++j) any |= row[j] >= 0;
                if (!any) continue;
                const int ow = flip ? K - 1 - o : o;
                const float *Wo = W + (int64_t)ow * cin * cout;
                if (w_is_transposed) {
                    /* W[ow] is (cout, cin): transpose once per (tile, offset) into (cin, cout) so that the inner loop is contiguous */
                    for (int d = 0; d < cout; ++d)
                        for (int c = 0; c < cin; ++c) wbuf[(size_t)c * cout + d] = Wo[(size_t)d * cin + c];
                    Wo = wbuf;
                }
                for (int64_t j = j0; j < j1; ++j) {
                    const int32_t i = row[j];
                    if (i < 0) continue;
                    const float *xr = x + (int64_t)i * cin;
                    float *a = acc + (size_t)(j - j0) * cout;
                    for (int c = 0; c < cin; ++c) {
                        const float xv = xr[c];
                        const float *w = Wo + (size_t)c * cout;
                        for (int d = 0; d < cout; ++d) a[d] += xv * w[d];
                    }
                }
            }
            for (int64_t j = j0; j < j1; ++j) memcpy(y + j * cout, acc + (size_t)(j - j0) * cout, sizeof(float) * (size_t)cout);
        }
        free(acc);
        free(wbuf);
    }
}

/* dW[o] += sum_j x[tbl[o*ld + j],:]^T dy[j,:]      dW: (K, cin, cout), zeroed by the caller */
__attribute__((target_clones("avx512f", "avx2", "default")))
void cpuf_gconv_dw(const float *x, const float *dy, const int32_t *tbl, int64_t ld, int K, int64_t n_out, int cin,
                   int cout, float *dw)
{
    const size_t wn = (size_t)K * cin * cout;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    float *priv = (float *)calloc((size_t)nthreads * wn, sizeof(float));
    const int64_t ntiles = (n_out + TILE - 1) / TILE;
#pragma omp parallel
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        float *mine = priv + (size_t)tid * wn;
#pragma omp for schedule(dynamic, 4)
        for (int64_t t = 0; t < ntiles; ++t) {
            const int64_t j0 = t * TILE, j1 = j0 + TILE < n_out ? j0 + TILE : n_out;
            for (int o = 0; o < K; ++o) {
                const int32_t *row = tbl + (int64_t)o * ld;
                float *wo = mine + (size_t)o * cin * cout;
                for (int64_t j = j0; j < j1; ++j) {
                    const int32_t i = row[j];
                    if (i < 0) continue;
                    const float *xr = x + (int64_t)i * cin;
                    const float *g = dy + j * cout;
                    for (int c = 0; c < cin; ++c) {
                        const float xv = xr[c];
                        float *w = wo + (size_t)c * cout;
                        for (int d = 0; d < cout; ++d) w[d] += xv * g[d];
                    }
                }
            }
        }
        /* reduce the private copies: element ranges over threads */
#pragma omp for schedule(static)
        for (int64_t e = 0; e < (int64_t)wn; ++e) {
            float s = 0.f;
            for (int k = 0; k < nthreads; ++k) s += priv[(size_t)k * wn + e];
            dw[e] += s;
        }
    }
    free(priv);
}
