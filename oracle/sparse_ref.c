/*
 * oracle/sparse_ref.c -- CPU restatement of the sparse U-ResNet arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (uresnet_pytorch_amd/)
 * may import, link or call this file; it is the checker used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * PARITY STATUS: "parity unpinned" by the reference.  The reference delegates
 * all sparse arithmetic to the third-party package `sparseconvnet` (import at
 * reference uresnet/models/uresnet_sparse.py:9; no pinned version: the tree has
 * no requirements.txt / setup.py / lockfile), which is not vendored and not
 * installable here, and the reference ships no tests or golden vectors.  This
 * file therefore restates the library's *published* algorithm (Graham et al.,
 * "3D Semantic Segmentation with Submanifold Sparse Convolutional Networks")
 * at the reference's call sites, and is pinned instead by a dense-equivalence
 * check against torch.nn.functional.conv3d / conv_transpose3d / batch_norm and
 * by a brute-force numpy rulebook enumerator (tests/test_oracle_*.py).
 *
 * Call sites restated (reference uresnet/models/uresnet_sparse.py):
 *   :20  scn.InputLayer(dimension, SPATIAL_SIZE, mode=3)   -> orc_sites_build
 *   :21  scn.SubmanifoldConvolution(d, 1, m, 3, False)     -> orc_rulebook_subm + orc_conv_*
 *   :22  scn.UNet(..., residual_blocks=True, downsample=[2,2])
 *          Convolution(k2,s2)/Deconvolution(k2,s2)         -> orc_level_down + orc_conv_*
 *          BatchNormLeakyReLU(leak 0)                      -> orc_bn_relu_*
 *   :23  scn.BatchNormReLU(m)                              -> orc_bn_relu_*
 *   :24  scn.OutputLayer(d)                                -> row gather by row2site (python side)
 *
 * Conventions:
 *   coords row = (x, y, z, batch) int32.
 *   subm offset index o = ((dx+1)*3 + (dy+1))*3 + (dz+1); neighbour = out + (dx,dy,dz).
 *   strided k2/s2: coarse = fine >> 1; offset o = ((x&1)*2 + (y&1))*2 + (z&1).
 *   site order: first occurrence (level 0: scanning input rows; level l+1:
 *   scanning level-l sites in index order).
 *   neighbour tables are [K][N] int32, -1 = no active neighbour.
 *   weights are (K, Cin, Cout) row-major, no bias.
 * All reductions accumulate in double and round once to float.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- hash -- */
typedef struct {
    uint64_t *keys;
    int32_t *vals;
    uint64_t mask;
} orc_hash;

#define ORC_EMPTY 0xFFFFFFFFFFFFFFFFull

static uint64_t orc_key(int32_t x, int32_t y, int32_t z, int32_t b)
{
    return ((uint64_t)(uint16_t)b << 48) | ((uint64_t)(uint16_t)x << 32) |
           ((uint64_t)(uint16_t)y << 16) | (uint64_t)(uint16_t)z;
}

static uint64_t orc_mix(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}

static int orc_hash_init(orc_hash *h, int64_t n)
{
    uint64_t cap = 16;
    while (cap < (uint64_t)(2 * n + 2)) cap <<= 1;
    h->keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
    h->vals = (int32_t *)malloc(cap * sizeof(int32_t));
    if (!h->keys || !h->vals) return -1;
    memset(h->keys, 0xFF, cap * sizeof(uint64_t));
    h->mask = cap - 1;
    return 0;
}

static void orc_hash_free(orc_hash *h) { free(h->keys); free(h->vals); }

/* returns existing value or inserts `val` and returns -1 */
static int32_t orc_hash_get_or_put(orc_hash *h, uint64_t key, int32_t val)
{
    uint64_t s = orc_mix(key) & h->mask;
    for (;;) {
        if (h->keys[s] == key) return h->vals[s];
        if (h->keys[s] == ORC_EMPTY) { h->keys[s] = key; h->vals[s] = val; return -1; }
        s = (s + 1) & h->mask;
    }
}

static int32_t orc_hash_find(const orc_hash *h, uint64_t key)
{
    uint64_t s = orc_mix(key) & h->mask;
    for (;;) {
        if (h->keys[s] == key) return h->vals[s];
        if (h->keys[s] == ORC_EMPTY) return -1;
        s = (s + 1) & h->mask;
    }
}

/* ------------------------------------------------------ integer phase -- */

/* InputLayer: active sites in first-occurrence order, duplicates summed
 * (mode 3), or first kept (mode 2), last kept (mode 1), mean (mode 4).
 * coords (N,4); feats (N,nf).  Outputs row2site (N), site_coords (<=N,4),
 * site_feats (<=N,nf).  Returns number of active sites. */
int64_t orc_sites_build(const int32_t *coords, int64_t N, const float *feats, int nf, int mode,
                        int32_t *row2site, int32_t *site_coords, float *site_feats)
{
    orc_hash h;
    if (orc_hash_init(&h, N)) return -1;
    int64_t na = 0;
    double *acc = (double *)calloc((size_t)(N > 0 ? N : 1) * (size_t)nf, sizeof(double));
    int32_t *cnt = (int32_t *)calloc((size_t)(N > 0 ? N : 1), sizeof(int32_t));
    for (int64_t i = 0; i < N; ++i) {
        const int32_t *c = coords + 4 * i;
        int32_t s = orc_hash_get_or_put(&h, orc_key(c[0], c[1], c[2], c[3]), (int32_t)na);
        if (s < 0) {
            s = (int32_t)na++;
            memcpy(site_coords + 4 * (int64_t)s, c, 4 * sizeof(int32_t));
            for (int f = 0; f < nf; ++f) acc[(int64_t)s * nf + f] = feats[i * nf + f];
            cnt[s] = 1;
        } else {
            for (int f = 0; f < nf; ++f) {
                double *a = &acc[(int64_t)s * nf + f];
                if (mode == 3 || mode == 4) {
                    /* fp64 sum, rounded once at the end: independent of row order */
                    *a += (double)feats[i * nf + f];
                } else if (mode == 1) {
                    *a = feats[i * nf + f];
                } /* mode 2: keep first */
            }
            cnt[s]++;
        }
        row2site[i] = s;
    }
    for (int64_t s = 0; s < na; ++s)
        for (int f = 0; f < nf; ++f) {
            double v = acc[s * nf + f];
            if (mode == 4) v /= cnt[s];
            site_feats[s * nf + f] = (float)v;
        }
    free(acc); free(cnt);
    orc_hash_free(&h);
    return na;
}

/* Submanifold 3^3 neighbour table nbr[27][Na]; returns number of rules. */
int64_t orc_rulebook_subm(const int32_t *site_coords, int64_t Na, int spatial, int32_t *nbr)
{
    orc_hash h;
    if (orc_hash_init(&h, Na)) return -1;
    for (int64_t j = 0; j < Na; ++j) {
        const int32_t *c = site_coords + 4 * j;
        orc_hash_get_or_put(&h, orc_key(c[0], c[1], c[2], c[3]), (int32_t)j);
    }
    int64_t R = 0;
    for (int64_t j = 0; j < Na; ++j) {
        const int32_t *c = site_coords + 4 * j;
        for (int dx = -1; dx <= 1; ++dx)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dz = -1; dz <= 1; ++dz) {
                    int o = ((dx + 1) * 3 + (dy + 1)) * 3 + (dz + 1);
                    int32_t x = c[0] + dx, y = c[1] + dy, z = c[2] + dz, v = -1;
                    if (x >= 0 && y >= 0 && z >= 0 && x < spatial && y < spatial && z < spatial)
                        v = orc_hash_find(&h, orc_key(x, y, z, c[3]));
                    nbr[(int64_t)o * Na + j] = v;
                    R += (v >= 0);
                }
    }
    orc_hash_free(&h);
    return R;
}

/* Strided (k2,s2) level: coarse sites in first-touch order.
 * parent[Nf], off[Nf] (0..7), coarse_coords (<=Nf,4).  Returns Nc.
 * The children table chd[8][Nc] is filled by orc_children (needs Nc). */
int64_t orc_level_down(const int32_t *fine_coords, int64_t Nf, int32_t *coarse_coords,
                       int32_t *parent, int32_t *off)
{
    orc_hash h;
    if (orc_hash_init(&h, Nf)) return -1;
    int64_t nc = 0;
    for (int64_t i = 0; i < Nf; ++i) {
        const int32_t *c = fine_coords + 4 * i;
        int32_t X = c[0] >> 1, Y = c[1] >> 1, Z = c[2] >> 1;
        int32_t s = orc_hash_get_or_put(&h, orc_key(X, Y, Z, c[3]), (int32_t)nc);
        if (s < 0) {
            s = (int32_t)nc++;
            int32_t *cc = coarse_coords + 4 * (int64_t)s;
            cc[0] = X; cc[1] = Y; cc[2] = Z; cc[3] = c[3];
        }
        parent[i] = s;
        off[i] = ((c[0] & 1) * 2 + (c[1] & 1)) * 2 + (c[2] & 1);
    }
    orc_hash_free(&h);
    return nc;
}

void orc_children(const int32_t *parent, const int32_t *off, int64_t Nf, int64_t Nc, int32_t *chd)
{
    for (int64_t k = 0; k < 8 * Nc; ++k) chd[k] = -1;
    for (int64_t i = 0; i < Nf; ++i) chd[(int64_t)off[i] * Nc + parent[i]] = (int32_t)i;
}

/* Deconvolution as a gather table over fine rows: up[8][Nf], up[o][i] = parent[i] iff off[i]==o */
void orc_up_table(const int32_t *parent, const int32_t *off, int64_t Nf, int32_t *up)
{
    for (int64_t k = 0; k < 8 * Nf; ++k) up[k] = -1;
    for (int64_t i = 0; i < Nf; ++i) up[(int64_t)off[i] * Nf + i] = parent[i];
}

/* ------------------------------------------------------- float phase -- */

/* y[j,:] = sum_o x[nbr[o][j],:] @ W[o]      (rows with nbr<0 skipped)
 * x (Nin,Cin), W (K,Cin,Cout), nbr [K][Nout], y (Nout,Cout). */
void orc_conv_fwd(const float *x, const float *W, const int32_t *nbr, int K, int64_t Nout,
                  int Cin, int Cout, float *y)
{
#pragma omp parallel
    {
        double *acc = (double *)malloc(sizeof(double) * (size_t)Cout);
#pragma omp for schedule(static)
        for (int64_t j = 0; j < Nout; ++j) {
            for (int c = 0; c < Cout; ++c) acc[c] = 0.0;
            for (int o = 0; o < K; ++o) {
                int32_t i = nbr[(int64_t)o * Nout + j];
                if (i < 0) continue;
                const float *xi = x + (int64_t)i * Cin;
                const float *w = W + (int64_t)o * Cin * Cout;
                for (int a = 0; a < Cin; ++a) {
                    double xa = xi[a];
                    const float *wr = w + (int64_t)a * Cout;
                    for (int c = 0; c < Cout; ++c) acc[c] += xa * (double)wr[c];
                }
            }
            for (int c = 0; c < Cout; ++c) y[j * Cout + c] = (float)acc[c];
        }
        free(acc);
    }
}

/* dx[i,:] = sum over rules (o, i->j) of dy[j,:] @ W[o]^T.  `inv` is the
 * inverse gather table [K][Nin]: inv[o][i] = j with nbr[o][j] == i, or -1
 * (each (o,i) has at most one j in every table this path builds). */
void orc_conv_bwd_dx(const float *dy, const float *W, const int32_t *inv, int K, int64_t Nin,
                     int Cin, int Cout, float *dx)
{
#pragma omp parallel
    {
        double *acc = (double *)malloc(sizeof(double) * (size_t)Cin);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < Nin; ++i) {
            for (int a = 0; a < Cin; ++a) acc[a] = 0.0;
            for (int o = 0; o < K; ++o) {
                int32_t j = inv[(int64_t)o * Nin + i];
                if (j < 0) continue;
                const float *g = dy + (int64_t)j * Cout;
                const float *w = W + (int64_t)o * Cin * Cout;
                for (int a = 0; a < Cin; ++a) {
                    const float *wr = w + (int64_t)a * Cout;
                    double s = 0.0;
                    for (int c = 0; c < Cout; ++c) s += (double)g[c] * (double)wr[c];
                    acc[a] += s;
                }
            }
            for (int a = 0; a < Cin; ++a) dx[i * Cin + a] = (float)acc[a];
        }
        free(acc);
    }
}

/* Inverse of a gather table: inv[o][i] = j where nbr[o][j] == i. */
void orc_invert_table(const int32_t *nbr, int K, int64_t Nout, int64_t Nin, int32_t *inv)
{
    for (int64_t k = 0; k < (int64_t)K * Nin; ++k) inv[k] = -1;
    for (int o = 0; o < K; ++o)
        for (int64_t j = 0; j < Nout; ++j) {
            int32_t i = nbr[(int64_t)o * Nout + j];
            if (i >= 0) inv[(int64_t)o * Nin + i] = (int32_t)j;
        }
}

/* dW[o] = sum_j x[nbr[o][j],:]^T (outer) dy[j,:] */
void orc_conv_bwd_dw(const float *x, const float *dy, const int32_t *nbr, int K, int64_t Nout,
                     int Cin, int Cout, float *dW)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int o = 0; o < K; ++o) {
        double *acc = (double *)calloc((size_t)Cin * (size_t)Cout, sizeof(double));
        for (int64_t j = 0; j < Nout; ++j) {
            int32_t i = nbr[(int64_t)o * Nout + j];
            if (i < 0) continue;
            const float *xi = x + (int64_t)i * Cin;
            const float *g = dy + j * Cout;
            for (int a = 0; a < Cin; ++a) {
                double xa = xi[a];
                double *ar = acc + (int64_t)a * Cout;
                for (int c = 0; c < Cout; ++c) ar[c] += xa * (double)g[c];
            }
        }
        float *d = dW + (int64_t)o * Cin * Cout;
        for (int64_t k = 0; k < (int64_t)Cin * Cout; ++k) d[k] = (float)acc[k];
        free(acc);
    }
}

/* BatchNorm (+ optional ReLU) over the (N,C) row matrix, batch statistics,
 * biased variance, eps inside the sqrt.  mean/invstd (C) are outputs. */
void orc_bn_relu_fwd(const float *x, int64_t N, int C, const float *gamma, const float *beta,
                     double eps, int relu, float *y, float *mean, float *invstd)
{
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        double s = 0.0;
        for (int64_t i = 0; i < N; ++i) s += x[i * C + c];
        double m = N > 0 ? s / (double)N : 0.0;
        double v = 0.0;
        for (int64_t i = 0; i < N; ++i) { double d = x[i * C + c] - m; v += d * d; }
        v = N > 0 ? v / (double)N : 0.0;
        double is = 1.0 / sqrt(v + eps);
        mean[c] = (float)m; invstd[c] = (float)is;
        for (int64_t i = 0; i < N; ++i) {
            double u = (x[i * C + c] - m) * is * gamma[c] + beta[c];
            if (relu && u < 0.0) u = 0.0;
            y[i * C + c] = (float)u;
        }
    }
}

/* Backward of the above.  y is the forward output (used for the ReLU mask). */
void orc_bn_relu_bwd(const float *x, const float *y, const float *dy, int64_t N, int C,
                     const float *gamma, const float *mean, const float *invstd, int relu,
                     float *dx, float *dgamma, float *dbeta)
{
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        double m = mean[c], is = invstd[c], sg = 0.0, sb = 0.0;
        for (int64_t i = 0; i < N; ++i) {
            double g = dy[i * C + c];
            if (relu && !(y[i * C + c] > 0.0f)) g = 0.0;
            sb += g;
            sg += g * (x[i * C + c] - m) * is;
        }
        dgamma[c] = (float)sg; dbeta[c] = (float)sb;
        double a = gamma[c] * is, invn = N > 0 ? 1.0 / (double)N : 0.0;
        for (int64_t i = 0; i < N; ++i) {
            double g = dy[i * C + c];
            if (relu && !(y[i * C + c] > 0.0f)) g = 0.0;
            double xh = (x[i * C + c] - m) * is;
            dx[i * C + c] = (float)(a * (g - sb * invn - xh * sg * invn));
        }
    }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
