// Integer phase: coordinate hash, first-occurrence site numbering, submanifold
// rulebook, strided-level tables.  HBM-bound integer work: coalesced reads of the
// COO coordinate list, 64-bit CAS linear-probing hash in HBM (L2-resident at these
// sizes), wave64 ballot + popcount prefix sums for the order-preserving compaction.
//
// Replaces the host-side hash-map work behind scn.InputLayer / scn.SubmanifoldConvolution /
// scn.Convolution at reference uresnet/models/uresnet_sparse.py:20-22.
#include "urn_common.h"
#include "urn_prof.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void urn_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *urn_last_error(void) { return g_err; }
extern "C" int urn_version(void) { return 100; }

#define UB 256  // threads per block in the unique pipeline

struct HashView {
    unsigned long long *keys;
    int *first;  // smallest row index that inserted the key
    int *site;   // site number of the key
    unsigned long long mask;
};

static inline HashView hash_view(void *hash, int64_t hcap)
{
    HashView h;
    h.keys = (unsigned long long *)hash;
    h.first = (int *)((char *)hash + 8 * hcap);
    h.site = h.first + hcap;
    h.mask = (unsigned long long)hcap - 1;
    return h;
}

extern "C" int64_t urn_hash_capacity(int64_t n)
{
    int64_t cap = 1024;
    while (cap < 2 * n + 2) cap <<= 1;
    return cap;
}
extern "C" int64_t urn_hash_bytes(int64_t hcap) { return hcap * 16; }

static inline int64_t n_blocks(int64_t n) { return (n + UB - 1) / UB; }

extern "C" int64_t urn_unique_scratch_bytes(int64_t n)
{
    // rowslot[n] | blocksum[nblk + 1], 256-byte aligned pieces
    int64_t a = ((4 * n + 255) / 256) * 256;
    int64_t b = ((4 * (n_blocks(n) + 1) + 255) / 256) * 256;
    return a + b + 256;
}

extern "C" int urn_hash_clear(void *hash, int64_t bytes, void *stream)
{
    URN_CHECK_ARG(hash && bytes >= 0, "null hash");
    if (hipMemsetAsync(hash, 0x7F, (size_t)bytes, (hipStream_t)stream) != hipSuccess) {
        urn_set_error("urn_hash_clear: hipMemsetAsync failed");
        return URN_EHIP;
    }
    return URN_OK;
}

__global__ void k_fill_i32(int *p, long n, int v)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

extern "C" int urn_fill_i32(int32_t *p, int64_t n, int32_t v, void *stream)
{
    if (n <= 0) return URN_OK;
    URN_CHECK_ARG(p, "null pointer");
    int grid = urn_cdiv(n, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_fill_i32, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, (long)n, v);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

// ---- unique-by-first-occurrence pipeline --------------------------------------------
// U1: insert keys, remember the smallest inserting row per slot.
__global__ void k_insert(const int *__restrict__ coords, const int *n_dev, long n_cap, int shift,
                         HashView h, int *__restrict__ rowslot)
{
    long n = n_dev ? (long)*n_dev : n_cap;
    long i = (long)blockIdx.x * UB + threadIdx.x;
    if (i >= n) return;
    int4 c = ((const int4 *)coords)[i];
    unsigned long long key = urn_key(c.x >> shift, c.y >> shift, c.z >> shift, c.w);
    unsigned long long s = urn_mix(key) & h.mask;
    for (;;) {
        unsigned long long prev = atomicCAS(&h.keys[s], URN_EMPTY_KEY, key);
        if (prev == URN_EMPTY_KEY || prev == key) break;
        s = (s + 1) & h.mask;
    }
    atomicMin(&h.first[s], (int)i);
    rowslot[i] = (int)s;
}

// block-wide exclusive prefix of a 0/1 flag; returns this thread's prefix, total in *tot
__device__ __forceinline__ int block_prefix(bool flag, int *tot)
{
    __shared__ int s_w[UB / 64 + 1];
    unsigned long long b = __ballot(flag);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int pre = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) s_w[w] = __popcll(b);
    __syncthreads();
    int base = 0, t = 0;
#pragma unroll
    for (int k = 0; k < UB / 64; ++k) {
        int v = s_w[k];
        if (k < w) base += v;
        t += v;
    }
    *tot = t;
    __syncthreads();
    return base + pre;
}

// U2: per-block count of first-occurrence rows
__global__ void k_flag_count(const int *n_dev, long n_cap, HashView h, const int *__restrict__ rowslot,
                             int *__restrict__ blocksum)
{
    long n = n_dev ? (long)*n_dev : n_cap;
    long i = (long)blockIdx.x * UB + threadIdx.x;
    bool f = (i < n) && (h.first[rowslot[i]] == (int)i);
    int tot;
    block_prefix(f, &tot);
    if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}

// U4: number the first-occurrence rows in row order, write site coords, publish slot->site
// (U3, the exclusive scan of the per-block counts, happens here: every block adds up the counts of the blocks before
// it -- a few hundred integers -- instead of a separate one-block scan launch; the last block publishes the total)
__global__ void k_assign(const int *__restrict__ coords, const int *n_dev, long n_cap, int shift,
                         HashView h, const int *__restrict__ rowslot,
                         const int *__restrict__ blocksum, int *__restrict__ site_coords, int *__restrict__ n_unique)
{
    __shared__ int s_part[UB / 64];
    long n = n_dev ? (long)*n_dev : n_cap;
    long i = (long)blockIdx.x * UB + threadIdx.x;
    int slot = (i < n) ? rowslot[i] : 0;
    bool f = (i < n) && (h.first[slot] == (int)i);
    int mine = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += UB) mine += blocksum[b];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = mine;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < UB / 64; ++k) base += s_part[k];
    int tot;
    int pre = block_prefix(f, &tot);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_unique = base + tot;
    if (f) {
        int s = base + pre;
        int4 c = ((const int4 *)coords)[i];
        c.x >>= shift; c.y >>= shift; c.z >>= shift;
        ((int4 *)site_coords)[s] = c;
        h.site[slot] = s;
    }
}

// U5a: input rows -> site
__global__ void k_row2site(const int *n_dev, long n_cap, HashView h, const int *__restrict__ rowslot,
                           int *__restrict__ row2site)
{
    long n = n_dev ? (long)*n_dev : n_cap;
    long i = (long)blockIdx.x * UB + threadIdx.x;
    if (i < n) row2site[i] = h.site[rowslot[i]];
}

// U5b: fine site -> coarse parent and 2^3 offset
__global__ void k_parent_off(const int *__restrict__ coords, const int *n_dev, long n_cap, HashView h,
                             const int *__restrict__ rowslot, int *__restrict__ parent,
                             int *__restrict__ off, int *__restrict__ chd, long ld_c, int *__restrict__ up, long ld_f)
{
    long n = n_dev ? (long)*n_dev : n_cap;
    long i = (long)blockIdx.x * UB + threadIdx.x;
    if (i >= n) return;
    int4 c = ((const int4 *)coords)[i];
    const int p = h.site[rowslot[i]];
    const int o = ((c.x & 1) * 2 + (c.y & 1)) * 2 + (c.z & 1);
    parent[i] = p;
    off[i] = o;
    if (chd) {   // the gather tables of the strided pair in the same pass (urn_level_down_tables)
        chd[(long)o * ld_c + p] = (int)i;
        up[(long)o * ld_f + i] = p;
    }
}

static int run_unique(const int32_t *coords, const int *n_dev, int64_t n_cap, int shift, void *hash,
                      int64_t hcap, void *scratch, int64_t scratch_bytes, int32_t *site_coords,
                      int32_t *n_unique, int **rowslot_out, hipStream_t st)
{
    if (scratch_bytes < urn_unique_scratch_bytes(n_cap)) {
        urn_set_error("unique: scratch too small");
        return URN_EINVAL;
    }
    if ((hcap & (hcap - 1)) != 0 || hcap < 2 * n_cap) {
        urn_set_error("unique: hash capacity must be a power of two >= 2n");
        return URN_EINVAL;
    }
    HashView h = hash_view(hash, hcap);
    int *rowslot = (int *)scratch;
    int *blocksum = (int *)((char *)scratch + ((4 * n_cap + 255) / 256) * 256);
    int nblk = (int)n_blocks(n_cap);
    *rowslot_out = rowslot;
    if (nblk == 0) {
        (void)hipMemsetAsync(n_unique, 0, 4, st);
        return URN_OK;
    }
    hipLaunchKernelGGL(k_insert, dim3(nblk), dim3(UB), 0, st, coords, n_dev, (long)n_cap, shift, h, rowslot);
    hipLaunchKernelGGL(k_flag_count, dim3(nblk), dim3(UB), 0, st, n_dev, (long)n_cap, h, rowslot, blocksum);
    hipLaunchKernelGGL(k_assign, dim3(nblk), dim3(UB), 0, st, coords, n_dev, (long)n_cap, shift, h, rowslot,
                       blocksum, site_coords, n_unique);
    return URN_OK;
}

extern "C" int urn_sites_build(const int32_t *coords, int64_t n, int spatial, void *hash, int64_t hcap,
                               void *scratch, int64_t scratch_bytes, int32_t *row2site,
                               int32_t *site_coords, int32_t *n_active, void *stream)
{
    URN_CHECK_ARG(n >= 0 && hash && scratch && n_active, "null pointer");
    URN_CHECK_ARG(n == 0 || (coords && row2site && site_coords), "null pointer");
    URN_CHECK_ARG(spatial > 0 && spatial <= 32768, "spatial size must be in 1..32768");
    URN_CHECK_ARG(n < 0x7F000000ll, "too many rows");
    hipStream_t st = (hipStream_t)stream;
    int *rowslot;
    const bool prof = urn_prof_on();
    if (prof) urn_prof_begin(URN_PROF_INTEGER, st);
    int rc = run_unique(coords, nullptr, n, 0, hash, hcap, scratch, scratch_bytes, site_coords, n_active,
                        &rowslot, st);
    if (rc) return rc;
    if (n > 0) {
        HashView h = hash_view(hash, hcap);
        hipLaunchKernelGGL(k_row2site, dim3((int)n_blocks(n)), dim3(UB), 0, st, (const int *)nullptr, (long)n,
                           h, rowslot, row2site);
    }
    if (prof) urn_prof_end(st);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

extern "C" int urn_level_down(const int32_t *fine_coords, const int32_t *n_fine, int64_t n_cap, void *hash,
                              int64_t hcap, void *scratch, int64_t scratch_bytes,
                              int32_t *coarse_coords, int32_t *parent, int32_t *off, int32_t *n_coarse,
                              void *stream)
{
    return urn_level_down_tables(fine_coords, n_fine, n_cap, hash, hcap, scratch, scratch_bytes, coarse_coords, parent,
                                 off, n_coarse, nullptr, 0, nullptr, 0, stream);
}

extern "C" int urn_level_down_tables(const int32_t *fine_coords, const int32_t *n_fine, int64_t n_cap, void *hash,
                                     int64_t hcap, void *scratch, int64_t scratch_bytes, int32_t *coarse_coords,
                                     int32_t *parent, int32_t *off, int32_t *n_coarse, int32_t *chd, int64_t ld_c,
                                     int32_t *up, int64_t ld_f, void *stream)
{
    URN_CHECK_ARG(n_cap >= 0 && hash && scratch && n_coarse, "null pointer");
    URN_CHECK_ARG(n_cap == 0 || (fine_coords && coarse_coords && parent && off), "null pointer");
    URN_CHECK_ARG((chd == nullptr) == (up == nullptr), "chd and up go together");
    hipStream_t st = (hipStream_t)stream;
    int *rowslot;
    const bool prof = urn_prof_on();
    if (prof) urn_prof_begin(URN_PROF_INTEGER, st);
    int rc = run_unique(fine_coords, n_fine, n_cap, 1, hash, hcap, scratch, scratch_bytes, coarse_coords,
                        n_coarse, &rowslot, st);
    if (rc) return rc;
    if (n_cap > 0) {
        HashView h = hash_view(hash, hcap);
        hipLaunchKernelGGL(k_parent_off, dim3((int)n_blocks(n_cap)), dim3(UB), 0, st, fine_coords, n_fine,
                           (long)n_cap, h, rowslot, parent, off, chd, (long)ld_c, up, (long)ld_f);
    }
    if (prof) urn_prof_end(st);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

// ---- all levels of a geometry from the input rows ------------------------------------------------------------
// Level l's sites are the distinct (x >> l, y >> l, z >> l, batch) of the INPUT ROWS numbered by first occurrence among the
// rows.  That is the numbering the level-by-level pipeline gives (first occurrence among the level l-1 sites in index
// order): level-(l-1) sites are ordered by their smallest row, so the first level-(l-1) site below a level-l key is the one
// holding the key's smallest row, and keys ordered by "first site" are ordered by "smallest row".  The levels therefore do
// not depend on each other and every stage runs once for all of them (grid.y = level): 4 launches instead of 4 per level.
#define URN_GEO_MAX_LEVELS 8
struct GeoLevels {
    HashView h[URN_GEO_MAX_LEVELS];
    int *site_coords[URN_GEO_MAX_LEVELS];
    int *parent[URN_GEO_MAX_LEVELS], *off[URN_GEO_MAX_LEVELS], *chd[URN_GEO_MAX_LEVELS], *up[URN_GEO_MAX_LEVELS];
};

__global__ void k_insert_lv(const int *__restrict__ coords, long n, GeoLevels g, int *__restrict__ rowslot)
{
    const int l = blockIdx.y;
    const long i = (long)blockIdx.x * UB + threadIdx.x;
    if (i >= n) return;
    const HashView h = g.h[l];
    const int4 c = ((const int4 *)coords)[i];
    const unsigned long long key = urn_key(c.x >> l, c.y >> l, c.z >> l, c.w);
    unsigned long long s = urn_mix(key) & h.mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&h.keys[s], URN_EMPTY_KEY, key);
        if (prev == URN_EMPTY_KEY || prev == key) break;
        s = (s + 1) & h.mask;
    }
    atomicMin(&h.first[s], (int)i);
    rowslot[(long)l * n + i] = (int)s;
}

__global__ void k_flag_count_lv(long n, GeoLevels g, const int *__restrict__ rowslot, int *__restrict__ blocksum)
{
    const int l = blockIdx.y;
    const long i = (long)blockIdx.x * UB + threadIdx.x;
    const bool f = (i < n) && (g.h[l].first[rowslot[(long)l * n + i]] == (int)i);
    int tot;
    block_prefix(f, &tot);
    if (threadIdx.x == 0) blocksum[(long)l * gridDim.x + blockIdx.x] = tot;
}

__global__ void k_assign_lv(const int *__restrict__ coords, long n, GeoLevels g, const int *__restrict__ rowslot,
                            const int *__restrict__ blocksum, int *__restrict__ site_row, int *__restrict__ n_sites)
{
    __shared__ int s_part[UB / 64];
    const int l = blockIdx.y;
    const HashView h = g.h[l];
    const long i = (long)blockIdx.x * UB + threadIdx.x;
    const int slot = (i < n) ? rowslot[(long)l * n + i] : 0;
    const bool f = (i < n) && (h.first[slot] == (int)i);
    const int *bs = blocksum + (long)l * gridDim.x;
    int mine = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += UB) mine += bs[b];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = mine;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < UB / 64; ++k) base += s_part[k];
    int tot;
    const int pre = block_prefix(f, &tot);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) n_sites[l] = base + tot;
    if (f) {
        const int s = base + pre;
        int4 c = ((const int4 *)coords)[i];
        c.x >>= l; c.y >>= l; c.z >>= l;
        ((int4 *)g.site_coords[l])[s] = c;
        h.site[slot] = s;
        site_row[(long)l * n + s] = (int)i;
    }
}

// input row -> level-0 site; level-l site -> parent at level l+1, 2^3 offset, and the gather tables of the strided pair
__global__ void k_links_lv(long n, int num_levels, GeoLevels g, const int *__restrict__ rowslot,
                           const int *__restrict__ site_row, const int *__restrict__ n_sites,
                           int *__restrict__ row2site, long ld)
{
    const int l = blockIdx.y;
    const long i = (long)blockIdx.x * UB + threadIdx.x;
    if (l == 0 && i < n) row2site[i] = g.h[0].site[rowslot[i]];
    if (l + 1 >= num_levels || i >= (long)n_sites[l]) return;
    const int r = site_row[(long)l * n + i];
    const int p = g.h[l + 1].site[rowslot[(long)(l + 1) * n + r]];
    const int4 c = ((const int4 *)g.site_coords[l])[i];
    const int o = ((c.x & 1) * 2 + (c.y & 1)) * 2 + (c.z & 1);
    g.parent[l][i] = p;
    g.off[l][i] = o;
    if (g.chd[l]) {
        g.chd[l][(long)o * ld + p] = (int)i;
        g.up[l][(long)o * ld + i] = p;
    }
}

extern "C" int64_t urn_levels_scratch_bytes(int64_t n, int num_levels)
{
    // rowslot[L][n] | site_row[L][n] | blocksum[L][nblk]
    return ((int64_t)num_levels * (8 * n + 4 * n_blocks(n)) + 255) / 256 * 256 + 256;
}

extern "C" int urn_sites_build_levels(const int32_t *coords, int64_t n, int spatial, int num_levels, void *const *hash,
                                      int64_t hcap, void *scratch, int64_t scratch_bytes, int32_t *row2site,
                                      int32_t *const *site_coords, int32_t *n_sites, int32_t *const *parent,
                                      int32_t *const *off, int32_t *const *chd, int32_t *const *up, int64_t ld,
                                      void *stream)
{
    URN_CHECK_ARG(n >= 0 && num_levels >= 1 && num_levels <= URN_GEO_MAX_LEVELS && hash && scratch && n_sites && site_coords,
                  "bad argument");
    URN_CHECK_ARG(n == 0 || (coords && row2site), "null pointer");
    URN_CHECK_ARG(num_levels == 1 || (parent && off), "null pointer");
    URN_CHECK_ARG(spatial > 0 && spatial <= 32768, "spatial size must be in 1..32768");
    URN_CHECK_ARG(n < 0x7F000000ll / num_levels, "too many rows");
    URN_CHECK_ARG(scratch_bytes >= urn_levels_scratch_bytes(n, num_levels), "scratch too small");
    URN_CHECK_ARG((hcap & (hcap - 1)) == 0 && hcap >= 2 * n, "hash capacity must be a power of two >= 2n");
    URN_CHECK_ARG(ld >= n, "ld < n");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (hipMemsetAsync(n_sites, 0, 4 * (size_t)num_levels, st) != hipSuccess) { urn_set_error("urn_sites_build_levels: memset failed"); return URN_EHIP; }
        return URN_OK;
    }
    GeoLevels g;
    for (int l = 0; l < num_levels; ++l) {
        URN_CHECK_ARG(hash[l] && site_coords[l], "null pointer");
        g.h[l] = hash_view(hash[l], hcap);
        g.site_coords[l] = site_coords[l];
        const bool link = l + 1 < num_levels;
        URN_CHECK_ARG(!link || (parent[l] && off[l]), "null pointer");
        URN_CHECK_ARG(!link || !chd || !up || ((chd[l] == nullptr) == (up[l] == nullptr)), "chd and up go together");
        g.parent[l] = link ? parent[l] : nullptr; g.off[l] = link ? off[l] : nullptr;
        g.chd[l] = link && chd && up ? chd[l] : nullptr; g.up[l] = link && chd && up ? up[l] : nullptr;
    }
    int *rowslot = (int *)scratch, *site_row = rowslot + (long)num_levels * n, *blocksum = site_row + (long)num_levels * n;
    const int nblk = (int)n_blocks(n);
    const dim3 grid(nblk, num_levels);
    const bool prof = urn_prof_on();
    if (prof) urn_prof_begin(URN_PROF_INTEGER, st);
    hipLaunchKernelGGL(k_insert_lv, grid, dim3(UB), 0, st, coords, (long)n, g, rowslot);
    hipLaunchKernelGGL(k_flag_count_lv, grid, dim3(UB), 0, st, (long)n, g, rowslot, blocksum);
    hipLaunchKernelGGL(k_assign_lv, grid, dim3(UB), 0, st, coords, (long)n, g, rowslot, blocksum, site_row, n_sites);
    hipLaunchKernelGGL(k_links_lv, grid, dim3(UB), 0, st, (long)n, num_levels, g, rowslot, site_row, n_sites, row2site, (long)ld);
    if (prof) urn_prof_end(st);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

__global__ void k_down_tables(const int *__restrict__ parent, const int *__restrict__ off, const int *n_dev,
                              long n_cap, int *__restrict__ chd, long ld_c, int *__restrict__ up, long ld_f)
{
    long n = n_dev ? (long)*n_dev : n_cap;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int p = parent[i], o = off[i];
    chd[(long)o * ld_c + p] = (int)i;
    up[(long)o * ld_f + i] = p;
}

extern "C" int urn_down_tables(const int32_t *parent, const int32_t *off, const int32_t *n_fine, int64_t n_cap,
                               int32_t *chd, int64_t ld_c, int32_t *up, int64_t ld_f, void *stream)
{
    if (n_cap <= 0) return URN_OK;
    URN_CHECK_ARG(parent && off && chd && up, "null pointer");
    hipLaunchKernelGGL(k_down_tables, dim3(urn_cdiv(n_cap, 256)), dim3(256), 0, (hipStream_t)stream, parent, off,
                       n_fine, (long)n_cap, chd, (long)ld_c, up, (long)ld_f);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

// ---- submanifold rulebook -------------------------------------------------------------
// grid (row blocks, 27): one thread per (site, offset): probe the hash for site + d(o).
__global__ void k_rulebook_subm(const int *__restrict__ coords, const int *n_dev, long n_cap, int spatial,
                                HashView h, int *__restrict__ nbr, long ld, int *n_rules)
{
    long n = n_dev ? (long)*n_dev : n_cap;
    long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int o = blockIdx.y;
    int dx = o / 9 - 1, dy = (o / 3) % 3 - 1, dz = o % 3 - 1;
    int v = -1;
    if (j < n) {
        int4 c = ((const int4 *)coords)[j];
        int x = c.x + dx, y = c.y + dy, z = c.z + dz;
        if (x >= 0 && y >= 0 && z >= 0 && x < spatial && y < spatial && z < spatial) {
            unsigned long long key = urn_key(x, y, z, c.w);
            unsigned long long s = urn_mix(key) & h.mask;
            for (;;) {
                unsigned long long k = h.keys[s];
                if (k == key) { v = h.site[s]; break; }
                if (k == URN_EMPTY_KEY) break;
                s = (s + 1) & h.mask;
            }
        }
        nbr[(long)o * ld + j] = v;
    }
    if (n_rules) {
        unsigned long long b = __ballot(v >= 0);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_rules, __popcll(b));
    }
}

// all levels of one geometry in a single launch: grid (row blocks, 27, levels)
#define URN_RB_MAX_LEVELS 8
struct RBLevels {
    const int *coords[URN_RB_MAX_LEVELS];
    const int *n_dev[URN_RB_MAX_LEVELS];
    int *nbr[URN_RB_MAX_LEVELS];
    HashView h[URN_RB_MAX_LEVELS];
    int spatial[URN_RB_MAX_LEVELS];
};

__global__ void k_rulebook_subm_multi(RBLevels lv, long n_cap, long ld)
{
    const int l = blockIdx.z;
    const long n = (long)*lv.n_dev[l];
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int o = blockIdx.y;
    const int dx = o / 9 - 1, dy = (o / 3) % 3 - 1, dz = o % 3 - 1;
    const HashView h = lv.h[l];
    const int spatial = lv.spatial[l];
    int v = -1;
    const int4 c = ((const int4 *)lv.coords[l])[j];
    const int x = c.x + dx, y = c.y + dy, z = c.z + dz;
    if (x >= 0 && y >= 0 && z >= 0 && x < spatial && y < spatial && z < spatial) {
        const unsigned long long key = urn_key(x, y, z, c.w);
        unsigned long long s = urn_mix(key) & h.mask;
        for (;;) {
            const unsigned long long k = h.keys[s];
            if (k == key) { v = h.site[s]; break; }
            if (k == URN_EMPTY_KEY) break;
            s = (s + 1) & h.mask;
        }
    }
    lv.nbr[l][(long)o * ld + j] = v;
}

extern "C" int urn_rulebook_subm_multi(int num_levels, const int32_t *const *site_coords, const int32_t *const *n_dev,
                                       int64_t n_cap, const int *spatial, const void *const *hash, int64_t hcap,
                                       int32_t *const *nbr, int64_t ld, void *stream)
{
    if (n_cap <= 0 || num_levels <= 0) return URN_OK;
    URN_CHECK_ARG(num_levels <= URN_RB_MAX_LEVELS && site_coords && n_dev && spatial && hash && nbr, "bad argument");
    URN_CHECK_ARG(ld >= n_cap, "ld < n_cap");
    RBLevels lv;
    for (int l = 0; l < num_levels; ++l) {
        URN_CHECK_ARG(site_coords[l] && n_dev[l] && hash[l] && nbr[l], "null pointer");
        lv.coords[l] = site_coords[l]; lv.n_dev[l] = n_dev[l]; lv.nbr[l] = nbr[l];
        lv.h[l] = hash_view((void *)hash[l], hcap); lv.spatial[l] = spatial[l];
    }
    const bool prof = urn_prof_on();
    if (prof) urn_prof_begin(URN_PROF_INTEGER, (hipStream_t)stream);
    hipLaunchKernelGGL(k_rulebook_subm_multi, dim3(urn_cdiv(n_cap, 256), 27, num_levels), dim3(256), 0, (hipStream_t)stream,
                       lv, (long)n_cap, (long)ld);
    if (prof) urn_prof_end((hipStream_t)stream);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

extern "C" int urn_rulebook_subm(const int32_t *site_coords, const int32_t *n_dev, int64_t n_cap, int spatial,
                                 const void *hash, int64_t hcap, int32_t *nbr, int64_t ld, int32_t *n_rules,
                                 void *stream)
{
    if (n_cap <= 0) return URN_OK;
    URN_CHECK_ARG(site_coords && hash && nbr, "null pointer");
    URN_CHECK_ARG(ld >= n_cap, "ld < n_cap");
    HashView h = hash_view((void *)hash, hcap);
    hipLaunchKernelGGL(k_rulebook_subm, dim3(urn_cdiv(n_cap, 256), 27), dim3(256), 0, (hipStream_t)stream,
                       site_coords, n_dev, (long)n_cap, spatial, h, nbr, (long)ld, n_rules);
    URN_LAUNCH_CHECK();
    return URN_OK;
}

// ---- InputLayer feature merge (mode 3: sum duplicates) ---------------------------------
__global__ void k_feat_accum(const float *__restrict__ feats, const int *__restrict__ row2site, long n, int nf,
                             double *__restrict__ acc)
{
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * nf) return;
    long i = t / nf;
    int f = (int)(t - i * nf);
    atomicAdd(&acc[(long)row2site[i] * nf + f], (double)feats[t]);
}

__global__ void k_feat_round(const double *__restrict__ acc, const int *n_dev, long n_cap, int nf,
                             float *__restrict__ out)
{
    long n = n_dev ? (long)*n_dev : n_cap;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n * nf) out[t] = (float)acc[t];
}

extern "C" int urn_input_features(const float *feats, const int32_t *row2site, int64_t n, int nf,
                                  const int32_t *n_active, int64_t n_cap, double *acc64, float *site_feats,
                                  void *stream)
{
    if (n <= 0) return URN_OK;
    URN_CHECK_ARG(feats && row2site && acc64 && site_feats && nf > 0, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(acc64, 0, (size_t)(n_cap * nf) * 8, st) != hipSuccess) {
        urn_set_error("urn_input_features: memset failed");
        return URN_EHIP;
    }
    hipLaunchKernelGGL(k_feat_accum, dim3(urn_cdiv(n * nf, 256)), dim3(256), 0, st, feats, row2site, (long)n, nf,
                       acc64);
    hipLaunchKernelGGL(k_feat_round, dim3(urn_cdiv(n_cap * nf, 256)), dim3(256), 0, st, acc64, n_active,
                       (long)n_cap, nf, site_feats);
    URN_LAUNCH_CHECK();
    return URN_OK;
}
