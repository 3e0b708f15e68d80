// Whole-network executor for the sparse U-ResNet trunk (everything between scn.InputLayer and
// torch.nn.Linear at reference uresnet/models/uresnet_sparse.py:19-25, with the body of
// scn.UNet(reps, nPlanes, residual_blocks=True, downsample=[2,2]) expanded).
//
// Why it exists: one 50k-voxel event is ~350 small kernels per training step; driving them
// from Python costs more host time than the GPU needs to run them.  The executor walks the
// fixed module tree in C++, carves every activation out of one caller-provided workspace
// (bump allocator, no hipMalloc), and issues the same kernels as the per-layer C ABI
// back-to-back on the caller's stream.  Parameters and gradients are ONE flat fp32 buffer
// each (registration order of the module tree), which is also what the single RCCL
// all-reduce per step consumes.
#include "urn_common.h"
#include <algorithm>
#include <string.h>
#include <memory>
#include <vector>
#include <chrono>

#ifndef URN_SUM_SLOTS
#define URN_SUM_SLOTS 8       // rows of an accumulated-statistics slab (every conv workgroup adds its column sums into row tile % slots); cfg3 step, builds A/B on one box: 4 rows 2.574 ms, 8 rows 2.504-2.515, 16 rows 2.556, 32 rows 2.653
#endif
int g_dw_group = 1;   // weight gradients per fork to the side stream (urn_set_option "dw_group"); measured: 1: 3.61 ms, 2: 3.66, 4: 3.64, 8: 3.74, 16: 3.85

int g_net_wfrag = 1;       // fragment-ordered weight copies for the pair-list kernel (urn_set_option "net_wfrag")
int g_dw_2stage = 0;       // weight gradients of the table kernel as per-chunk partial slabs + a fixed-order reduce instead of fp32 atomics (urn_set_option "dw_2stage")
int g_net_skip_dw = 0;     // timing only (urn_set_option "net_dbg_skip_dw"): no weight-gradient launches -- what the step costs without them
int g_dw_pairs = 0;        // weight gradients on the two-stage pair-list kernel (bitwise reproducible) instead of the atomics kernel (urn_set_option "dw_pairs")
int g_net_side_probe = 4;   // candidate side streams tried by an executor's first backward (urn_set_option "net_side_probe"; 0/1 = keep the first)
int g_net_side_verbose = 0;
int g_net_side2 = 1;        // weight gradients alternate between TWO side streams (urn_set_option "net_side2"): the backward pass is bound by the kernel time queued per stream (cfg3: 1.25 ms on the caller's stream, 1.30 ms of weight gradients); two side streams 2.65 -> 2.59 ms per step

namespace {

// Side-stream probe.  HIP multiplexes streams onto a few hardware queues, and some (main stream, side stream) pairs
// run the fork -> concurrent kernels pattern of the backward pass 2.5x slower than others (measured: the 4th and 5th
// executor of a process with GPU_MAX_HW_QUEUES=4, the 4th with 8; which one is hit depends on how many streams the
// process -- torch, RCCL -- has created before).  The mapping cannot be queried, so the first backward of an
// executor times the pattern itself on a few candidate streams and keeps the fastest.
__global__ void k_probe_spin(long ticks)
{
    const long t0 = (long)wall_clock64();            // 100 MHz; bounded: every wave leaves after `ticks`
    while ((long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

double urn_now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// side2 != nullptr: the forks alternate between the two side streams, as the weight gradients of a backward pass do
double probe_pair(hipStream_t main, hipStream_t side, const std::vector<hipEvent_t> &ev, int rounds, hipStream_t side2 = nullptr)
{
    if (hipStreamSynchronize(main) != hipSuccess) return 1e30;
    const double t0 = urn_now_ms();
    for (int r = 0; r < rounds; ++r) {
        hipEvent_t e = ev[r % ev.size()];
        hipStream_t tgt = (side2 && (r & 1)) ? side2 : side;
        if (hipEventRecord(e, main) != hipSuccess || hipStreamWaitEvent(tgt, e, 0) != hipSuccess) return 1e30;
        hipLaunchKernelGGL(k_probe_spin, dim3(64), dim3(64), 0, tgt, 1500L);          // ~15 us, like a weight gradient
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_probe_spin, dim3(64), dim3(64), 0, main, 300L);   // the dX chain
    }
    hipEvent_t e = ev[rounds % ev.size()];
    if (hipEventRecord(e, side) != hipSuccess || hipStreamWaitEvent(main, e, 0) != hipSuccess) return 1e30;
    if (side2) {
        hipEvent_t e2 = ev[(rounds + 1) % ev.size()];
        if (hipEventRecord(e2, side2) != hipSuccess || hipStreamWaitEvent(main, e2, 0) != hipSuccess) return 1e30;
    }
    if (hipStreamSynchronize(main) != hipSuccess) return 1e30;
    return urn_now_ms() - t0;
}

struct Arena {
    char *base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    bool dry = false;
    bool overflow = false;
    void reset(void *p, size_t bytes, bool dry_run)
    {
        base = (char *)p; cap = bytes; off = 0; peak = 0; dry = dry_run; overflow = false;
    }
    void *alloc_bytes(size_t n)
    {
        n = (n + 255) & ~(size_t)255;
        size_t at = off;
        off += n;
        if (off > peak) peak = off;
        if (!dry && off > cap) { overflow = true; return nullptr; }
        return dry ? (void *)(uintptr_t)256 : (void *)(base + at);
    }
    float *f32(int64_t n) { return (float *)alloc_bytes((size_t)(n > 0 ? n : 1) * 4); }
};

struct Geo {
    int L = 0;
    int64_t ld = 0, n_rows = 0;
    std::vector<int64_t> n;
    std::vector<const int32_t *> nbr, chd, up;
    const int32_t *row2site = nullptr;
    // compacted rule lists of the tables (urn_pairs_build), handed over by urn_net_set_pairs; tile 0 = none
    std::vector<const int32_t *> p_nbr, p_chd, p_up;
    std::vector<int> t_nbr, t_chd, t_up;   // tile size per level (0 = no list)
    // list and tile of a table of this geometry; the centre row of a submanifold table is the identity (list NULL)
    void pairs_of(const int32_t *tbl, const int32_t *&list, int &tile) const
    {
        list = nullptr; tile = 0;
        for (int l = 0; l < L; ++l) {
            if (l < (int)p_nbr.size() && p_nbr[l] && t_nbr[l]) {
                if (tbl == nbr[l]) { list = p_nbr[l]; tile = t_nbr[l]; return; }
                if (tbl == nbr[l] + 13 * ld) { tile = t_nbr[l]; return; }
            }
            if (l + 1 < L) {
                if (l < (int)p_chd.size() && p_chd[l] && t_chd[l] && tbl == chd[l]) { list = p_chd[l]; tile = t_chd[l]; return; }
                if (l < (int)p_up.size() && p_up[l] && t_up[l] && tbl == up[l]) { list = p_up[l]; tile = t_up[l]; return; }
            }
        }
    }
};

struct ConvP { int64_t w; int K, cin, cout; const float *x = nullptr; };           // saved input
struct BNP {
    int64_t w, b, run; int c;
    const float *x = nullptr, *y = nullptr;
    float *mean = nullptr, *invstd = nullptr;
    float *scale = nullptr, *shift = nullptr;   // fused path: relu(x*scale + shift) == BatchNormReLU(x)
    int64_t nrows = 0;                         // rows of x in the last forward (urn_net_bn_export)
    unsigned long stamp = 0;                   // forward in which the arrays above were carved
};
// column partials (sum, sum of squares; fp64) of a tensor, written by its producer's epilogue
struct Stats { double *part = nullptr; int n_part = 0, ld = 0; };
// an activation tensor of the fused path: rows, channels and where its statistics come from;
// a channel concat carries two slabs (columns [0,c0) from st, [c0,c) from st2)
struct Act { float *x = nullptr; int64_t n = 0; int c = 0; Stats st, st2; int c0 = 0; };
struct Cons { BNP *bn = nullptr; int coff = 0; };
struct Block { bool has_nin; ConvP nin; BNP bn1; ConvP conv1; BNP bn2; ConvP conv2; };
struct ULevel {
    std::vector<Block> pre, post;
    bool has_sub = false;
    BNP bn_d; ConvP down; std::unique_ptr<ULevel> sub; BNP bn_u; ConvP up;
};

}  // namespace

struct urn_net {
    int m, L, reps, nc;
    double eps, momentum;
    std::vector<int> planes;
    int64_t n_params = 0, n_running = 0;
    std::vector<int64_t> p_off, p_numel;   // registration order
    ConvP stem;
    ULevel u;
    BNP bn_out;
    // per-call state
    Arena arena;
    Geo geo;
    const float *params = nullptr;
    float *running = nullptr, *grads = nullptr;
    int training = 1;
    hipStream_t st = nullptr;
    hipStream_t side = nullptr;          // weight gradients run here, off the dX -> BN critical path
    hipStream_t side2 = nullptr;         // ... and here, alternating (g_net_side2)
    bool side2_used = false;
    int dw_toggle = 0;
    std::vector<hipEvent_t> events;
    size_t ev_next = 0;
    bool side_used = false;
    void *dw2_shared = nullptr;        // partial slabs of the two-stage weight gradients (dw_scratch), one region per backward
    hipStream_t dw2_stream = nullptr;  // the stream they all run on
    bool side_probed = false;
    bool pairs_armed = false;            // urn_net_set_pairs was called for the coming forward
    // urn_net_set_head: the Linear head inside the executor (coming forward + its backward)
    bool head_armed = false, head_live = false;
    const float *head_w = nullptr, *head_b = nullptr;
    float *head_logits = nullptr;
    // urn_net_backward_cb: called once, when every kernel that writes a gradient of the decoder + bottom suffix of the flat
    // buffer has been enqueued (and the side stream has been ordered behind the caller's stream at that point)
    void (*bottom_cb)(void *) = nullptr;
    void *bottom_user = nullptr;
    // keep the fastest of g_net_side_probe candidate side streams for this executor's main stream (see probe_pair)
    // the fastest of g_net_side_probe candidate streams (the first one is `have`) beside the executor's main stream
    hipStream_t pick_one(hipStream_t have, const char *what)
    {
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        std::vector<hipStream_t> cand{have};
        for (int i = 1; i < g_net_side_probe; ++i) {
            hipStream_t c = nullptr;
            if (hipStreamCreateWithPriority(&c, hipStreamNonBlocking, prio_lo) != hipSuccess) { (void)hipGetLastError(); break; }
            cand.push_back(c);
        }
        int best = 0; double best_ms = 1e30;
        for (size_t i = 0; i < cand.size(); ++i) {
            (void)probe_pair(st, cand[i], events, 4);
            const double ms = std::min(probe_pair(st, cand[i], events, 16), probe_pair(st, cand[i], events, 16));
            if (g_net_side_verbose) fprintf(stderr, "urn_net: %s candidate %zu: %.3f ms\n", what, i, ms);
            if (ms < best_ms * 0.8) { best_ms = ms; best = (int)i; }   // later candidates must be clearly better
        }
        (void)hipGetLastError();
        for (size_t i = 0; i < cand.size(); ++i)
            if ((int)i != best) { (void)hipStreamSynchronize(cand[i]); (void)hipStreamDestroy(cand[i]); }
        return cand[best];
    }
    // keep the fastest of g_net_side_probe candidate side streams for this executor's main stream (see probe_pair)
    void pick_side()
    {
        side_probed = true;
        if (g_net_side_probe < 2 || events.size() < 64) return;
        side = pick_one(side, "side stream");
        if (side2) {
            side2 = pick_one(side2, "second side stream");
            // the three streams together: a trio that runs the pattern clearly slower than the pair gives the second stream up
            const double pair = std::min(probe_pair(st, side, events, 16), probe_pair(st, side, events, 16));
            const double trio = std::min(probe_pair(st, side, events, 16, side2), probe_pair(st, side, events, 16, side2));
            if (g_net_side_verbose) fprintf(stderr, "urn_net: one side stream %.3f ms, two %.3f ms\n", pair, trio);
            if (trio > 1.3 * pair) { (void)hipStreamSynchronize(side2); (void)hipStreamDestroy(side2); side2 = nullptr; }
        }
    }
    float *wt_all = nullptr;             // transposed copy of every conv weight, same offsets as params
    float *wf_fwd = nullptr, *wf_bwd = nullptr;   // wt_all / params once more in MFMA-fragment order (urn_gconv_args.wt_frag)
    int wf_prec = 0;                              // their element type (urn_gconv_args.wt_frag_prec): the operand precision in force when they were written
    float *ext_w = nullptr;              // urn_net_prepare_weights: the three copies live in the caller's buffer, already written
    bool w_prepared = false;
    hipEvent_t ev_w = nullptr;           // ... recorded behind them on the stream they were written on
    std::vector<ConvP *> convs;          // every conv, for the batched transpose
    int rc = URN_OK;
    float *trunk_out = nullptr;   // (n0, m) features after the last BNReLU

    int64_t add_param(int64_t numel)
    {
        int64_t o = n_params;
        p_off.push_back(o); p_numel.push_back(numel);
        n_params += numel;
        return o;
    }
    ConvP make_conv(int K, int cin, int cout) { ConvP c; c.K = K; c.cin = cin; c.cout = cout; c.w = add_param((int64_t)K * cin * cout); return c; }
    BNP make_bn(int c)
    {
        BNP b; b.c = c; b.w = add_param(c); b.b = add_param(c); b.run = n_running; n_running += 2 * (int64_t)c;
        return b;
    }
    Block make_block(int a, int b)
    {
        Block k;
        k.has_nin = a != b;
        if (k.has_nin) k.nin = make_conv(1, a, b);
        k.bn1 = make_bn(a); k.conv1 = make_conv(27, a, b); k.bn2 = make_bn(b); k.conv2 = make_conv(27, b, b);
        return k;
    }
    void make_u(ULevel &lv, int l)
    {
        const int P = planes[l];
        for (int i = 0; i < reps; ++i) lv.pre.push_back(make_block(P, P));
        if (l + 1 < L) {
            lv.has_sub = true;
            lv.bn_d = make_bn(P);
            lv.down = make_conv(8, P, planes[l + 1]);
            lv.sub.reset(new ULevel());
            make_u(*lv.sub, l + 1);
            lv.bn_u = make_bn(planes[l + 1]);
            lv.up = make_conv(8, planes[l + 1], P);
            for (int i = 0; i < reps; ++i) lv.post.push_back(make_block(i == 0 ? 2 * P : P, P));
        }
    }

    // ---- primitive steps ------------------------------------------------------------
    void check(int r) { if (r != URN_OK && rc == URN_OK) rc = r; }
    bool live() const { return !arena.dry && rc == URN_OK && !arena.overflow; }

    // a gather convolution without fusions, on the compacted rule list of its table when the geometry carries one
    // the fragment-ordered copy of a weight operand (wt_all + off -> wf_fwd + off, params + off -> wf_bwd + off)
    const float *frag_of(const float *wt, int cin, int cout) const
    {
        if (!g_net_wfrag || !wf_fwd || cin % 16 || cout % 16) return nullptr;
        if (wt >= wt_all && wt < wt_all + n_params) return wf_fwd + (wt - wt_all);
        if (wt >= params && wt < params + n_params) return wf_bwd + (wt - params);
        return nullptr;
    }
    int gconv_plain(const float *x, const float *wt, const int32_t *tbl, int K, int flip, int64_t n_out, int cin, int cout,
                    const float *res, float *y)
    {
        urn_gconv_args a;
        memset(&a, 0, sizeof(a));
        a.x = x; a.wt = wt; a.wt_frag = frag_of(wt, cin, cout); a.wt_frag_prec = wf_prec; a.tbl = tbl; a.ld = geo.ld; a.K = K; a.flip = flip; a.n_out = n_out; a.cin = cin; a.cout = cout;
        a.res = res; a.y = y;
        geo.pairs_of(tbl, a.pairs, a.pairs_tile);
        return urn_gconv_fwd_ex(&a, nullptr, st);
    }
    // weight gradient: on the compacted rule list of the forward table when the geometry carries one (two stages, no
    // atomics; `scratch` = its partial slab, carved by dw_scratch), else on the dense table (fp32 atomics)
    void *dw_scratch(const ConvP &c, const int32_t *tbl_f, int64_t n_out)
    {
        const int32_t *list; int tile;
        if (!g_dw_pairs) {
            // the dense-table kernel: partial slabs + fixed-order reduce (dw_2stage), or fp32 atomics
            // ONE slab region for all of them (a launch writes at most ~ workgroups x tile floats whatever the level size,
            // and the weight gradients of a backward pass run one after the other on one stream)
            if (!g_dw_2stage || c.cin % 16 || c.cout % 16) return nullptr;
            if (!dw2_shared) dw2_shared = arena.alloc_bytes((size_t)urn_gconv_dw_2stage_scratch_max());
            return dw2_shared;
        }
        geo.pairs_of(tbl_f, list, tile);
        if (arena.dry) tile = 64;   // workspace sizing: the lists arrive with the real forward (the size does not depend on the tile)
        if (!tile || c.cin % 16 || c.cout % 16) return nullptr;
        return arena.alloc_bytes((size_t)urn_gconv_dw_pairs_scratch_bytes(n_out, tile, c.K, c.cin, c.cout));
    }
    int dw_call(const ConvP &c, const BNP *xf, const float *dy, int64_t ld_dy, const int32_t *tbl_f, int64_t n_out, hipStream_t ws,
                void *scratch)
    {
        const int32_t *list; int tile;
        if (g_net_skip_dw) return URN_OK;
        geo.pairs_of(tbl_f, list, tile);
        if (scratch && !g_dw_pairs) {
            if (!dw2_stream) dw2_stream = ws;
            if (dw2_stream == ws)      // (a launch that could not fork onto the side stream must not share the slab: atomics)
                return urn_gconv_bwd_dw_2stage(c.x, xf ? xf->scale : nullptr, xf ? xf->shift : nullptr, dy, ld_dy > 0 ? ld_dy : c.cout,
                                               tbl_f, geo.ld, c.K, n_out, c.cin, c.cout, grads + c.w, scratch,
                                               urn_gconv_dw_2stage_scratch_max(), ws);
            scratch = nullptr;
        }
        if (scratch && tile)
            return urn_gconv_bwd_dw_pairs(c.x, 0, xf ? xf->scale : nullptr, xf ? xf->shift : nullptr, dy, ld_dy, list, tile, c.K,
                                          n_out, c.cin, c.cout, grads + c.w, scratch,
                                          urn_gconv_dw_pairs_scratch_bytes(n_out, tile, c.K, c.cin, c.cout), ws);
        return urn_gconv_bwd_dw_strided(c.x, xf ? xf->scale : nullptr, xf ? xf->shift : nullptr, dy, ld_dy > 0 ? ld_dy : c.cout,
                                        tbl_f, geo.ld, c.K, n_out, c.cin, c.cout, grads + c.w, ws);
    }
    float *conv_fwd(ConvP &c, const float *x, const int32_t *tbl, int64_t n_out, const float *res)
    {
        c.x = x;
        float *y = arena.f32(n_out * c.cout);
        if (live()) check(gconv_plain(x, wt_all + c.w, tbl, c.K, 0, n_out, c.cin, c.cout, res, y));
        return y;
    }
    // returns dx (n_in, cin); accumulates dW into grads
    float *conv_bwd(ConvP &c, const float *dy, const int32_t *tbl_f, const int32_t *tbl_b, int flip_b, int64_t n_out,
                    int64_t n_in, bool need_dx)
    {
        float *dx = need_dx ? arena.f32(n_in * c.cin) : nullptr;
        void *scratch = dw_scratch(c, tbl_f, n_out);
        if (live()) {
            if (need_dx) check(gconv_plain(dy, params + c.w, tbl_b, c.K, flip_b, n_in, c.cout, c.cin, nullptr, dx));
            hipStream_t ws = st;
            if (side && !events.empty()) {
                // fork: dy (and everything before it) is complete on the main stream at this point
                hipEvent_t e = events[ev_next++ % events.size()];
                if (hipEventRecord(e, st) == hipSuccess && hipStreamWaitEvent(side, e, 0) == hipSuccess) {
                    ws = side;
                    side_used = true;
                }
            }
            check(dw_call(c, nullptr, dy, 0, tbl_f, n_out, ws, scratch));
        }
        return dx;
    }
    float *bn_fwd(BNP &b, const float *x, int64_t n)
    {
        b.x = x;
        float *y = arena.f32(n * b.c);
        b.mean = arena.f32(b.c); b.invstd = arena.f32(b.c);
        void *scratch = arena.alloc_bytes((size_t)urn_bn_scratch_bytes(b.c));
        b.y = y;
        if (live()) {
            float *rm = running ? running + b.run : nullptr;
            float *rv = running ? running + b.run + b.c : nullptr;
            if (training) {
                check(urn_bn_relu_fwd(x, n, b.c, params + b.w, params + b.b, eps, 1, y, b.mean, b.invstd, rm, rv, momentum,
                                      scratch, st));
            } else {
                urn_set_error("urn_net_forward: eval mode runs through the per-layer path");
                check(URN_EUNSUPPORTED);
            }
        }
        return y;
    }
    float *bn_bwd(BNP &b, const float *dy, int64_t n)
    {
        float *dx = arena.f32(n * b.c);
        void *scratch = arena.alloc_bytes((size_t)urn_bn_scratch_bytes(b.c));
        if (live())
            check(urn_bn_relu_bwd(b.x, b.y, dy, n, b.c, params + b.w, b.mean, b.invstd, 1, dx, grads + b.w, grads + b.b, scratch, st));
        return dx;
    }

    // ---- composite forward ----------------------------------------------------------
    float *block_fwd(Block &k, const float *x, int l)
    {
        const int64_t n = geo.n[l];
        const int32_t *nbr = geo.nbr[l];
        const float *sc = x;
        if (k.has_nin) sc = conv_fwd(k.nin, x, nbr + 13 * geo.ld, n, nullptr);
        float *t = bn_fwd(k.bn1, x, n);
        t = conv_fwd(k.conv1, t, nbr, n, nullptr);
        t = bn_fwd(k.bn2, t, n);
        return conv_fwd(k.conv2, t, nbr, n, sc);   // residual add in the epilogue
    }
    float *u_fwd(ULevel &lv, float *x, int l)
    {
        for (auto &k : lv.pre) x = block_fwd(k, x, l);
        if (lv.has_sub) {
            const int64_t n = geo.n[l], nc_ = geo.n[l + 1];
            const int P = planes[l];
            float *t = bn_fwd(lv.bn_d, x, n);
            t = conv_fwd(lv.down, t, geo.chd[l], nc_, nullptr);
            t = u_fwd(*lv.sub, t, l + 1);
            t = bn_fwd(lv.bn_u, t, nc_);
            t = conv_fwd(lv.up, t, geo.up[l], n, nullptr);
            float *cat = arena.f32(n * 2 * P);
            if (live()) {
                check(hipMemcpy2DAsync(cat, 2 * P * 4, x, P * 4, P * 4, n, hipMemcpyDeviceToDevice, st) == hipSuccess ? URN_OK : URN_EHIP);
                check(hipMemcpy2DAsync(cat + P, 2 * P * 4, t, P * 4, P * 4, n, hipMemcpyDeviceToDevice, st) == hipSuccess ? URN_OK : URN_EHIP);
            }
            x = cat;
            for (auto &k : lv.post) x = block_fwd(k, x, l);
        }
        return x;
    }

    // ---- composite backward ---------------------------------------------------------
    float *add_into_new(const float *a, const float *b, int64_t count);   // defined below (needs a kernel)

    float *block_bwd(Block &k, const float *dy, int l)
    {
        const int64_t n = geo.n[l];
        const int32_t *nbr = geo.nbr[l];
        float *d = conv_bwd(k.conv2, dy, nbr, nbr, 1, n, n, true);
        d = bn_bwd(k.bn2, d, n);
        d = conv_bwd(k.conv1, d, nbr, nbr, 1, n, n, true);
        d = bn_bwd(k.bn1, d, n);
        const float *dsc = dy;
        if (k.has_nin) dsc = conv_bwd(k.nin, dy, nbr + 13 * geo.ld, nbr + 13 * geo.ld, 0, n, n, true);
        return add_into_new(d, dsc, n * k.bn1.c);
    }
    float *u_bwd(ULevel &lv, float *dy, int l)
    {
        if (lv.has_sub) {
            const int64_t n = geo.n[l], nc_ = geo.n[l + 1];
            const int P = planes[l];
            for (int i = (int)lv.post.size() - 1; i >= 0; --i) dy = block_bwd(lv.post[i], dy, l);
            // dy is (n, 2P): split into skip and up-branch gradients
            float *dskip = arena.f32(n * P), *dz = arena.f32(n * P);
            if (live()) {
                check(hipMemcpy2DAsync(dskip, P * 4, dy, 2 * P * 4, P * 4, n, hipMemcpyDeviceToDevice, st) == hipSuccess ? URN_OK : URN_EHIP);
                check(hipMemcpy2DAsync(dz, P * 4, dy + P, 2 * P * 4, P * 4, n, hipMemcpyDeviceToDevice, st) == hipSuccess ? URN_OK : URN_EHIP);
            }
            float *d = conv_bwd(lv.up, dz, geo.up[l], geo.chd[l], 0, n, nc_, true);
            d = bn_bwd(lv.bn_u, d, nc_);
            d = u_bwd(*lv.sub, d, l + 1);
            d = conv_bwd(lv.down, d, geo.chd[l], geo.up[l], 0, nc_, n, true);
            d = bn_bwd(lv.bn_d, d, n);
            dy = add_into_new(d, dskip, n * P);
        }
        for (int i = (int)lv.pre.size() - 1; i >= 0; --i) dy = block_bwd(lv.pre[i], dy, l);
        return dy;
    }

    // ==== fused path: BatchNorm+ReLU folded into the gather convolutions =================
    // forward : producer conv writes column partials [epilogue 1] -> bn_finalize -> consumer conv
    //           applies relu(x*scale+shift) on load; the normalised tensor is never written.
    // backward: dX conv masks with the ReLU and reduces (sum g, sum g*xhat) [epilogue 2] ->
    //           finalize (dgamma, dbeta, coefficients) -> apply (+ residual-branch gradient).
    int fused = 1;

    // a BatchNorm that consumes a conv output: the producing kernel finalizes its statistics (columns
    // [coff, coff + cout) of the BatchNorm when the consumer is the second half of a channel concat)
    unsigned long fwd_stamp = 0;
    uint32_t *sync_word = nullptr;
    // accumulated statistics (default): conv epilogues add their column sums into [SUM_SLOTS][2][c] slabs carved
    // from one zeroed region per pass; the consuming kernels derive the BatchNorm coefficients themselves, so the
    // chain conv -> finalize -> conv loses its middle launch (94 launches per cfg3 step)
    static constexpr int SUM_SLOTS = URN_SUM_SLOTS;
    int slab_stats = 0;                  // URN_NET_SLAB_STATS: per-workgroup slabs + finalize launches instead
    double *sums_base = nullptr;
    size_t sums_cap = 0, sums_off = 0;   // in doubles
    int64_t sum_channels = 0;            // sum over convs of max(cin, cout): bounds either pass
    bool sums_mode() const { return fused && !slab_stats; }
    void sums_begin()
    {
        sums_cap = (size_t)sum_channels * SUM_SLOTS * 2;
        sums_base = (double *)arena.alloc_bytes(sums_cap * 8);
        sums_off = 0;
        if (live()) check(hipMemsetAsync(sums_base, 0, sums_cap * 8, st) == hipSuccess ? URN_OK : URN_EHIP);
    }
    double *sums_alloc(int c)
    {
        const size_t need = (size_t)SUM_SLOTS * 2 * c;
        if (sums_off + need > sums_cap) { if (!arena.dry) { urn_set_error("urn_net: statistics region exhausted"); check(URN_EINVAL); } return nullptr; }
        double *p = arena.dry ? (double *)(uintptr_t)256 : sums_base + sums_off;
        sums_off += need;
        return p;
    }

    void alloc_bn(BNP &b)
    {
        if (b.stamp == fwd_stamp) return;
        b.stamp = fwd_stamp;
        b.mean = arena.f32(b.c); b.invstd = arena.f32(b.c); b.scale = arena.f32(b.c); b.shift = arena.f32(b.c);
    }
    // dst / ld_dst: write the output as a column block of a wider matrix (the up-branch half of a channel concat)
    Act conv_f(ConvP &c, const Act &in, BNP *xf, const int32_t *tbl, int64_t n_out, const float *res, Cons c0,
               Cons c1 = Cons(), float *dst = nullptr, int64_t ld_dst = 0)
    {
        c.x = in.x;
        Act y;
        y.n = n_out; y.c = c.cout;
        y.x = dst ? dst : arena.f32(n_out * c.cout);
        const bool mfma = (c.cin % 16 == 0) && (c.cout % 16 == 0);
        const bool stats = c0.bn != nullptr && training;   // eval: running statistics, nothing to collect
        const bool stem16 = c.cin == 1 && (c.cout == 16 || c.cout == 32 || c.cout == 64) && c.K == 27 && !res && !xf && !dst;   // k_gconv_stem
        const bool acc = sums_mode() && (mfma || stem16);   // accumulated statistics on the producing side
        const bool xs = xf && in.st.part != nullptr;  // ... and on the consuming side (the input carries its sums)
        double *part = nullptr;
        if (stats) {
            if (acc) {
                part = sums_alloc(c.cout);
                y.st.part = part; y.st.n_part = SUM_SLOTS; y.st.ld = c.cout;
            } else {
                part = (double *)arena.alloc_bytes(mfma ? (size_t)urn_gconv_part_bytes(n_out, c.cout)
                                                        : (size_t)urn_bn_scratch_bytes(c.cout));
                alloc_bn(*c0.bn);
                if (c1.bn) alloc_bn(*c1.bn);
            }
        }
        if (xs) alloc_bn(*xf);
        if (xf) xf->nrows = in.n;
        if (!live()) return y;
        urn_gconv_args a;
        memset(&a, 0, sizeof(a));
        a.x = in.x; a.wt = wt_all + c.w; a.wt_frag = frag_of(a.wt, c.cin, c.cout); a.wt_frag_prec = wf_prec; a.tbl = tbl; a.ld = geo.ld; a.K = c.K; a.flip = 0; a.n_out = n_out;
        a.cin = c.cin; a.cout = c.cout; a.res = res; a.y = y.x;
        a.ldy = dst ? ld_dst : 0;
        geo.pairs_of(tbl, a.pairs, a.pairs_tile);
        a.fin_eps = eps; a.fin_momentum = momentum;
        if (xs) {
            a.xs_slots = SUM_SLOTS; a.xs_n = in.n;
            a.xs_sums[0] = in.st.part; a.xs_ld[0] = in.st.ld;
            a.xs_split = in.st2.part ? in.c0 : c.cin;
            if (in.st2.part) { a.xs_sums[1] = in.st2.part; a.xs_ld[1] = in.st2.ld; }
            a.xs_gamma = params + xf->w; a.xs_beta = params + xf->b;
            a.xs_mean = xf->mean; a.xs_invstd = xf->invstd; a.xs_scale = xf->scale; a.xs_shift = xf->shift;
            a.xs_running_mean = running ? running + xf->run : nullptr;
            a.xs_running_var = running ? running + xf->run + xf->c : nullptr;
        } else if (xf) {
            a.xf_scale = xf->scale; a.xf_shift = xf->shift;   // from the stem's finalize launch, or from eval_coeffs()
        }
        const Cons cons[2] = {c0, c1};
        if (stats && acc) {
            a.epilogue = 1; a.part = part; a.part_slots = SUM_SLOTS;
        } else if (stats && mfma) {
            a.epilogue = 1; a.part = part; a.sync_word = sync_word; a.fin_n = n_out;
            for (int i = 0; i < 2; ++i) {
                if (!cons[i].bn) continue;
                BNP &b = *cons[i].bn;
                const int o = cons[i].coff;
                a.fin_bn[i].gamma = params + b.w + o; a.fin_bn[i].beta = params + b.b + o;
                a.fin_bn[i].mean = b.mean + o; a.fin_bn[i].invstd = b.invstd + o;
                a.fin_bn[i].scale = b.scale + o; a.fin_bn[i].shift = b.shift + o;
                a.fin_bn[i].running_mean = running ? running + b.run + o : nullptr;
                a.fin_bn[i].running_var = running ? running + b.run + b.c + o : nullptr;
            }
        }
        int n_part = 0;
        check(urn_gconv_fwd_ex(&a, &n_part, st));
        if (stats && !mfma && !acc) {   // the VALU fallback has no epilogue: separate passes
            check(urn_bn_stats_partial(y.x, n_out, c.cout, part, &n_part, st));
            for (int i = 0; i < 2; ++i) {
                if (!cons[i].bn) continue;
                BNP &b = *cons[i].bn;
                const int o = cons[i].coff;
                check(urn_bn_finalize_fwd(part, n_part, n_out, c.cout, c.cout, eps, params + b.w + o, params + b.b + o, b.mean + o,
                                          b.invstd + o, b.scale + o, b.shift + o, running ? running + b.run + o : nullptr,
                                          running ? running + b.run + b.c + o : nullptr, momentum, st));
            }
        }
        return y;
    }
    // Weight gradients go to the side stream behind a fork (event record on the main stream, wait on the side stream).
    // They can be queued and forked in groups (g_dw_group > 1) -- measured slower: the later a weight gradient
    // starts, the less of it overlaps with the dX chain -- so the default is one fork per convolution.
    struct DwJob { ConvP *c; const BNP *xf; const float *dy; const int32_t *tbl_f; int64_t n_out; int64_t ld_dy; void *scratch; };
    std::vector<DwJob> dw_queue;
    void dw_flush()
    {
        if (dw_queue.empty()) return;
        hipStream_t ws = st;
        if (side && !events.empty()) {
            const bool second = g_net_side2 && side2 && (dw_toggle++ & 1);
            hipStream_t target = second ? side2 : side;
            hipEvent_t e = events[ev_next++ % events.size()];
            if (hipEventRecord(e, st) == hipSuccess && hipStreamWaitEvent(target, e, 0) == hipSuccess) {
                ws = target;
                if (second) side2_used = true; else side_used = true;
            }
        }
        for (const DwJob &j : dw_queue) check(dw_call(*j.c, j.xf, j.dy, j.ld_dy, j.tbl_f, j.n_out, ws, j.scratch));
        dw_queue.clear();
    }
    void dw_launch(ConvP &c, const BNP *xf, const float *dy, const int32_t *tbl_f, int64_t n_out, int64_t ld_dy, void *scratch)
    {
        dw_queue.push_back(DwJob{&c, xf, dy, tbl_f, n_out, ld_dy, scratch});
        if ((int)dw_queue.size() >= g_dw_group) dw_flush();
    }
    // backward of conv(BNReLU_b(x)): returns d/dx (raw input of the BatchNorm), adds `extra` when given
    // ld_dy / ld_extra: dy / extra are column blocks of a wider matrix (the halves of a concat's gradient); 0 = dense
    float *conv_b_fused(ConvP &c, BNP &b, const float *dy, const int32_t *tbl_f, const int32_t *tbl_b, int flip_b,
                        int64_t n_out, int64_t n_in, const float *extra, int64_t ld_dy = 0, int64_t ld_extra = 0)
    {
        float *g = arena.f32(n_in * c.cin);
        const bool acc = sums_mode();
        double *part = acc ? sums_alloc(c.cin) : (double *)arena.alloc_bytes((size_t)urn_gconv_part_bytes(n_in, c.cin));
        float *coef = acc ? nullptr : arena.f32(2 * (int64_t)c.cin);
        float *dx = arena.f32(n_in * c.cin);
        void *scratch = dw_scratch(c, tbl_f, n_out);
        if (live()) {
            dw_launch(c, &b, dy, tbl_f, n_out, ld_dy, scratch);
            urn_gconv_args a;
            memset(&a, 0, sizeof(a));
            a.x = dy; a.wt = params + c.w; a.wt_frag = frag_of(a.wt, c.cout, c.cin); a.wt_frag_prec = wf_prec; a.tbl = tbl_b; a.ld = geo.ld; a.K = c.K; a.flip = flip_b; a.n_out = n_in;
            a.cin = c.cout; a.cout = c.cin; a.y = g; a.ldx = ld_dy;
            geo.pairs_of(tbl_b, a.pairs, a.pairs_tile);
            a.epilogue = 2; a.part = part;
            a.e_x = b.x; a.e_scale = b.scale; a.e_shift = b.shift; a.e_mean = b.mean; a.e_invstd = b.invstd;
            int n_part = 0;
            if (acc) {
                a.part_slots = SUM_SLOTS;
                check(urn_gconv_fwd_ex(&a, &n_part, st));
                check(urn_bn_bwd_apply_sums(b.x, g, extra, ld_extra, n_in, c.cin, params + b.w, b.mean, b.invstd, part, SUM_SLOTS,
                                            grads + b.w, grads + b.b, dx, st));
            } else {
                a.sync_word = sync_word; a.fin_n = n_in;
                a.fin_dgamma = grads + b.w; a.fin_dbeta = grads + b.b; a.fin_coef0 = coef; a.fin_coef1 = coef + c.cin;
                check(urn_gconv_fwd_ex(&a, &n_part, st));
                check(urn_bn_bwd_apply(b.x, g, extra, n_in, c.cin, params + b.w, b.mean, b.invstd, coef, coef + c.cin, dx, st));
            }
        }
        return dx;
    }
    // eval mode: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale (and mean / invstd for the
    // last BatchNormReLU) of EVERY BatchNorm, one launch; the convolutions then fold them like the batch statistics
    std::vector<BNP *> all_bns;
    void collect_bns(ULevel &lv)
    {
        for (auto &k : lv.pre) { all_bns.push_back(&k.bn1); all_bns.push_back(&k.bn2); }
        if (lv.has_sub) { all_bns.push_back(&lv.bn_d); collect_bns(*lv.sub); all_bns.push_back(&lv.bn_u); }
        for (auto &k : lv.post) { all_bns.push_back(&k.bn1); all_bns.push_back(&k.bn2); }
    }
    void eval_coeffs();   // defined below (needs a kernel)

    // out0/out1: the BatchNorms that consume this block's output
    Act block_f(Block &k, const Act &x, int l, Cons out0, Cons out1 = Cons())
    {
        const int64_t n = geo.n[l];
        const int32_t *nbr = geo.nbr[l];
        const float *sc = x.x;
        if (k.has_nin) sc = conv_f(k.nin, x, nullptr, nbr + 13 * geo.ld, n, nullptr, Cons()).x;
        Cons c_bn2; c_bn2.bn = &k.bn2;
        Act t = conv_f(k.conv1, x, &k.bn1, nbr, n, nullptr, c_bn2);
        k.bn2.x = t.x;
        return conv_f(k.conv2, t, &k.bn2, nbr, n, sc, out0, out1);
    }
    // out: the BatchNorm that consumes this U's output
    Act u_f(ULevel &lv, Act x, int l, Cons out)
    {
        const int P = planes[l];
        for (size_t i = 0; i < lv.pre.size(); ++i) {
            Cons c0, c1;
            if (i + 1 < lv.pre.size()) c0.bn = &lv.pre[i + 1].bn1;
            else if (lv.has_sub) { c0.bn = &lv.bn_d; c1.bn = &lv.post[0].bn1; }
            else c0 = out;
            x = block_f(lv.pre[i], x, l, c0, c1);
            if (i + 1 < lv.pre.size()) lv.pre[i + 1].bn1.x = x.x;
            else if (lv.has_sub) lv.bn_d.x = x.x;
        }
        if (lv.has_sub) {
            const int64_t n = geo.n[l], nc_ = geo.n[l + 1];
            Cons c_sub; c_sub.bn = &lv.sub->pre[0].bn1;
            Act t = conv_f(lv.down, x, &lv.bn_d, geo.chd[l], nc_, nullptr, c_sub);
            lv.sub->pre[0].bn1.x = t.x;
            Cons c_up; c_up.bn = &lv.bn_u;
            t = u_f(*lv.sub, t, l + 1, c_up);
            lv.bn_u.x = t.x;
            Cons c_cat; c_cat.bn = &lv.post[0].bn1; c_cat.coff = P;
            Act cat;
            cat.n = n; cat.c = 2 * P;
            cat.x = arena.f32(n * 2 * P);
            // channel concat [skip | up-branch]: the up-conv writes its half in place (row stride 2P) when the kernel
            // supports it (accumulated-statistics mode = 2-D tile kernel); the skip half is one 2-D copy
            const bool in_place = sums_mode();
            Act z = conv_f(lv.up, t, &lv.bn_u, geo.up[l], n, nullptr, c_cat, Cons(), in_place ? cat.x + P : nullptr, 2 * P);
            cat.st = x.st; cat.st2 = z.st; cat.c0 = P;   // statistics of the two halves come from their producers
            if (live()) {
                check(hipMemcpy2DAsync(cat.x, 2 * P * 4, x.x, P * 4, P * 4, n, hipMemcpyDeviceToDevice, st) == hipSuccess ? URN_OK : URN_EHIP);
                if (!in_place)
                    check(hipMemcpy2DAsync(cat.x + P, 2 * P * 4, z.x, P * 4, P * 4, n, hipMemcpyDeviceToDevice, st) == hipSuccess ? URN_OK : URN_EHIP);
            }
            lv.post[0].bn1.x = cat.x;
            x = cat;
            for (size_t i = 0; i < lv.post.size(); ++i) {
                Cons c0;
                if (i + 1 < lv.post.size()) c0.bn = &lv.post[i + 1].bn1;
                else c0 = out;
                x = block_f(lv.post[i], x, l, c0);
                if (i + 1 < lv.post.size()) lv.post[i + 1].bn1.x = x.x;
            }
        }
        return x;
    }
    float *block_b(Block &k, const float *dy, int l)
    {
        const int64_t n = geo.n[l];
        const int32_t *nbr = geo.nbr[l];
        const float *dsc = dy;
        if (k.has_nin) dsc = conv_bwd(k.nin, dy, nbr + 13 * geo.ld, nbr + 13 * geo.ld, 0, n, n, true);
        float *d = conv_b_fused(k.conv2, k.bn2, dy, nbr, nbr, 1, n, n, nullptr);
        return conv_b_fused(k.conv1, k.bn1, d, nbr, nbr, 1, n, n, dsc);   // + gradient of the shortcut branch
    }
    float *u_b(ULevel &lv, float *dy, int l)
    {
        if (lv.has_sub) {
            const int64_t n = geo.n[l], nc_ = geo.n[l + 1];
            const int P = planes[l];
            for (int i = (int)lv.post.size() - 1; i >= 0; --i) dy = block_b(lv.post[i], dy, l);
            // dy is (n, 2P) = [gradient of the skip half | gradient of the up-branch half]: both halves are consumed in
            // place through row strides (accumulated-statistics mode); otherwise split by two 2-D copies
            float *d;
            if (sums_mode()) {
                const float *dcat = dy;
                d = conv_b_fused(lv.up, lv.bn_u, dcat + P, geo.up[l], geo.chd[l], 0, n, nc_, nullptr, 2 * P, 0);
                d = u_b(*lv.sub, d, l + 1);
                dy = conv_b_fused(lv.down, lv.bn_d, d, geo.chd[l], geo.up[l], 0, nc_, n, dcat, 0, 2 * P);   // + skip-path gradient
            } else {
                float *dskip = arena.f32(n * P), *dz = arena.f32(n * P);
                if (live()) {
                    check(hipMemcpy2DAsync(dskip, P * 4, dy, 2 * P * 4, P * 4, n, hipMemcpyDeviceToDevice, st) == hipSuccess ? URN_OK : URN_EHIP);
                    check(hipMemcpy2DAsync(dz, P * 4, dy + P, 2 * P * 4, P * 4, n, hipMemcpyDeviceToDevice, st) == hipSuccess ? URN_OK : URN_EHIP);
                }
                d = conv_b_fused(lv.up, lv.bn_u, dz, geo.up[l], geo.chd[l], 0, n, nc_, nullptr);
                d = u_b(*lv.sub, d, l + 1);
                dy = conv_b_fused(lv.down, lv.bn_d, d, geo.chd[l], geo.up[l], 0, nc_, n, dskip);   // + skip-path gradient
            }
        }
        for (int i = (int)lv.pre.size() - 1; i >= 0; --i) dy = block_b(lv.pre[i], dy, l);
        if (!lv.has_sub && bottom_cb && live()) {
            // bottom level done: the gradients of [bottom blocks | decoder | last BatchNorm | head] -- a contiguous suffix of the
            // flat buffer (registration order) -- are complete once what is queued so far has run.  Order the side stream
            // behind the caller's stream here, so that a collective issued on the side stream's tail sees both.
            dw_flush();
            if (side && !events.empty()) {
                hipEvent_t e = events[ev_next++ % events.size()];
                check(hipEventRecord(e, st) == hipSuccess && hipStreamWaitEvent(side, e, 0) == hipSuccess ? URN_OK : URN_EHIP);
                side_used = true;
            }
            bottom_cb(bottom_user);
        }
        return dy;
    }
};

__global__ void k_add2(const float *__restrict__ a, const float *__restrict__ b, long n, float *__restrict__ o)
{
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        f32x4 x = *(const f32x4 *)(a + i), y = *(const f32x4 *)(b + i);
        *(f32x4 *)(o + i) = x + y;
    } else {
        for (; i < n; ++i) o[i] = a[i] + b[i];
    }
}

float *urn_net::add_into_new(const float *a, const float *b, int64_t count)
{
    float *o = arena.f32(count);
    if (live() && count > 0) hipLaunchKernelGGL(k_add2, dim3(urn_cdiv((count + 3) / 4, 256)), dim3(256), 0, st, a, b, (long)count, o);
    return o;
}

#define URN_MAX_BNS 128
struct BNDesc { long w, b, run, out; int c; };
struct BNDescs { int n; BNDesc d[URN_MAX_BNS]; };

__global__ void k_bn_eval_coeffs(BNDescs t, const float *__restrict__ params, const float *__restrict__ running, double eps,
                                 float *__restrict__ coef)
{
    const BNDesc d = t.d[blockIdx.x];
    for (int e = threadIdx.x; e < d.c; e += blockDim.x) {
        const float mean = running[d.run + e], var = running[d.run + d.c + e];
        const float is = (float)(1.0 / sqrt((double)var + eps));
        const float sc = params[d.w + e] * is;
        coef[d.out + e] = mean;                       // [mean | invstd | scale | shift], c each
        coef[d.out + d.c + e] = is;
        coef[d.out + 2 * d.c + e] = sc;
        coef[d.out + 3 * d.c + e] = fmaf(-mean, sc, params[d.b + e]);
    }
}

void urn_net::eval_coeffs()
{
    if (all_bns.empty()) { collect_bns(u); all_bns.push_back(&bn_out); }
    long total = 0;
    for (BNP *b : all_bns) total += 4L * b->c;
    float *coef = arena.f32(total);
    BNDescs t;
    long off = 0;
    size_t i = 0;
    while (i < all_bns.size()) {
        t.n = 0;
        for (; i < all_bns.size() && t.n < URN_MAX_BNS; ++i) {
            BNP *b = all_bns[i];
            t.d[t.n++] = BNDesc{(long)b->w, (long)b->b, (long)b->run, off, b->c};
            b->mean = coef + off; b->invstd = coef + off + b->c; b->scale = coef + off + 2 * b->c; b->shift = coef + off + 3 * b->c;
            b->stamp = fwd_stamp;
            off += 4L * b->c;
        }
        if (live()) hipLaunchKernelGGL(k_bn_eval_coeffs, dim3(t.n), dim3(64), 0, st, t, params, running, eps, coef);
    }
}

extern "C" int urn_net_create(int m, int num_levels, int reps, int num_class, double eps, double momentum, int flags,
                              urn_net **out)
{
    URN_CHECK_ARG(out && m > 0 && m % 16 == 0 && num_levels >= 1 && reps >= 1 && num_class > 0, "bad configuration (m must be a multiple of 16)");
    urn_net *n = new urn_net();
    n->fused = (flags & URN_NET_UNFUSED) ? 0 : 1;
    n->slab_stats = (flags & URN_NET_SLAB_STATS) ? 1 : 0;
    n->m = m; n->L = num_levels; n->reps = reps; n->nc = num_class; n->eps = eps; n->momentum = momentum;
    for (int i = 1; i <= num_levels; ++i) n->planes.push_back(i * m);
    n->stem = n->make_conv(27, 1, m);
    n->make_u(n->u, 0);
    n->bn_out = n->make_bn(m);
    // side stream + a ring of events for the fork/join of the weight-gradient kernels; if the runtime is not
    // available (no GPU: build check, workspace sizing) the executor stays single-stream
    // lowest priority: the weight gradients only have to be done by the end of backward; the dX -> BatchNorm chain
    // on the caller's stream is the critical path and should win the CUs when both have work
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (!(flags & URN_NET_SINGLE_STREAM) &&
        hipStreamCreateWithPriority(&n->side, hipStreamNonBlocking, prio_lo) == hipSuccess) {
        for (int i = 0; i < 128; ++i) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) break;
            n->events.push_back(e);
        }
        if (g_net_side2 && hipStreamCreateWithPriority(&n->side2, hipStreamNonBlocking, prio_lo) != hipSuccess) { n->side2 = nullptr; (void)hipGetLastError(); }
    } else {
        n->side = nullptr;
        (void)hipGetLastError();
    }
    *out = n;
    return URN_OK;
}

extern "C" void urn_net_destroy(urn_net *n)
{
    if (!n) return;
    for (auto e : n->events) (void)hipEventDestroy(e);
    if (n->ev_w) (void)hipEventDestroy(n->ev_w);
    if (n->side) (void)hipStreamDestroy(n->side);
    if (n->side2) (void)hipStreamDestroy(n->side2);
    delete n;
}

// ---- batched weight transpose: every conv weight (K, a, b) -> (K, b, a) in one launch ----------
#define URN_MAX_CONVS 96
struct TDesc { int K, a, b; long off; };
struct TDescs { int n; TDesc d[URN_MAX_CONVS]; };

// fragment-ordered copies (urn_gconv_args.wt_frag) of every conv weight with channel counts that are multiples of 16:
// blockIdx.z = 0: of the transposed weights (forward operand), 1: of the parameters themselves (the input gradient's
// operand: (K, cin, cout) read as (K, "cout" = cin, "cin" = cout)).  Thread = one 16-byte piece of the destination.
template <int PREC>
__global__ void k_fragments_all(TDescs t, const float *__restrict__ wt_all, const float *__restrict__ params,
                                float *__restrict__ wf_fwd, float *__restrict__ wf_bwd)
{
    const TDesc d = t.d[blockIdx.y];
    if (d.a % 16 || d.b % 16) return;
    const bool bwd = blockIdx.z != 0;
    const float *src = (bwd ? params : wt_all) + d.off;
    float *dst = (bwd ? wf_bwd : wf_fwd) + d.off;
    const int cout = bwd ? d.a : d.b, cin = bwd ? d.b : d.a;
    const int kbn = cin / 16, cbn = cout / 16;
    const long total4 = (long)d.K * d.a * d.b / 4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(e & 63), r = lane & 15, q = lane >> 4;
        long f = e >> 6;
        const int kb = (int)(f % kbn); f /= kbn;
        const int cb = (int)(f % cbn); const long o = f / cbn;
        const f32x4 v = *(const f32x4 *)(src + ((long)o * cout + 16 * cb + r) * cin + 16 * kb + 4 * q);
        // PREC != 0: 16-bit fragments (urn_gconv_args.wt_frag_prec) in the first half of the conv's region
        if constexpr (PREC == 0) *(f32x4 *)(dst + e * 4) = v;
        else ((uint2 *)dst)[urn_frag16_slot(e, kbn)] = urn_round16x4<PREC>(v);
    }
}

// wt[o][j][i] = w[o][i][j] for every conv: 32 x 32 tiles through LDS (both sides coalesced; the one-element-per-thread
// form read with a stride of b floats: 428 us for the 30 M parameters of the 768^3 / uf 32 / uns 7 network, 240 MB of traffic)
__global__ __launch_bounds__(256) void k_transpose_all(TDescs t, const float *__restrict__ params, float *__restrict__ wt)
{
    __shared__ float s_t[32][33];
    const TDesc d = t.d[blockIdx.y];
    const int ti_n = (d.a + 31) / 32, tj_n = (d.b + 31) / 32;
    const long per = (long)d.a * d.b, ntiles = (long)d.K * ti_n * tj_n;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {     // workgroup-uniform
        const long o = tile / (ti_n * tj_n);
        const int rem = (int)(tile - o * (ti_n * tj_n));
        const int i0 = (rem / tj_n) * 32, j0 = (rem % tj_n) * 32;
        const float *src = params + d.off + o * per;
        float *dst = wt + d.off + o * per;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + ty + 8 * k, j = j0 + tx;
            if (i < d.a && j < d.b) s_t[ty + 8 * k][tx] = src[(long)i * d.b + j];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + ty + 8 * k, i = i0 + tx;
            if (i < d.a && j < d.b) dst[(long)j * d.a + i] = s_t[tx][ty + 8 * k];
        }
        __syncthreads();
    }
}

static void collect_convs(urn_net *net, ULevel &lv)
{
    for (auto &k : lv.pre) { if (k.has_nin) net->convs.push_back(&k.nin); net->convs.push_back(&k.conv1); net->convs.push_back(&k.conv2); }
    if (lv.has_sub) {
        net->convs.push_back(&lv.down);
        collect_convs(net, *lv.sub);
        net->convs.push_back(&lv.up);
    }
    for (auto &k : lv.post) { if (k.has_nin) net->convs.push_back(&k.nin); net->convs.push_back(&k.conv1); net->convs.push_back(&k.conv2); }
}

static void collect_all_convs(urn_net *net)
{
    if (net->convs.empty()) {
        net->convs.push_back(&net->stem); collect_convs(net, net->u);
        net->sum_channels = 0;
        for (const ConvP *c : net->convs) net->sum_channels += std::max(c->cin, c->cout);
    }
}

// launches that write the transposed and the fragment-ordered copies of every conv weight
extern int g_opt_precision;   // urn_set_option("gconv_precision")
static void launch_weight_copies(urn_net *net, const float *params, float *wt_all, float *wf_fwd, float *wf_bwd, hipStream_t st)
{
    net->wf_prec = g_opt_precision;
    for (size_t base = 0; base < net->convs.size(); base += URN_MAX_CONVS) {
        TDescs t;
        t.n = (int)std::min((size_t)URN_MAX_CONVS, net->convs.size() - base);
        for (int i = 0; i < t.n; ++i) {
            const ConvP *c = net->convs[base + i];
            t.d[i] = TDesc{c->K, c->cin, c->cout, (long)c->w};
        }
        // workgroups per conv by the largest conv of the batch (the small ones leave at once): 32 for the cfg3 network, 512 for 224 -> 224
        long big = 0;
        for (int i = 0; i < t.n; ++i) big = std::max(big, (long)t.d[i].K * t.d[i].a * t.d[i].b);
        const unsigned gx = (unsigned)std::min(512L, std::max(32L, big / 4096));
        hipLaunchKernelGGL(k_transpose_all, dim3(gx, t.n), dim3(256), 0, st, t, params, wt_all);
        if (wf_fwd) {
            if (net->wf_prec == 1) hipLaunchKernelGGL(k_fragments_all<1>, dim3(std::max(8u, gx / 4), t.n, 2), dim3(256), 0, st, t, (const float *)wt_all, params, wf_fwd, wf_bwd);
            else if (net->wf_prec == 2) hipLaunchKernelGGL(k_fragments_all<2>, dim3(std::max(8u, gx / 4), t.n, 2), dim3(256), 0, st, t, (const float *)wt_all, params, wf_fwd, wf_bwd);
            else hipLaunchKernelGGL(k_fragments_all<0>, dim3(std::max(8u, gx / 4), t.n, 2), dim3(256), 0, st, t, (const float *)wt_all, params, wf_fwd, wf_bwd);
        }
    }
}

static void transpose_all(urn_net *net)
{
    collect_all_convs(net);
    // (the arena copies are always carved, so that the workspace model does not depend on urn_net_prepare_weights)
    net->wt_all = net->arena.f32(net->n_params);
    net->wf_fwd = net->wf_bwd = nullptr;
    if (g_net_wfrag) { net->wf_fwd = net->arena.f32(net->n_params); net->wf_bwd = net->arena.f32(net->n_params); }
    if (!net->live()) return;
    if (net->w_prepared && net->ext_w) {
        // written ahead of this call (urn_net_prepare_weights), possibly on the side stream: order them in front of the convs
        net->w_prepared = false;
        net->wt_all = net->ext_w;
        net->wf_fwd = g_net_wfrag ? net->ext_w + net->n_params : nullptr;
        net->wf_bwd = g_net_wfrag ? net->ext_w + 2 * net->n_params : nullptr;
        if (net->ev_w) net->check(hipStreamWaitEvent(net->st, net->ev_w, 0) == hipSuccess ? URN_OK : URN_EHIP);
        return;
    }
    launch_weight_copies(net, net->params, net->wt_all, net->wf_fwd, net->wf_bwd, net->st);
}

// The weight copies of the coming forward (transposed + both fragment orders; wbuf = 3 * urn_net_param_count floats owned by
// the caller, valid until the backward of that forward has run), written NOW -- on the executor's side stream when it
// has one, behind everything queued on `stream` so far (the optimizer step).  Called before the integer phase is enqueued,
// the ~35 us of copy kernels run beside it instead of in front of the first convolution.  Optional: without it
// urn_net_forward writes the copies itself.
extern "C" int urn_net_prepare_weights(urn_net *net, const float *params, float *wbuf, int64_t wbuf_floats, void *stream)
{
    URN_CHECK_ARG(net && params && wbuf && wbuf_floats >= 3 * net->n_params, "null pointer or buffer smaller than 3 * urn_net_param_count floats");
    collect_all_convs(net);
    hipStream_t st = (hipStream_t)stream, ws = st;
    if (!net->ev_w && hipEventCreateWithFlags(&net->ev_w, hipEventDisableTiming) != hipSuccess) { net->ev_w = nullptr; (void)hipGetLastError(); }
    if (net->side && net->ev_w && !net->events.empty()) {
        hipEvent_t e = net->events[net->ev_next++ % net->events.size()];
        if (hipEventRecord(e, st) == hipSuccess && hipStreamWaitEvent(net->side, e, 0) == hipSuccess) ws = net->side;
    }
    launch_weight_copies(net, params, wbuf, g_net_wfrag ? wbuf + net->n_params : nullptr, g_net_wfrag ? wbuf + 2 * net->n_params : nullptr, ws);
    if (net->ev_w && hipEventRecord(net->ev_w, ws) != hipSuccess) return URN_EHIP;
    if (!net->ev_w && ws != st) return URN_EHIP;
    net->ext_w = wbuf; net->w_prepared = true;
    URN_LAUNCH_CHECK();
    return URN_OK;
}
// ---- test hook: the folded BatchNorm+ReLU of the last training forward, per BatchNorm -------------------------------
// The fused path never writes a normalised tensor; what decides a ReLU mask is relu(x * scale + shift) with the scale /
// shift the kernels derived.  Parity tests read them back (and the BatchNorm's input x) to pin the oracle's masks to
// the GPU's: a pre-activation within fp32 rounding of zero otherwise flips between any two evaluation orders.
extern "C" int urn_net_num_bn(urn_net *net)
{
    if (!net) return -1;
    if (net->all_bns.empty()) { net->collect_bns(net->u); net->all_bns.push_back(&net->bn_out); }
    return (int)net->all_bns.size();
}
extern "C" int urn_net_bn_info(urn_net *net, int i, int64_t *w_off, int64_t *rows, int *c)
{
    URN_CHECK_ARG(net && w_off && rows && c && i >= 0 && i < urn_net_num_bn(net), "bad index");
    const BNP &b = *net->all_bns[i];
    *w_off = b.w; *rows = b.nrows; *c = b.c;
    return URN_OK;
}
extern "C" int urn_net_bn_export(urn_net *net, int i, float *x, float *scale, float *shift, void *stream)
{
    URN_CHECK_ARG(net && x && scale && shift && i >= 0 && i < urn_net_num_bn(net), "bad argument");
    const BNP &b = *net->all_bns[i];
    URN_CHECK_ARG(net->fused && b.x && b.scale && b.shift && b.stamp == net->fwd_stamp, "no fused training forward recorded for this BatchNorm");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(x, b.x, (size_t)b.nrows * b.c * 4, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(scale, b.scale, (size_t)b.c * 4, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(shift, b.shift, (size_t)b.c * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        urn_set_error("urn_net_bn_export: copy failed");
        return URN_EHIP;
    }
    return URN_OK;
}

extern "C" int64_t urn_net_param_count(const urn_net *n) { return n ? n->n_params : -1; }
extern "C" int64_t urn_net_running_count(const urn_net *n) { return n ? n->n_running : -1; }
extern "C" int urn_net_num_tensors(const urn_net *n) { return n ? (int)n->p_off.size() : -1; }
extern "C" int urn_net_tensor(const urn_net *n, int i, int64_t *off, int64_t *numel)
{
    URN_CHECK_ARG(n && off && numel && i >= 0 && i < (int)n->p_off.size(), "bad index");
    *off = n->p_off[i]; *numel = n->p_numel[i];
    return URN_OK;
}

extern "C" int urn_net_set_pairs(urn_net *net, int num_levels, const void *const *nbr_pairs, const void *const *chd_pairs,
                                 const void *const *up_pairs, const int *tile_nbr, const int *tile_chd, const int *tile_up)
{
    URN_CHECK_ARG(net && num_levels == net->L, "level count does not match the network");
    URN_CHECK_ARG((!nbr_pairs || tile_nbr) && (!chd_pairs || tile_chd) && (!up_pairs || tile_up), "lists without tile sizes");
    Geo &g = net->geo;
    g.p_nbr.assign(num_levels, nullptr); g.p_chd.assign(num_levels, nullptr); g.p_up.assign(num_levels, nullptr);
    g.t_nbr.assign(num_levels, 0); g.t_chd.assign(num_levels, 0); g.t_up.assign(num_levels, 0);
    for (int l = 0; l < num_levels; ++l) {
        if (nbr_pairs) { g.p_nbr[l] = (const int32_t *)nbr_pairs[l]; g.t_nbr[l] = tile_nbr[l]; }
        if (l + 1 < num_levels) {
            if (chd_pairs) { g.p_chd[l] = (const int32_t *)chd_pairs[l]; g.t_chd[l] = tile_chd[l]; }
            if (up_pairs) { g.p_up[l] = (const int32_t *)up_pairs[l]; g.t_up[l] = tile_up[l]; }
        }
    }
    net->pairs_armed = true;
    return URN_OK;
}

// first parameter of the bottom level: [that offset, end) of the flat buffers = bottom blocks + decoder + last BatchNorm
extern "C" int64_t urn_net_suffix_offset(const urn_net *net)
{
    if (!net) return -1;
    const ULevel *lv = &net->u;
    while (lv->has_sub) lv = lv->sub.get();
    const Block &k = lv->pre[0];
    return k.has_nin ? k.nin.w : k.bn1.w;
}
extern "C" void *urn_net_side_stream(const urn_net *net) { return net ? (void *)net->side : nullptr; }

extern "C" int urn_net_set_head(urn_net *net, const float *W, const float *b)
{
    URN_CHECK_ARG(net && W, "null pointer");
    if (!net->sums_mode() || net->m > 32 || net->nc > 8) {
        urn_set_error("urn_net_set_head: needs the fused path with accumulated statistics, m <= 32 and num_class <= 8");
        return URN_EUNSUPPORTED;
    }
    net->head_w = W; net->head_b = b; net->head_armed = true;
    return URN_OK;
}

static int set_geo(urn_net *net, int num_levels, int64_t ld, const int64_t *n, const void *const *nbr, const void *const *chd,
                   const void *const *up, const int32_t *row2site, int64_t n_rows)
{
    URN_CHECK_ARG(num_levels == net->L && n, "geometry level count does not match the network");
    Geo &g = net->geo;
    if (!net->pairs_armed) { g.p_nbr.clear(); g.p_chd.clear(); g.p_up.clear(); }   // lists belong to ONE geometry: the next forward only
    net->pairs_armed = false;
    g.L = num_levels; g.ld = ld; g.n_rows = n_rows; g.row2site = row2site;
    g.n.assign(n, n + num_levels);
    g.nbr.assign(num_levels, nullptr); g.chd.assign(num_levels, nullptr); g.up.assign(num_levels, nullptr);
    for (int l = 0; l < num_levels; ++l) {
        if (nbr) g.nbr[l] = (const int32_t *)nbr[l];
        if (l + 1 < num_levels) {
            if (chd) g.chd[l] = (const int32_t *)chd[l];
            if (up) g.up[l] = (const int32_t *)up[l];
        }
    }
    return URN_OK;
}

static void run_forward(urn_net *net, const float *site_feats)
{
    transpose_all(net);
    const int64_t n0 = net->geo.n[0];
    if (net->fused) {
        net->fwd_stamp++;
        net->sync_word = (uint32_t *)net->arena.alloc_bytes(256);
        // (the ticket word is only used by the in-kernel finalize of the slab-statistics mode)
        if (net->live() && !net->sums_mode()) net->check(hipMemsetAsync(net->sync_word, 0, 256, net->st) == hipSuccess ? URN_OK : URN_EHIP);
        if (net->sums_mode()) net->sums_begin();
        if (!net->training) {
            if (!net->running) { urn_set_error("urn_net_forward: eval mode needs the running statistics"); net->check(URN_EINVAL); return; }
            net->eval_coeffs();
        }
        Act f;
        f.x = const_cast<float *>(site_feats); f.n = n0; f.c = 1;
        Cons c_first; c_first.bn = &net->u.pre[0].bn1;
        Act x = net->conv_f(net->stem, f, nullptr, net->geo.nbr[0], n0, nullptr, c_first);
        net->u.pre[0].bn1.x = x.x;
        Cons c_out; c_out.bn = &net->bn_out;
        x = net->u_f(net->u, x, 0, c_out);
        // the last BatchNormReLU feeds the OutputLayer, not a conv
        BNP &b = net->bn_out;
        b.x = x.x; b.nrows = n0;
        net->head_live = net->head_armed && net->sums_mode() && (x.st.part != nullptr || !net->training);
        net->head_armed = false;
        if (net->head_live) {
            // ... with the head armed: BatchNormReLU + OutputLayer + Linear in one kernel, straight to the logits
            net->alloc_bn(b);
            if (net->live()) {
                const bool st = x.st.part && net->training;
                net->check(urn_tail_fwd(x.x, net->geo.row2site, net->geo.n_rows, b.c, net->nc, net->head_w, net->head_b,
                                        st ? x.st.part : nullptr, st ? x.st.n_part : 0, n0, net->eps, net->params + b.w, net->params + b.b,
                                        b.mean, b.invstd, b.scale, b.shift, (st && net->running) ? net->running + b.run : nullptr,
                                        (st && net->running) ? net->running + b.run + b.c : nullptr, net->momentum, net->head_logits, net->st));
            }
            net->trunk_out = x.x;   // (marks "a forward was recorded"; the rows themselves are never formed)
            return;
        }
        // ... otherwise materialise it
        if (x.st.part && net->training) {   // accumulated statistics: the slab is a partial slab of SUM_SLOTS rows
            net->alloc_bn(b);
            if (net->live())
                net->check(urn_bn_finalize_fwd(x.st.part, x.st.n_part, n0, b.c, x.st.ld, net->eps, net->params + b.w, net->params + b.b,
                                               b.mean, b.invstd, b.scale, b.shift, net->running ? net->running + b.run : nullptr,
                                               net->running ? net->running + b.run + b.c : nullptr, net->momentum, net->st));
        }
        float *y = net->arena.f32(n0 * b.c);
        b.y = y;
        if (net->live())
            net->check(urn_bn_relu_apply(x.x, n0, b.c, net->params + b.w, net->params + b.b, b.mean, b.invstd, 1, y, net->st));
        net->trunk_out = y;
        return;
    }
    float *x = net->conv_fwd(net->stem, site_feats, net->geo.nbr[0], n0, nullptr);
    x = net->u_fwd(net->u, x, 0);
    net->trunk_out = net->bn_fwd(net->bn_out, x, n0);
}

static void run_backward(urn_net *net, const float *d_rows)
{
    // OutputLayer backward: scatter-add input-row gradients onto sites
    const int64_t n0 = net->geo.n[0];
    net->dw2_shared = nullptr; net->dw2_stream = nullptr;
    float *d = net->arena.f32(n0 * net->m);
    if (net->head_live) {
        // the armed head: d_rows are the logits' gradients.  One kernel forms the masked gradient of the last BatchNormReLU's
        // output on the sites (+ the head's parameter gradients + the BatchNorm-backward column sums), the apply finishes it
        BNP &b = net->bn_out;
        net->sums_begin();
        double *part = net->sums_alloc(b.c);
        float *dx = net->arena.f32(n0 * b.c);
        if (net->live()) {
            // (one row per site: the kernel stores every element of d, nothing to zero)
            if (net->geo.n_rows != n0) net->check(hipMemsetAsync(d, 0, (size_t)n0 * net->m * 4, net->st) == hipSuccess ? URN_OK : URN_EHIP);
            float *gw = net->grads + net->n_params;
            net->check(urn_tail_bwd(d_rows, b.x, net->geo.row2site, net->geo.n_rows, b.c, net->nc, net->head_w, b.scale, b.shift, b.mean,
                                    b.invstd, d, n0, gw, gw + (int64_t)net->nc * b.c, part, urn_net::SUM_SLOTS, net->st));
            net->check(urn_bn_bwd_apply_sums(b.x, d, nullptr, 0, n0, b.c, net->params + b.w, b.mean, b.invstd, part, urn_net::SUM_SLOTS,
                                             net->grads + b.w, net->grads + b.b, dx, net->st));
        }
        d = dx;
    } else {
        if (net->live()) {
            net->check(hipMemsetAsync(d, 0, (size_t)n0 * net->m * 4, net->st) == hipSuccess ? URN_OK : URN_EHIP);
            net->check(urn_rows_scatter_add(d_rows, net->geo.row2site, net->geo.n_rows, net->m, d, net->st));
        }
        d = net->bn_bwd(net->bn_out, d, n0);
        if (net->sums_mode()) net->sums_begin();
    }
    d = net->fused ? net->u_b(net->u, d, 0) : net->u_bwd(net->u, d, 0);
    net->conv_bwd(net->stem, d, net->geo.nbr[0], net->geo.nbr[0], 1, n0, n0, false);
    // join: the caller's stream continues only after every weight gradient has landed
    if (net->live()) net->dw_flush();
    if (net->side_used && net->live()) {
        hipEvent_t e = net->events[net->ev_next++ % net->events.size()];
        net->check(hipEventRecord(e, net->side) == hipSuccess && hipStreamWaitEvent(net->st, e, 0) == hipSuccess ? URN_OK : URN_EHIP);
        net->side_used = false;
    }
    if (net->side2_used && net->live()) {
        hipEvent_t e = net->events[net->ev_next++ % net->events.size()];
        net->check(hipEventRecord(e, net->side2) == hipSuccess && hipStreamWaitEvent(net->st, e, 0) == hipSuccess ? URN_OK : URN_EHIP);
        net->side2_used = false;
    }
}

// Workspace needed by one forward (+ backward when with_backward) for the given level sizes.
extern "C" int64_t urn_net_workspace_bytes(urn_net *net, int num_levels, const int64_t *n, int64_t n_rows, int with_backward)
{
    if (!net || !n || num_levels != net->L) return -1;
    if (set_geo(net, num_levels, 0, n, nullptr, nullptr, nullptr, nullptr, n_rows)) return -1;
    net->arena.reset(nullptr, 0, true);
    net->rc = URN_OK;
    run_forward(net, nullptr);
    net->arena.f32(n_rows * net->m);
    if (with_backward) run_backward(net, nullptr);
    return (int64_t)net->arena.peak + 4096;
}

// Forward of the trunk: site features (n0, 1) -> rows (n_rows, m) in input-row order (OutputLayer applied).
extern "C" int urn_net_forward(urn_net *net, int num_levels, int64_t ld, const int64_t *n, const void *const *nbr,
                               const void *const *chd, const void *const *up, const int32_t *row2site, int64_t n_rows,
                               const float *params, float *running, const float *site_feats, void *ws, int64_t ws_bytes,
                               float *out_rows, int training, void *stream)
{
    URN_CHECK_ARG(net && params && site_feats && ws && out_rows && nbr && row2site, "null pointer");
    int r = set_geo(net, num_levels, ld, n, nbr, chd, up, row2site, n_rows);
    if (r) return r;
    net->arena.reset(ws, (size_t)ws_bytes, false);
    net->params = params; net->running = running; net->training = training; net->st = (hipStream_t)stream;
    net->rc = URN_OK; net->grads = nullptr;
    net->head_logits = out_rows;
    run_forward(net, site_feats);
    if (net->arena.overflow) { urn_set_error("urn_net_forward: workspace too small (%zu needed so far)", net->arena.peak); return URN_EINVAL; }
    if (net->rc == URN_OK && !net->head_live) net->check(urn_rows_gather(net->trunk_out, row2site, n_rows, net->m, out_rows, net->st));
    return net->rc;
}

// Side-stream probe, an explicit call because it SYNCHRONISES (hipStreamSynchronize on `stream` and on the candidate
// streams): the executor's forward / backward never do.  Times the fork -> concurrent kernels -> join pattern of the
// backward pass on a few candidate side streams against `stream` and keeps the fastest (HIP multiplexes streams onto a
// few hardware queues; a side stream that shares its queue with the caller's stream serialises behind it: 9.1 instead of
// 3.4 ms per step measured).  Call once per handle and caller stream, at initialisation; without it the handle keeps the
// side stream it was created with.
extern "C" int urn_net_probe(urn_net *net, void *stream)
{
    URN_CHECK_ARG(net, "null handle");
    if (!net->side || net->side_probed) return URN_OK;
    net->st = (hipStream_t)stream;
    net->pick_side();
    return URN_OK;
}

// Backward of the last forward on this net (same workspace, same stream): d_rows (n_rows, m) ->
// grads (flat, ACCUMULATED into; caller zeroes).
extern "C" int urn_net_backward_cb(urn_net *net, const float *d_rows, float *grads, void *stream, void (*bottom_done)(void *), void *user)
{
    URN_CHECK_ARG(net, "null handle");
    net->bottom_cb = bottom_done; net->bottom_user = user;
    const int r = urn_net_backward(net, d_rows, grads, stream);
    net->bottom_cb = nullptr; net->bottom_user = nullptr;
    return r;
}

extern "C" int urn_net_backward(urn_net *net, const float *d_rows, float *grads, void *stream)
{
    URN_CHECK_ARG(net && d_rows && grads && net->trunk_out, "null pointer or no forward recorded");
    net->grads = grads; net->st = (hipStream_t)stream;
    run_backward(net, d_rows);
    if (net->arena.overflow) { urn_set_error("urn_net_backward: workspace too small (%zu needed)", net->arena.peak); return URN_EINVAL; }
    return net->rc;
}
