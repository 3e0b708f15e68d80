"""Host side of the sparse path: geometry (integer phase) and autograd Functions that
call the HIP kernels through the C ABI.  PyTorch only owns memory and the stream.
"""
import ctypes
import os

import torch

from . import lib as _l


MAX_MULTI = 8      # URN_GEO_MAX_LEVELS / URN_RB_MAX_LEVELS of the library


_PINNED_RING, _PINNED_NEXT = [], 0


def _pinned_counts(nl):
    """a pinned int32 buffer for the level counts of one geometry, from a ring of 16 (a geometry reads its counts back
    before 16 more are built; allocating pinned memory per geometry cost milliseconds whenever the host allocator had no
    free block -- seen as a 6 % slower 30-step bench than the 100-step one)"""
    global _PINNED_NEXT
    if not _PINNED_RING:
        _PINNED_RING.extend(torch.empty(16, dtype=torch.int32, pin_memory=True) for _ in range(16))
    b = _PINNED_RING[_PINNED_NEXT % 16]
    _PINNED_NEXT += 1
    assert nl <= 16
    return b[:nl]


class SparseGeometry:
    """Active sites, hash tables and gather tables of every level of one forward.

    Built entirely on the device (coordinate hash + rulebooks, no host work); one
    device->host copy of the per-level site counts ends the integer phase so that the
    float phase can size its tensors.  Restates the metadata that scn.InputLayer /
    SubmanifoldConvolution / Convolution build on the host in the reference
    (uresnet/models/uresnet_sparse.py:20-22).
    """

    def __init__(self, coords, spatial_size, num_levels=1, defer_sync=False, per_level=False, counts_hint=None):
        """defer_sync=True: everything is enqueued but the level counts are not read back yet -- the caller does the
        rest of its host-side preparation (measured: host time after this synchronisation is exposed one to one in
        the step, host time before it is hidden behind the previous step's kernels) and then calls sync().
        per_level=True: build the pyramid level by level (urn_sites_build + urn_level_down_tables per level) instead of
        all levels at once from the input rows (urn_sites_build_levels); same results, kept for the per-call C ABI."""
        _l.require_gpu(coords)
        L = _l.load()
        assert coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4
        coords = coords.contiguous()
        dev = coords.device
        self.device = dev
        self.spatial = int(spatial_size)
        self.num_levels = int(num_levels)
        N = coords.shape[0]
        self.n_rows = N
        cap = max(N, 1)
        self.cap = cap
        st = _l.stream()
        hcap = L.urn_hash_capacity(cap)
        hbytes = L.urn_hash_bytes(hcap)
        self.hcap = hcap
        # one arena for all level hashes, cleared with a single memset
        self.hash = torch.empty(self.num_levels * hbytes, dtype=torch.uint8, device=dev)
        _l.check(L.urn_hash_clear(self.hash.data_ptr(), self.hash.numel(), st), 'hash_clear')
        self._hptr = [self.hash.data_ptr() + l * hbytes for l in range(self.num_levels)]
        nl = self.num_levels
        PA, IA = ctypes.c_void_p * nl, ctypes.c_int * nl
        # counts: [n_l for each level] + [rules_l for each level]
        self.counts = torch.zeros(2 * nl, dtype=torch.int32, device=dev)
        cptr = self.counts.data_ptr()
        self.row2site = torch.empty(cap, dtype=torch.int32, device=dev)
        self._coords_all = torch.empty((nl, cap, 4), dtype=torch.int32, device=dev)
        self.coords = [self._coords_all[l] for l in range(nl)]
        self.ld = cap                      # leading dimension of every gather table
        self.nbr, self.parent, self.off, self.chd, self.up = [], [], [], [], []
        # strided tables are filled with -1 by one launch
        if nl > 1:
            self._strided = torch.empty((nl - 1, 2, 8, cap), dtype=torch.int32, device=dev)
            _l.check(L.urn_fill_i32(self._strided.data_ptr(), self._strided.numel(), -1, st), 'fill')
            self._links = torch.empty((nl - 1, 2, cap), dtype=torch.int32, device=dev)
            self.parent = [self._links[l, 0] for l in range(nl - 1)]
            self.off = [self._links[l, 1] for l in range(nl - 1)]
            self.chd = [self._strided[l, 0] for l in range(nl - 1)]
            self.up = [self._strided[l, 1] for l in range(nl - 1)]
        self._nbr_all = torch.empty((nl, 27, cap), dtype=torch.int32, device=dev)
        self.nbr = [self._nbr_all[l] for l in range(nl)]
        spatials = []
        sp = self.spatial
        for l in range(nl):
            spatials.append(sp)
            sp = (sp + 1) // 2
        if not per_level and nl <= MAX_MULTI:
            # every level straight from the input rows: four launches for the whole pyramid
            sbytes = L.urn_levels_scratch_bytes(cap, nl)
            scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev)
            pad = [None]
            _l.check(L.urn_sites_build_levels(
                coords.data_ptr(), N, self.spatial, nl, PA(*self._hptr), hcap, scratch.data_ptr(), sbytes,
                self.row2site.data_ptr(), PA(*[c.data_ptr() for c in self.coords]), cptr,
                PA(*([t.data_ptr() for t in self.parent] + pad)), PA(*([t.data_ptr() for t in self.off] + pad)),
                PA(*([t.data_ptr() for t in self.chd] + pad)), PA(*([t.data_ptr() for t in self.up] + pad)), cap, st),
                'sites_build_levels')
        else:
            # level by level (the per-call C ABI: urn_sites_build, then urn_level_down_tables per level)
            sbytes = L.urn_unique_scratch_bytes(cap)
            scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev)
            _l.check(L.urn_sites_build(coords.data_ptr(), N, self.spatial, self._hptr[0], hcap, scratch.data_ptr(),
                                       sbytes, self.row2site.data_ptr(), self.coords[0].data_ptr(), cptr, st),
                     'sites_build')
            for l in range(nl - 1):
                _l.check(L.urn_level_down_tables(self.coords[l].data_ptr(), cptr + 4 * l, cap, self._hptr[l + 1], hcap,
                                                 scratch.data_ptr(), sbytes, self.coords[l + 1].data_ptr(),
                                                 self.parent[l].data_ptr(), self.off[l].data_ptr(), cptr + 4 * (l + 1),
                                                 self.chd[l].data_ptr(), cap, self.up[l].data_ptr(), cap, st),
                         'level_down_tables')
        # The site counts are final here: their copy to (pinned) host memory is enqueued NOW, in front of the rulebook and
        # pair-list launches, so that when sync() returns the queue still holds those (~50 us of work on the cfg3 event)
        # and the host's first float-phase launches are not exposed (kernel trace before: 35 us idle at the blocking
        # copy + 34 us until the next launch arrived, per step).
        self._hint = None if counts_hint is None else [int(v) for v in counts_hint]
        if self._hint is None:
            self._n_host = _pinned_counts(nl)
            self._n_gen = _PINNED_NEXT                 # (the ring slot is ours until 16 more geometries have been built)
            self._n_host.copy_(self.counts[:nl], non_blocking=True)
            self._n_event = torch.cuda.Event()
            self._n_event.record(torch.cuda.current_stream(dev))
        # the 27-offset tables of every level in one launch (the multi-level entry points take up to MAX_MULTI levels)
        if nl <= MAX_MULTI:
            _l.check(L.urn_rulebook_subm_multi(nl, PA(*[c.data_ptr() for c in self.coords]), PA(*[cptr + 4 * l for l in range(nl)]),
                                               cap, IA(*spatials), PA(*self._hptr), hcap, PA(*[t.data_ptr() for t in self.nbr]),
                                               cap, st), 'rulebook_subm_multi')
        else:
            for l in range(nl):
                _l.check(L.urn_rulebook_subm(self.coords[l].data_ptr(), cptr + 4 * l, cap, spatials[l], self._hptr[l], hcap,
                                             self.nbr[l].data_ptr(), cap, None, st), 'rulebook_subm')
        self._scratch = scratch
        self._rules = None
        self.n = None
        self._build_pairs(L, st)
        if not defer_sync:
            self.sync()

    # Tile sizes of the compacted rule lists, by level: 64 output rows; 128 for the fine<-parent table (one rule per row
    # spread over 8 offsets: a 64-row tile would half-fill its blocks).  32-row tiles for the small deep levels were
    # measured slower (cfg3 step 3.25 vs 3.24 ms; level 3, 64 -> 64 alone 33 vs 26 us).  URN_PAIRS_TILES="nbr;chd;up"
    # (comma-separated per level) overrides.
    PAIRS_TILES = {'nbr': [64], 'chd': [64], 'up': [128]}

    @classmethod
    def pairs_tile(cls, kind, level):
        env = os.environ.get('URN_PAIRS_TILES')
        tiles = cls.PAIRS_TILES
        if env:
            tiles = dict(zip(('nbr', 'chd', 'up'), [[int(v) for v in part.split(',')] for part in env.split(';')]))
        lst = tiles[kind]
        return lst[min(level, len(lst) - 1)]

    def _build_pairs(self, L, st):
        """Compacted rule lists of every gather table (one launch): what the MFMA kernels walk.  Row counts stay on the
        device.  self.pairs[kind][level] = (int32 tensor, tile) for kind in nbr / chd / up."""
        nl, cap, dev = self.num_levels, self.cap, self.device
        cptr = self.counts.data_ptr()
        jobs = [('nbr', l, self.nbr[l], 27, cptr + 4 * l) for l in range(nl)]
        jobs += [('chd', l, self.chd[l], 8, cptr + 4 * (l + 1)) for l in range(nl - 1)]
        jobs += [('up', l, self.up[l], 8, cptr + 4 * l) for l in range(nl - 1)]
        self.pairs = {'nbr': [None] * nl, 'chd': [None] * max(nl - 1, 0), 'up': [None] * max(nl - 1, 0)}
        tiles = [self.pairs_tile(kind, l) for kind, l, _, _, _ in jobs]
        sizes = [L.urn_pairs_bytes(cap, K, T) // 4 for (_, _, _, K, _), T in zip(jobs, tiles)]
        self._pairs_all = torch.empty(sum(sizes), dtype=torch.int32, device=dev)
        outs, off = [], 0
        for (kind, l, _, _, _), sz, T in zip(jobs, sizes, tiles):
            t = self._pairs_all[off:off + sz]
            self.pairs[kind][l] = (t, T)
            outs.append(t.data_ptr()); off += sz
        n = len(jobs)
        PA, I64A, IA = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
        _l.check(L.urn_pairs_build(n, PA(*[j[2].data_ptr() for j in jobs]), I64A(*([cap] * n)), IA(*[j[3] for j in jobs]),
                                   PA(*[j[4] for j in jobs]), I64A(*([cap] * n)), IA(*tiles),
                                   PA(*outs), st), 'pairs_build')

    def sync(self):
        """the one host synchronisation of the integer phase: per-level site counts"""
        if self.n is None and self._hint is not None:
            # counts_hint: the caller knows the level counts (a captured step replays the geometry build of the event it was
            # captured on): no read-back, nothing that synchronises
            self.n = list(self._hint)
            return self
        if self.n is None:
            if os.environ.get('URN_LATE_COUNTS'):     # A/B: the blocking copy at the end of the integer phase
                self.n = self.counts.cpu().tolist()[:self.num_levels]
                return self
            if _PINNED_NEXT - self._n_gen >= 16:     # our ring slot was handed out again: read the device copy instead
                self.n = self.counts.cpu().tolist()[:self.num_levels]
                return self
            self._n_event.synchronize()
            self.n = self._n_host.tolist()
        return self

    @property
    def rules(self):
        """Number of (offset, in, out) rules per level (only measurement and tests need it)."""
        if self._rules is None:
            self._rules = [int((self.nbr[l][:, :self.n[l]] >= 0).sum().item()) for l in range(self.num_levels)]
        return self._rules

    # canonical exports for parity tests (device -> host)
    def export_nbr(self, level):
        return self.nbr[level][:, :self.n[level]].cpu().numpy()

    def export_coords(self, level):
        return self.coords[level][:self.n[level]].cpu().numpy()


def input_features(geo, feats):
    """InputLayer mode 3: site features = sum of the rows that share a site.  With an unsynchronised geometry the
    site count stays on the device: the result then has geo.cap rows of which the first n[0] are sites."""
    L = _l.load()
    feats = feats.contiguous().float()
    nf = feats.shape[1]
    if geo.n is None:
        n0, n_dev = geo.cap, geo.counts.data_ptr()
    else:
        n0, n_dev = geo.n[0], None
    out = torch.empty((n0, nf), dtype=torch.float32, device=feats.device)
    acc = torch.empty(max(n0 * nf, 1), dtype=torch.float64, device=feats.device)
    _l.check(L.urn_input_features(feats.data_ptr(), geo.row2site.data_ptr(), geo.n_rows, nf, n_dev, n0,
                                  acc.data_ptr(), out.data_ptr(), _l.stream()), 'input_features')
    return out


IDENT_PAIRS = (None, 64)     # the identity table of a 1x1 convolution needs no list
# Weight gradients without fp32 atomics (bitwise reproducible): False = the dense-table kernel adding its partial tiles with
# atomics (last bits depend on the arrival order); 'slabs' = the same kernel storing the partials of every (row chunk,
# offset, channel tile) workgroup, added by a second launch in a fixed order; 'pairs' = the two-stage kernel on the
# compacted rule lists (1.5-2.5x slower per launch).
DETERMINISTIC_DW = False


def set_deterministic_dw(on, kind='slabs'):
    """Both routes: the per-layer autograd path and the executor (urn_set_option "dw_2stage" / "dw_pairs")."""
    global DETERMINISTIC_DW
    assert kind in ('slabs', 'pairs')
    DETERMINISTIC_DW = kind if on else False
    L = _l.load()
    L.urn_set_option(b'dw_pairs', int(bool(on) and kind == 'pairs'))
    L.urn_set_option(b'dw_2stage', int(bool(on) and kind == 'slabs'))


def _gconv(x, wt, tbl, ld, K, flip, n_out, cin, cout, res=None, pairs=None):
    """pairs = (list tensor or None, tile): run on the compacted rule list of `tbl` (see SparseGeometry.pairs)"""
    L = _l.load()
    y = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    if pairs is None:
        _l.check(L.urn_gconv_fwd(_l.ptr(x), _l.ptr(wt), tbl.data_ptr(), ld, K, flip, n_out, cin, cout, _l.ptr(res),
                                 y.data_ptr(), _l.stream()), 'gconv_fwd')
        return y
    a = _l.GConvArgs()
    a.x = _l.ptr(x); a.wt = _l.ptr(wt); a.tbl = tbl.data_ptr(); a.ld = ld; a.K = K; a.flip = flip; a.n_out = n_out
    a.cin = cin; a.cout = cout; a.res = _l.ptr(res); a.y = y.data_ptr()
    a.pairs = None if pairs[0] is None else pairs[0].data_ptr()
    a.pairs_tile = pairs[1]
    wf = None
    if cin % 16 == 0 and cout % 16 == 0 and K > 1:
        # the weights once more in MFMA-fragment order, as the executor hands them to the pair-list kernel (same kernel
        # choice and the same bits on both routes)
        wf = torch.empty_like(wt)
        prec = _l.precision()
        if prec:        # reduced precision: the pair-list kernel reads 16-bit fragments (urn_gconv_args.wt_frag_prec)
            _l.check(L.urn_weight_fragments16(_l.ptr(wt), K, cout, cin, prec, wf.data_ptr(), _l.stream()), 'weight_fragments16')
        else:
            _l.check(L.urn_weight_fragments(_l.ptr(wt), K, cout, cin, wf.data_ptr(), _l.stream()), 'weight_fragments')
        a.wt_frag = wf.data_ptr()
        a.wt_frag_prec = prec
    _l.check(L.urn_gconv_fwd_ex(ctypes.byref(a), None, _l.stream()), 'gconv_fwd_ex')
    return y


def _transpose_w(w):
    L = _l.load()
    K, a, b = w.shape
    wt = torch.empty((K, b, a), dtype=torch.float32, device=w.device)
    _l.check(L.urn_transpose_w(_l.ptr(w), K, a, b, wt.data_ptr(), _l.stream()), 'transpose_w')
    return wt


class GConvFunction(torch.autograd.Function):
    """y = gather-conv(x, W) (+ res).  tbl_f / tbl_b are the forward and the inverse
    gather tables ([K][ld]); flip_b mirrors the offset index in the backward table
    (submanifold: the inverse of nbr[o] is nbr[26-o])."""

    @staticmethod
    def forward(ctx, x, weight, res, tbl_f, tbl_b, flip_b, ld, n_out, n_in, pairs_f=None, pairs_b=None):
        _l.require_gpu(x)
        x = x.contiguous(); weight = weight.contiguous()
        K, cin, cout = weight.shape
        assert x.shape == (n_in, cin), (x.shape, n_in, cin)
        wt = _transpose_w(weight)
        y = _gconv(x, wt, tbl_f, ld, K, 0, n_out, cin, cout, None if res is None else res.contiguous(), pairs_f)
        ctx.save_for_backward(x, weight)
        ctx.meta = (tbl_f, tbl_b, flip_b, ld, n_out, n_in, res is not None)
        ctx.pairs_f, ctx.pairs_b = pairs_f, pairs_b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        tbl_f, tbl_b, flip_b, ld, n_out, n_in, has_res = ctx.meta
        K, cin, cout = weight.shape
        dy = dy.contiguous()
        L = _l.load()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # dx[i] = sum_o dy[inv[o][i]] @ W[o]^T : same kernel, W itself is the
            # "(K, cout_eff=cin, cin_eff=cout)" transposed operand
            dx = _gconv(dy, weight, tbl_b, ld, K, flip_b, n_in, cout, cin, None, ctx.pairs_b)
        if ctx.needs_input_grad[1]:
            dw = torch.zeros_like(weight)
            pf = ctx.pairs_f
            if DETERMINISTIC_DW == 'slabs' and cin % 16 == 0 and cout % 16 == 0:
                sb = L.urn_gconv_dw_2stage_scratch_bytes(K, n_out, cin, cout)
                scratch = torch.empty(sb, dtype=torch.uint8, device=x.device)
                _l.check(L.urn_gconv_bwd_dw_2stage(_l.ptr(x), None, None, _l.ptr(dy), cout, tbl_f.data_ptr(), ld, K, n_out, cin, cout,
                                                   dw.data_ptr(), scratch.data_ptr(), sb, _l.stream()), 'gconv_bwd_dw_2stage')
            elif DETERMINISTIC_DW == 'pairs' and pf is not None and cin % 16 == 0 and cout % 16 == 0:
                # two-stage sum over the compacted rule list: no atomics, bitwise reproducible
                sb = L.urn_gconv_dw_pairs_scratch_bytes(n_out, pf[1], K, cin, cout)
                scratch = torch.empty(sb, dtype=torch.uint8, device=x.device)
                _l.check(L.urn_gconv_bwd_dw_pairs(_l.ptr(x), 0, None, None, _l.ptr(dy), 0,
                                                  None if pf[0] is None else pf[0].data_ptr(), pf[1], K, n_out, cin, cout,
                                                  dw.data_ptr(), scratch.data_ptr(), sb, _l.stream()), 'gconv_bwd_dw_pairs')
            else:
                _l.check(L.urn_gconv_bwd_dw(_l.ptr(x), _l.ptr(dy), tbl_f.data_ptr(), ld, K, n_out, cin, cout,
                                            dw.data_ptr(), _l.stream()), 'gconv_bwd_dw')
        dres = dy if (has_res and ctx.needs_input_grad[2]) else None
        return dx, dw, dres, None, None, None, None, None, None, None, None


class BNReLUFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu, training):
        _l.require_gpu(x)
        L = _l.load()
        x = x.contiguous()
        n, c = x.shape
        y = torch.empty_like(x)
        st = _l.stream()
        if training:
            mean = torch.empty(c, dtype=torch.float32, device=x.device)
            invstd = torch.empty(c, dtype=torch.float32, device=x.device)
            scratch = torch.empty(L.urn_bn_scratch_bytes(c), dtype=torch.uint8, device=x.device)
            _l.check(L.urn_bn_relu_fwd(_l.ptr(x), n, c, _l.ptr(gamma), _l.ptr(beta), float(eps), int(relu),
                                       y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _l.ptr(running_mean),
                                       _l.ptr(running_var), float(momentum), scratch.data_ptr(), st), 'bn_fwd')
        else:
            mean = running_mean
            invstd = torch.rsqrt(running_var + eps)
            _l.check(L.urn_bn_relu_apply(_l.ptr(x), n, c, _l.ptr(gamma), _l.ptr(beta), _l.ptr(mean),
                                         _l.ptr(invstd), int(relu), y.data_ptr(), st), 'bn_apply')
        ctx.save_for_backward(x, y, gamma, mean, invstd)
        ctx.relu = int(relu)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, invstd = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError('BatchNormReLU backward in eval mode is not supported')
        L = _l.load()
        n, c = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg = torch.zeros(c, dtype=torch.float32, device=x.device)   # the kernel accumulates
        db = torch.zeros(c, dtype=torch.float32, device=x.device)
        scratch = torch.empty(L.urn_bn_scratch_bytes(c), dtype=torch.uint8, device=x.device)
        _l.check(L.urn_bn_relu_bwd(_l.ptr(x), _l.ptr(y), _l.ptr(dy), n, c, _l.ptr(gamma), _l.ptr(mean),
                                   _l.ptr(invstd), ctx.relu, dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                   scratch.data_ptr(), _l.stream()), 'bn_bwd')
        return dx, dg, db, None, None, None, None, None, None


class RowsGatherFunction(torch.autograd.Function):
    """OutputLayer: y[i] = x[idx[i]]"""

    @staticmethod
    def forward(ctx, x, idx, n_rows):
        _l.require_gpu(x)
        L = _l.load()
        x = x.contiguous()
        c = x.shape[1]
        y = torch.empty((n_rows, c), dtype=torch.float32, device=x.device)
        _l.check(L.urn_rows_gather(_l.ptr(x), idx.data_ptr(), n_rows, c, y.data_ptr(), _l.stream()), 'rows_gather')
        ctx.idx = idx
        ctx.shape = (x.shape[0], c, n_rows)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _l.load()
        na, c, n_rows = ctx.shape
        dy = dy.contiguous()
        dx = torch.zeros((na, c), dtype=torch.float32, device=dy.device)
        _l.check(L.urn_rows_scatter_add(_l.ptr(dy), ctx.idx.data_ptr(), n_rows, c, dx.data_ptr(), _l.stream()),
                 'rows_scatter_add')
        return dx, None, None


class HeadFunction(torch.autograd.Function):
    """logits = rows @ W^T + b on the HIP head kernel (identity row map: the trunk already applied OutputLayer)."""

    @staticmethod
    def forward(ctx, rows, weight, bias):
        _l.require_gpu(rows)
        L = _l.load()
        rows = rows.contiguous(); weight = weight.contiguous()
        n, m = rows.shape
        nc = weight.shape[0]
        logits = torch.empty((n, nc), dtype=torch.float32, device=rows.device)
        _l.check(L.urn_head_fwd(rows.data_ptr(), None, n, m, nc, weight.data_ptr(), _l.ptr(bias), logits.data_ptr(),
                                _l.stream()), 'head_fwd')
        ctx.save_for_backward(rows, weight)
        ctx.has_bias = bias is not None
        return logits

    @staticmethod
    def backward(ctx, dl):
        rows, weight = ctx.saved_tensors
        L = _l.load()
        n, m = rows.shape
        nc = weight.shape[0]
        dl = dl.contiguous()
        dx = torch.empty_like(rows)
        acc = torch.zeros(nc * m + nc, dtype=torch.float32, device=rows.device)   # the kernel accumulates: one fill for both
        dW = acc[:nc * m].view(nc, m); db = acc[nc * m:]
        _l.check(L.urn_head_bwd(dl.data_ptr(), rows.data_ptr(), None, n, m, nc, weight.data_ptr(), dx.data_ptr(),
                                dW.data_ptr(), db.data_ptr(), _l.stream()), 'head_bwd')
        return dx, dW, (db if ctx.has_bias else None)


class SegmentationCEFunction(torch.autograd.Function):
    """sum over events of the mean (weighted) voxel cross-entropy; also returns the summed per-event accuracy.
    One pass over the rows, per-event sums on the device, no host synchronisation."""

    @staticmethod
    def forward(ctx, logits, data, label, weight):
        _l.require_gpu(logits)
        L = _l.load()
        logits = logits.contiguous()
        n, nc = logits.shape
        assert data.is_contiguous() and data.dtype == torch.float32 and label.dtype == torch.float32
        label = label.contiguous()
        w = None if weight is None else weight.contiguous().float()
        stride = data.shape[1]
        bid_ptr = data.data_ptr() + 4 * (stride - 2)            # column -2 = batch id (reference :57)
        row_lse = torch.empty(n, dtype=torch.float32, device=logits.device)
        ev = torch.empty(L.urn_ce_scratch_bytes() // 8, dtype=torch.float64, device=logits.device)
        out = torch.empty(2, dtype=torch.float32, device=logits.device)
        _l.check(L.urn_ce_fwd(logits.data_ptr(), label.data_ptr(), bid_ptr, stride, _l.ptr(w), n, nc, row_lse.data_ptr(),
                              ev.data_ptr(), out.data_ptr(), _l.stream()), 'ce_fwd')
        ctx.save_for_backward(logits, data, label, row_lse, ev)
        ctx.w = w
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)      # (no zero tensor for the non-differentiable [loss, accuracy] pair: one fill launch per step)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, gloss, _gout):
        if gloss is None:
            return None, None, None, None
        logits, data, label, row_lse, ev = ctx.saved_tensors
        L = _l.load()
        n, nc = logits.shape
        stride = data.shape[1]
        g = gloss.contiguous().float().reshape(1)
        dl = torch.empty_like(logits)
        _l.check(L.urn_ce_bwd(logits.data_ptr(), label.data_ptr(), data.data_ptr() + 4 * (stride - 2), stride,
                              _l.ptr(ctx.w), row_lse.data_ptr(), ev.data_ptr(), g.data_ptr(), n, nc, dl.data_ptr(),
                              _l.stream()), 'ce_bwd')
        return dl, None, None, None
