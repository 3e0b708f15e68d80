"""Dense convolutions of the dense U-ResNet on the implicit-GEMM kernel (csrc/urn_dense.hip), through the C ABI.

Replaces, for GPU tensors, what the reference asks of torch/cuDNN (reference uresnet/models/uresnet_dense.py):
  F.pad(mode='replicate') + nn.Conv{2,3}d k{1,3} s{1,2}   (:38-67, 75-80, 128-134, 179-197)   -> conv()
  nn.ConvTranspose{2,3}d k3 s2 p1 op1                      (:165-172)                          -> conv_transpose()
Activations are channels-last row matrices (B * Z * Y * X rows, C columns, fp32); nothing is padded or tabulated: the
replicate clamp lives in the kernel's addressing, the four gather forms (forward, input gradient, transposed forward,
its input gradient) are geometry descriptions of ONE kernel (urn_dense_geom), and the weight gradient is its own kernel.
"""
import ctypes

import torch

from . import lib as _l

PRECISION = 0        # 0 fp32 operands, 1 bf16 operands (fp32 accumulate); set_precision()


def set_precision(name):
    global PRECISION
    PRECISION = {'fp32': 0, 'bf16': 1}[name]


def _dims3(spatial):
    """spatial dims (2 or 3 of them) -> [Z, Y, X] with Z = 1 for 2-D"""
    return [1] * (3 - len(spatial)) + list(spatial)


def _geom(In, Out, Sub, p, os, s, taps, kdim, mode):
    """taps: per dim list of (e, w)"""
    g = _l.DenseGeom()
    for d in range(3):
        g.In[d], g.Out[d], g.Sub[d], g.p[d], g.os[d], g.s[d] = In[d], Out[d], Sub[d], p[d], os[d], s[d]
        g.nt[d] = len(taps[d]); g.kdim[d] = kdim[d]
        for j, (e, w) in enumerate(taps[d]):
            g.e[d][j] = e; g.wi[d][j] = w
    g.mode = mode
    return g


_SCRATCH = {}


STAT_SLOTS = 256     # rows of a statistics slab (urn_dense_conv: stats); the 128^3 levels launch 32768 workgroups


class _ZeroPool(object):
    """fp64 accumulation slabs (statistics of ~60 convolutions, BatchNorm-backward sums) come out of ONE buffer that is
    zeroed by one memset when the network's forward pass begins (pool_begin) instead of ~100 small fills per step.  Slabs are
    scratch: consumed on the same stream before anything reuses the buffer, never handed to autograd as results."""

    def __init__(self):
        self.buf, self.off, self.want = None, 0, 8 << 20

    def begin(self, device):
        if self.buf is None or self.buf.device != device or self.buf.numel() * 8 < self.want:
            self.buf = torch.empty(self.want // 8, dtype=torch.float64, device=device)
        self.buf.zero_()
        self.off = 0

    def take(self, shape, device):
        n = 1
        for d in shape:
            n *= d
        if self.buf is None or self.buf.device != device or self.off + n > self.buf.numel():
            if self.buf is not None and self.buf.device == device:
                self.want = max(self.want, 2 * (self.off + n) * 8)      # grow at the next pool_begin
                self.off += n
            return torch.zeros(shape, dtype=torch.float64, device=device)
        t = self.buf[self.off:self.off + n].view(shape)
        self.off += (n + 31) & ~31
        return t


_POOL = _ZeroPool()


def pool_begin(device):
    _POOL.begin(device)
    _ZGRADS.begin(device)


def zeros_f64(shape, device):
    return _POOL.take(tuple(shape), device)


class _ZeroGrads(object):
    """The (exactly zero) bias gradient of a convolution that a BatchNorm follows -- ~60 per cfg2 step, a fill launch each
    before: fresh views of ONE float buffer that nothing in this package ever writes except its own memset when a forward
    pass begins (begin).  autograd's AccumulateGrad takes such a view over as .grad without a copy (or adds it to an older
    view of the same zeros); the regions of a pass are distinct.  Deliberately NOT the statistics pool above: a .grad that
    outlives the step (zero_grad(set_to_none=False), gradient accumulation over several passes) must not alias memory the
    next forward pass accumulates sums in."""

    def __init__(self):
        self.buf, self.off = None, 0

    def begin(self, device):
        if self.buf is not None and self.buf.device == device:
            self.buf.zero_()          # (someone may have written into a .grad in place)
        self.off = 0

    def take(self, n, device):
        if self.buf is None or self.buf.device != device:
            self.buf = torch.zeros(1 << 16, dtype=torch.float32, device=device)
            self.off = 0
        if self.off + n > self.buf.numel():
            return torch.zeros(n, dtype=torch.float32, device=device)
        t = self.buf[self.off:self.off + n]
        self.off += (n + 3) & ~3
        return t


_ZGRADS = _ZeroGrads()


def zeros_f32(n, device):
    return _ZGRADS.take(n, device)


def new_stats(cout_p, device):
    """zeroed [STAT_SLOTS][2][cout_p] fp64 slab for the column statistics a convolution's epilogue accumulates"""
    return zeros_f64((STAT_SLOTS, 2, cout_p), device)


def stats_ok(c):
    """channel counts the fused statistics / row passes take (a thread keeps its 4 channels: 256 % (c / 4) == 0)"""
    cp = c + (-c) % 16
    return c % 4 == 0 and 256 % (c // 4) == 0 and 256 % (cp // 4) == 0


def _launch(x, ldx, cin, wt, bias, y, ldy, cout, B, g, stats=None, xf=None):
    L = _l.load()
    sb = L.urn_dense_conv_scratch_bytes(cout, B, ctypes.byref(g))
    key = (x.device, sb > 256)
    scratch = _SCRATCH.get(key)
    if scratch is None or scratch.numel() < sb:
        # split-contraction slabs of the small launches (same stream: reuse is ordered); the big launches never split
        scratch = _SCRATCH[key] = torch.empty(max(sb, 256), dtype=torch.uint8, device=x.device)
    _l.check(L.urn_dense_conv(x.data_ptr(), ldx, cin, wt.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(), ldy,
                              cout, B, ctypes.byref(g), PRECISION, None if stats is None else stats.data_ptr(),
                              0 if stats is None else stats.shape[0], None if xf is None else xf[0].data_ptr(),
                              None if xf is None else xf[1].data_ptr(), scratch.data_ptr(), scratch.numel(), _l.stream()), 'dense_conv')


def _colsum(rows, c):
    """column sums of the first c columns of a row matrix (bias gradient): the BatchNorm statistics kernel (fp64 partials
    per workgroup, summed here) instead of a torch reduction over 2M x 16 elements (1.6 ms at 128^3)"""
    L = _l.load()
    rows = rows.contiguous()
    n, cp = rows.shape
    part = torch.empty(L.urn_bn_scratch_bytes(cp) // 8, dtype=torch.float64, device=rows.device)
    n_part = ctypes.c_int(0)
    _l.check(L.urn_bn_stats_partial(rows.data_ptr(), n, cp, part.data_ptr(), ctypes.byref(n_part), _l.stream()), 'bn_stats_partial')
    return part[:n_part.value * 2 * cp].reshape(n_part.value, 2, cp)[:, 0, :c].sum(0).float()


# ---- weight layouts of the whole model in one launch -------------------------------------------------------------------
# urn_dense_conv reads [tap][cout_p][cin_p] (forward) and [tap][cin_p][cout_p] (input gradient); torch keeps (cout, cin, taps)
# / (cin, cout, taps).  prepare_weights() -- called by the model at the head of a forward pass -- writes both layouts of every
# convolution with ONE launch into persistent buffers (a permute + contiguous per convolution and pass otherwise: ~120
# launches per cfg2 step); the autograd Functions below look their operand up by the parameter's identity and fall back to
# the per-call copies for weights nobody prepared (direct calls in tests).
class _WeightLayouts(object):
    def __init__(self):
        self.key, self.entries, self.descs, self.buf, self.valid = None, {}, None, None, False

    def prepare(self, convs):
        """convs: list of (weight parameter, transposed) in any order"""
        import numpy as np
        key = tuple((id(w), w.data_ptr(), tuple(w.shape), bool(t)) for w, t in convs)
        if key != self.key:
            total, plan = 0, []
            for w, t in convs:
                d0, d1 = w.shape[0], w.shape[1]
                taps = w.numel() // (d0 * d1)
                cin, cout = (d0, d1) if t else (d1, d0)
                cin_p, cout_p = cin + (-cin) % 16, cout + (-cout) % 16
                n = taps * cin_p * cout_p
                plan.append((w, t, taps, cin, cout, cin_p, cout_p, total, total + n))
                total += 2 * n
            self.buf = torch.empty(total, dtype=torch.float32, device=convs[0][0].device)
            base = self.buf.data_ptr()
            rec = np.zeros((len(plan), 11), np.int64)
            self.entries = {}
            for i, (w, t, taps, cin, cout, cin_p, cout_p, o_f, o_b) in enumerate(plan):
                n = taps * cin_p * cout_p
                rec[i] = [w.data_ptr(), base + 4 * o_f, base + 4 * o_b, taps, int(t), cin, cout, cin_p, cout_p, 0, 0]
                self.entries[id(w)] = (w.data_ptr(), self.buf[o_f:o_f + n].view(taps, cout_p, cin_p), self.buf[o_b:o_b + n].view(taps, cin_p, cout_p))
            self.descs = rec
            self.key = key
        _l.check(_l.load().urn_dense_weight_layouts(len(self.descs), self.descs.ctypes.data, _l.stream()), 'dense_weight_layouts')
        self.valid = True

    def get(self, weight):
        e = self.entries.get(id(weight)) if self.valid else None
        if e is None or e[0] != weight.data_ptr():
            return None
        return e[1], e[2]


_WL = _WeightLayouts()


def prepare_weights(convs):
    if convs and convs[0][0].is_cuda:
        _WL.prepare(convs)


def invalidate_weights():
    """the prepared layouts describe the parameters as they were at prepare_weights(): callers that change a weight between a
    forward pass and another use of these Functions without a new forward (tests) drop them"""
    _WL.valid = False


_PAD_ROWS = {}


def _pad16_rows(t):
    """a row matrix with its channel count zero-padded to a multiple of 16, for use WITHIN the calling backward pass only (the
    gradient of the num_class-wide output layer: 2M x 5 -> 16 at 128^3): the rows are copied into the first columns of a cached
    buffer whose pad columns were zeroed once and are never written -- a strided 40 MB copy instead of a 134 MB cat + a fill +
    a contiguous copy.  The buffer is reused by the next call of the same shape (same stream: ordered)."""
    n, c = t.shape
    padn = (-c) % 16
    if not padn:
        return t.contiguous()
    key = (t.device, n, c + padn)
    buf = _PAD_ROWS.get(key)
    if buf is None:
        if len(_PAD_ROWS) >= 4:
            _PAD_ROWS.clear()
        buf = _PAD_ROWS[key] = torch.zeros((n, c + padn), dtype=t.dtype, device=t.device)
    buf[:, :c].copy_(t)
    return buf


def _pad16(t, dim):
    n = t.shape[dim]
    padn = (-n) % 16
    if not padn:
        return t
    shape = list(t.shape); shape[dim] = padn
    return torch.cat([t, t.new_zeros(shape)], dim=dim)


def conv_geoms(spatial, k, stride, pad_lo, pad_hi):
    """forward geometry of Conv k s on the replicate-padded volume, and the geometries of its input gradient on the
    PADDED volume (one per parity class for stride 2).  Returns (out_spatial3, fwd, [bwd...], padded3)."""
    nd = len(spatial)
    In = _dims3(spatial)
    real = [False] * (3 - nd) + [True] * nd
    kk = [k if r else 1 for r in real]
    ss = [stride if r else 1 for r in real]
    lo = [pad_lo if r else 0 for r in real]
    hi = [pad_hi if r else 0 for r in real]
    Pd = [In[d] + lo[d] + hi[d] for d in range(3)]
    Out = [(Pd[d] - kk[d]) // ss[d] + 1 for d in range(3)]
    fwd = _geom(In, Out, Out, [0] * 3, [1] * 3, ss, [[(j - lo[d], j) for j in range(kk[d])] for d in range(3)], kk, 0)
    # input gradient: dxp[pp] = sum_t dy[(pp - t) / s] W[t]^T where divisible and in range (zero mode)
    bwd = []
    if stride == 1:
        bwd.append(_geom(Out, Pd, Pd, [0] * 3, [1] * 3, [1] * 3, [[(-j, j) for j in range(kk[d])] for d in range(3)], kk, 1))
    else:
        def classes(d):
            if not real[d]:
                return [(0, 1, [(0, 0)])]                      # (parity origin, output step, taps)
            out = []
            for c in range(2):
                # pp = 2u + c, o = (pp - t) / 2 = u + (c - t) / 2
                taps = [((c - t) // 2, t) for t in range(kk[d]) if (c - t) % 2 == 0]
                out.append((c, 2, taps))
            return out
        for cz in classes(0):
            for cy in classes(1):
                for cx in classes(2):
                    cls = [cz, cy, cx]
                    if any(len(c[2]) == 0 for c in cls):
                        continue                                   # no tap reaches this class: stays zero
                    Sub = [(Pd[d] - cls[d][0] + cls[d][1] - 1) // cls[d][1] for d in range(3)]
                    bwd.append(_geom(Out, Pd, Sub, [c[0] for c in cls], [c[1] for c in cls], [1] * 3, [c[2] for c in cls], kk, 1))
    return Out, fwd, bwd, (In, Pd, lo, hi)


def convT_geoms(spatial):
    """ConvTranspose k3 s2 p1 op1: forward = one geometry per output parity class; input gradient = one geometry."""
    nd = len(spatial)
    In = _dims3(spatial)
    real = [False] * (3 - nd) + [True] * nd
    Out = [2 * In[d] if real[d] else 1 for d in range(3)]
    kk = [3 if r else 1 for r in real]

    def classes(d):
        if not real[d]:
            return [(0, 1, [(0, 0)])]
        # o = 2u + c, j = (o + 1 - t) / 2 = u + (c + 1 - t) / 2
        return [(c, 2, [((c + 1 - t) // 2, t) for t in range(3) if (c + 1 - t) % 2 == 0]) for c in range(2)]
    fwd = []
    for cz in classes(0):
        for cy in classes(1):
            for cx in classes(2):
                cls = [cz, cy, cx]
                fwd.append(_geom(In, Out, In, [c[0] for c in cls], [c[1] for c in cls], [1] * 3, [c[2] for c in cls], kk, 1))
    # dIn[j] = sum_t dOut[2j - 1 + t] w[t]^T
    bwd = _geom(Out, In, In, [0] * 3, [1] * 3, [2 if r else 1 for r in real],
                [[(t - 1, t) for t in range(kk[d])] if real[d] else [(0, 0)] for d in range(3)], kk, 1)
    return Out, fwd, bwd


class DenseConvFunction(torch.autograd.Function):
    """y rows = Conv(k, stride) of the replicate-padded volume (+ bias).  rows: (B * prod(spatial), Cin)."""

    @staticmethod
    def forward(ctx, rows, weight, bias, B, spatial, stride, pad_lo, pad_hi, stats=None, bias_grad=True, xf=None):
        """xf = (scale, shift) of cin floats each (not differentiable here, see dense_hip.BNConvFunction): the rows are
        used as rows * scale + shift (the producer's BatchNorm folded into this convolution's load).
        stats: zeroed new_stats(cout_p) slab to receive the column statistics of y (not differentiable);
        bias_grad False: the bias feeds a batch-statistics BatchNorm, its gradient is identically zero (the sum of a
        BatchNorm's input gradient over the rows vanishes) and is returned as zeros without the column-sum pass"""
        _l.require_gpu(rows)
        rows = rows.contiguous()
        nd = len(spatial)
        k = weight.shape[2]
        cout, cin = weight.shape[0], weight.shape[1]
        Out, fwd, bwd, padinfo = conv_geoms(spatial, k, stride, pad_lo, pad_hi)
        # weights as [tap][cout][cin], channel counts zero-padded to multiples of 16 (the 1-channel input conv, the
        # num_class-channel output conv)
        prepared = _WL.get(weight)
        if prepared is not None:
            wt, wb = prepared
        else:
            wt = weight.reshape(cout, cin, -1).permute(2, 0, 1).contiguous()
            wt = _pad16(_pad16(wt, 1), 2)
            wb = None
        xin = _pad16(rows, 1)
        cin_p, cout_p = wt.shape[2], wt.shape[1]
        bias_p = None if bias is None else _pad16(bias.contiguous(), 0)
        n_out = B * Out[0] * Out[1] * Out[2]
        y = torch.empty((n_out, cout_p), dtype=torch.float32, device=rows.device)
        xfp = None
        if xf is not None:        # padded channels: x = 0 there, scale 1 / shift 0 keeps them 0
            xfp = (torch.cat([xf[0], xf[0].new_ones(cin_p - cin)]) if cin_p != cin else xf[0].contiguous(),
                   torch.cat([xf[1], xf[1].new_zeros(cin_p - cin)]) if cin_p != cin else xf[1].contiguous())
        _launch(xin, cin_p, cin_p, wt, bias_p, y, cout_p, cout_p, B, fwd, stats, xfp)
        ctx.save_for_backward(xin, weight)
        ctx.xfp = xfp
        ctx.wb = wb                # the input gradient's operand, written by prepare_weights() together with wt (or None)
        ctx.meta = (B, tuple(spatial), stride, Out, bwd, padinfo, cin, cout, cin_p, cout_p, bias is not None)
        ctx.bias_grad = bias_grad
        ctx.out_spatial = tuple(Out[3 - nd:])
        return y[:, :cout].contiguous() if cout_p != cout else y

    @staticmethod
    def backward(ctx, dy):
        xin, weight = ctx.saved_tensors
        B, spatial, stride, Out, bwd, (In, Pd, lo, hi), cin, cout, cin_p, cout_p, has_bias = ctx.meta
        L = _l.load()
        dy = _pad16_rows(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dxp[pp][ci] = sum dy[o][co] W[co][ci][t]: weights as [tap][ci][co]
            wb = ctx.wb
            if wb is None:
                wb = weight.reshape(cout, cin, -1).permute(2, 1, 0).contiguous()
                wb = _pad16(_pad16(wb, 1), 2)
            n_p = B * Pd[0] * Pd[1] * Pd[2]
            full = stride == 1 or len(bwd) == 2 ** len(spatial)      # every padded position is written by some launch
            dxp = (torch.empty if full else torch.zeros)((n_p, cin_p), dtype=torch.float32, device=dy.device)
            I3 = ctypes.c_int * 3
            if stride == 1 and len(bwd) == 1 and (any(lo) or any(hi)):
                # one call: the kernel writes the voxels inside the volume straight to dxf, the fold touches the boundary only
                dxf = torch.empty((B * In[0] * In[1] * In[2], cin_p), dtype=torch.float32, device=dy.device)
                sb = L.urn_dense_conv_scratch_bytes(cin_p, B, ctypes.byref(bwd[0]))
                key = (dy.device, sb > 256)
                scratch = _SCRATCH.get(key)
                if scratch is None or scratch.numel() < sb:
                    scratch = _SCRATCH[key] = torch.empty(max(sb, 256), dtype=torch.uint8, device=dy.device)
                _l.check(L.urn_dense_conv_dgrad_fold(dy.data_ptr(), cout_p, cout_p, wb.data_ptr(), dxp.data_ptr(), dxf.data_ptr(), cin_p,
                                                     cin_p, B, ctypes.byref(bwd[0]), I3(*In), I3(*lo), I3(*hi), PRECISION,
                                                     scratch.data_ptr(), scratch.numel(), _l.stream()), 'dense_conv_dgrad_fold')
            else:
                for g in bwd:
                    _launch(dy, cout_p, cout_p, wb, None, dxp, cin_p, cin_p, B, g)
                if any(lo) or any(hi):
                    dxf = torch.empty((B * In[0] * In[1] * In[2], cin_p), dtype=torch.float32, device=dy.device)
                    _l.check(L.urn_dense_fold(dxp.data_ptr(), dxf.data_ptr(), B, I3(*In), I3(*lo), I3(*hi), cin_p, _l.stream()),
                             'dense_fold')
                else:
                    dxf = dxp
            dx = dxf[:, :cin].contiguous() if cin_p != cin else dxf
        if ctx.needs_input_grad[1]:
            dw = dense_conv_dw(xin, dy, weight.shape, B, spatial, stride, lo, Out, cin, cout, ctx.xfp)
        if has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy, cout) if ctx.bias_grad else zeros_f32(cout, dy.device)
        return dx, dw, db, None, None, None, None, None, None, None, None


def _dw_call(x, cx, dy, cy, B, in_dims, out_dims, kk, ss, lo, mode, vx, vy, wshape, xf=None):
    """dw[tap][cx][cy] = sum_o x[in(o, tap)] (x) dy[o]  (two stages, deterministic), written in torch's parameter layout
    (vy, vx, taps) = wshape -- the zero-padded channels beyond vx / vy dropped"""
    L = _l.load()
    I3 = ctypes.c_int * 3
    dw = torch.empty(wshape, dtype=torch.float32, device=x.device)
    sb = L.urn_dense_dw_scratch_bytes(B, I3(*out_dims), I3(*kk), cx, cy)
    scratch = torch.empty(sb, dtype=torch.uint8, device=x.device)
    _l.check(L.urn_dense_dw(x.data_ptr(), cx, cx, dy.data_ptr(), cy, cy, B, I3(*in_dims), I3(*out_dims), I3(*kk), I3(*ss),
                            I3(*lo), mode, dw.data_ptr(), 1, vx, vy, None if xf is None else xf[0].data_ptr(),
                            None if xf is None else xf[1].data_ptr(), scratch.data_ptr(), sb, PRECISION, _l.stream()), 'dense_dw')
    return dw


def dense_conv_dw(xin, dy, wshape, B, spatial, stride, lo, Out, cin, cout, xf=None):
    """weight gradient of the padded convolution; xin (rows, cin_p), dy (rows_out, cout_p), both zero-padded to 16"""
    nd = len(spatial)
    k = wshape[2]
    In = _dims3(spatial)
    real = [False] * (3 - nd) + [True] * nd
    kk = [k if r else 1 for r in real]
    ss = [stride if r else 1 for r in real]
    return _dw_call(xin, xin.shape[1], dy, dy.shape[1], B, In, Out, kk, ss, lo, 0, cin, cout, wshape, xf)


class DenseConvTransposeFunction(torch.autograd.Function):
    """y rows = ConvTranspose k3 s2 p1 op1 (+ bias); weight (Cin, Cout, *k) like torch."""

    @staticmethod
    def forward(ctx, rows, weight, bias, B, spatial, stats=None, bias_grad=True):
        _l.require_gpu(rows)
        rows = rows.contiguous()
        nd = len(spatial)
        cin, cout = weight.shape[0], weight.shape[1]
        Out, fwd, bwd = convT_geoms(spatial)
        prepared = _WL.get(weight)
        if prepared is not None:
            wt, wb = prepared
        else:
            wt = weight.reshape(cin, cout, -1).permute(2, 1, 0).contiguous()            # [tap][cout][cin]
            wt = _pad16(_pad16(wt, 1), 2)
            wb = None
        xin = _pad16(rows, 1)
        cin_p, cout_p = wt.shape[2], wt.shape[1]
        bias_p = None if bias is None else _pad16(bias.contiguous(), 0)
        y = torch.empty((B * Out[0] * Out[1] * Out[2], cout_p), dtype=torch.float32, device=rows.device)
        for g in fwd:
            _launch(xin, cin_p, cin_p, wt, bias_p, y, cout_p, cout_p, B, g, stats)
        ctx.save_for_backward(xin, weight)
        ctx.wb = wb
        ctx.meta = (B, tuple(spatial), Out, bwd, cin, cout, cin_p, cout_p, bias is not None)
        ctx.bias_grad = bias_grad
        ctx.out_spatial = tuple(Out[3 - nd:])
        return y[:, :cout].contiguous() if cout_p != cout else y

    @staticmethod
    def backward(ctx, dy):
        xin, weight = ctx.saved_tensors
        B, spatial, Out, bwd, cin, cout, cin_p, cout_p, has_bias = ctx.meta
        L = _l.load()
        nd = len(spatial)
        dy = _pad16_rows(dy)
        dx = dw = db = None
        In = _dims3(spatial)
        if ctx.needs_input_grad[0]:
            wb = ctx.wb
            if wb is None:
                wb = weight.reshape(cin, cout, -1).permute(2, 0, 1).contiguous()          # [tap][cin][cout]: kernel cout = cin
                wb = _pad16(_pad16(wb, 1), 2)
            dxf = torch.empty((B * In[0] * In[1] * In[2], cin_p), dtype=torch.float32, device=dy.device)
            _launch(dy, cout_p, cout_p, wb, None, dxf, cin_p, cin_p, B, bwd)
            dx = dxf[:, :cin].contiguous() if cin_p != cin else dxf
        if ctx.needs_input_grad[1]:
            # dW[ci][co][t] = sum_j x[j][ci] dy[2j - 1 + t][co]: the weight gradient of a stride-2 conv with the roles of
            # input and output swapped (dy is the "input" volume, x the "output" rows), pad_lo 1
            real = [False] * (3 - nd) + [True] * nd
            kk = [3 if r else 1 for r in real]
            ss = [2 if r else 1 for r in real]
            lo = [1 if r else 0 for r in real]
            dw = _dw_call(dy, cout_p, xin, cin_p, B, Out, In, kk, ss, lo, 1, cout, cin, weight.shape)
        if has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy, cout) if ctx.bias_grad else zeros_f32(cout, dy.device)
        return dx, dw, db, None, None, None, None
