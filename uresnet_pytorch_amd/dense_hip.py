"""GPU route of the dense U-ResNet blocks (reference uresnet/models/uresnet_dense.py:29-226).

Layout: activations are channels-last row matrices (B*D*H*W, C) fp32 inside this module (NC[D]HW at the boundary, as
views).  Convolutions run on the dense implicit-GEMM kernels (csrc/urn_dense.hip via dense_conv.py):
  * F.pad(mode='replicate') + Conv k{1,3} s{1,2} (reference :38-67, 75-80, 128-134, 179-197): ONE kernel call, the clamp
    is part of its addressing; input gradient = the same kernel on the padded volume + urn_dense_fold; weight gradient =
    urn_dense_dw (two stages, deterministic);
  * ConvTranspose k3 s2 p1 op1 (reference :164-172): one call per output parity class (only the valid taps each);
  * BatchNorm with batch statistics: the statistics come out of the producing convolution's epilogue (urn_dense_conv:
    stats), BatchNorm-apply + shortcut add (identity, or the shortcut conv's own BatchNorm) + ReLU are ONE row pass
    (urn_dense_bn_act_fwd), their gradient two (reduce, apply: urn_dense_bn_act_bwd_*) -- BNActFunction below; where
    neither a ReLU nor a shortcut follows (residual1 -> residual2) the BatchNorm is folded into the next convolution's load
    and never applied (BNFoldFunction, DeferredBN).  Channel
    counts those passes do not take (the num_class-wide output layer) use the sparse path's row kernels
    (urn_bn_relu_fwd/bwd).
No index tables, no padded copies (the first GPU route ran every conv as a gather convolution over [27][n] int32 tables:
226 MB per 128^3 conv, 8.4 GB per cfg2 step).  `-prec bf16` (BASELINE configs[1]) rounds the MFMA operands to bf16 in LDS;
tensors in HBM and the accumulation stay fp32.
"""
import torch

from . import dense_conv as dc
from . import lib as _l
from . import sparse_ops as so


def to_rows(x):
    """(B, C, *spatial) -> ((B*prod(spatial), C) rows, B, spatial).  Free (a view) when x is channels-last, which is
    what from_rows() hands out: activations stay row matrices between the layers, only the network input and whatever
    torch produces NC[D]HW-contiguous are copied."""
    B, C = x.shape[0], x.shape[1]
    spatial = tuple(x.shape[2:])
    perm = [0] + list(range(2, x.dim())) + [1]
    return x.permute(*perm).reshape(-1, C).contiguous(), B, spatial


def from_rows(rows, B, spatial):
    """rows -> (B, C, *spatial) as a channels-last VIEW of the row matrix (no copy)"""
    C = rows.shape[1]
    nd = len(spatial)
    perm = [0, nd + 1] + list(range(1, nd + 1))
    return rows.reshape(B, *spatial, C).permute(*perm)


def _conv_rows(rows, B, spatial, w, b, stride, pad):
    """replicate-pad + conv on row matrices; w is the torch weight (Cout, Cin, *k); pad = F.pad tuple of the model's
    padding(): the same (lo, hi) for every spatial dimension"""
    y = dc.DenseConvFunction.apply(rows, w, b, B, tuple(spatial), stride, int(pad[0]) if len(pad) else 0,
                                   int(pad[1]) if len(pad) else 0)
    k = w.shape[2]
    lo, hi = (int(pad[0]), int(pad[1])) if len(pad) else (0, 0)
    out_spatial = tuple((s + lo + hi - k) // stride + 1 for s in spatial)
    return y, out_spatial


def _bn(rows, gamma, beta, eps, relu):
    return so.BNReLUFunction.apply(rows, gamma, beta, None, None, eps, 0.0, bool(relu), True)


class DeferredBN(object):
    """raw output of a convolution whose BatchNorm has not been applied: handed to conv_bn_act(residual=...) so that the
    shortcut branch's BatchNorm, the add and the ReLU of a ResNetModule (reference uresnet_dense.py:72-82) are one pass"""

    is_cuda = True

    def __init__(self, raw, stats, gamma, beta, eps, B, spatial):
        self.raw, self.stats, self.gamma, self.beta, self.eps, self.B, self.spatial = raw, stats, gamma, beta, eps, B, spatial

    def size(self):
        """shape of the tensor this stands for: (B, C, *spatial)"""
        return torch.Size((self.B, self.raw.shape[1]) + tuple(self.spatial))

    def dim(self):
        return 2 + len(self.spatial)


def _finalize(stats, n, c, eps, gamma, beta):
    """statistics slab -> (mean, invstd, scale, shift), each (c,) fp32"""
    L = _l.load()
    o = torch.empty((4, c), dtype=torch.float32, device=stats.device)
    _l.check(L.urn_bn_finalize_fwd(stats.data_ptr(), stats.shape[0], n, c, stats.shape[2], float(eps), gamma.data_ptr(),
                                   beta.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), None, None,
                                   0.0, _l.stream()), 'bn_finalize_fwd')
    return o


class BNActFunction(torch.autograd.Function):
    """out = [relu](BN(raw) [+ res | + BN_s(res)]) with the batch statistics already in `stats` (convolution epilogue)"""

    @staticmethod
    def forward(ctx, raw, gamma, beta, stats, eps, relu, res, res_gamma, res_beta, res_stats, res_eps):
        L = _l.load()
        n, c = raw.shape
        gamma = gamma.contiguous(); beta = beta.contiguous()
        f = _finalize(stats, n, c, eps, gamma, beta)
        fr = None
        if res is not None:
            res = res.contiguous()
            if res_stats is not None:
                res_gamma = res_gamma.contiguous()
                fr = _finalize(res_stats, n, c, res_eps, res_gamma, res_beta.contiguous())
        out = torch.empty_like(raw)
        _l.check(L.urn_dense_bn_act_fwd(raw.data_ptr(), f[2].data_ptr(), f[3].data_ptr(), None if res is None else res.data_ptr(),
                                        None if fr is None else fr[2].data_ptr(), None if fr is None else fr[3].data_ptr(),
                                        1 if relu else 0, out.data_ptr(), n, c, _l.stream()), 'dense_bn_act_fwd')
        ctx.relu, ctx.has_res, ctx.res_bn = bool(relu), res is not None, fr is not None
        # the shortcut's raw rows are only needed for ITS BatchNorm's xhat
        ctx.save_for_backward(raw, out if relu else None, gamma, f, res if fr is not None else None,
                              res_gamma if fr is not None else None, fr)
        return out

    @staticmethod
    def backward(ctx, d_out):
        raw, out, gamma, f, res_raw, res_gamma, fr = ctx.saved_tensors
        L = _l.load()
        n, c = raw.shape
        d_out = d_out.contiguous()
        dev = raw.device
        slots = 64
        nb = 2 if ctx.res_bn else 1
        sums = dc.zeros_f64((nb, slots, 2, c), dev)
        o = torch.empty((8, c), dtype=torch.float32, device=dev)     # dgamma, dbeta, coef0, coef1 (main | shortcut)
        P = lambda t: None if t is None else t.data_ptr()
        _l.check(L.urn_dense_bn_act_bwd_reduce(d_out.data_ptr(), P(out), raw.data_ptr(), f[0].data_ptr(), f[1].data_ptr(),
                                               P(res_raw), None if fr is None else fr[0].data_ptr(),
                                               None if fr is None else fr[1].data_ptr(), n, c, sums[0].data_ptr(),
                                               sums[1].data_ptr() if ctx.res_bn else None, slots, _l.stream()), 'dense_bn_act_bwd_reduce')
        _l.check(L.urn_dense_bn_bwd_finalize(sums.data_ptr(), nb, slots, n, c, o.data_ptr(), _l.stream()), 'dense_bn_bwd_finalize')
        d_raw = torch.empty_like(raw)
        d_res = None
        if ctx.has_res:
            # identity shortcut without a ReLU: its gradient IS d_out
            d_res = torch.empty_like(raw) if (ctx.res_bn or ctx.relu) else d_out
        _l.check(L.urn_dense_bn_act_bwd_apply(d_out.data_ptr(), P(out), raw.data_ptr(), gamma.data_ptr(), f[0].data_ptr(),
                                              f[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), P(res_raw), P(res_gamma),
                                              None if fr is None else fr[0].data_ptr(), None if fr is None else fr[1].data_ptr(),
                                              o[6].data_ptr() if ctx.res_bn else None, o[7].data_ptr() if ctx.res_bn else None,
                                              d_raw.data_ptr(), d_res.data_ptr() if (ctx.has_res and d_res is not d_out) else None,
                                              n, c, _l.stream()), 'dense_bn_act_bwd_apply')
        return (d_raw, o[0], o[1], None, None, None, d_res, o[4] if ctx.res_bn else None, o[5] if ctx.res_bn else None, None, None)


class BNFoldFunction(torch.autograd.Function):
    """The BatchNorm of a raw convolution output, NOT applied: returns (raw, scale, shift) for a consumer that folds
    x * scale + shift into its load (urn_dense_conv / urn_dense_dw: xf).  The consumer's input gradient is the gradient
    w.r.t. the BatchNorm OUTPUT; it arrives here as the gradient of the first result and is taken through the BatchNorm
    (reduce + apply, urn_dense_bn_act_bwd_*) -- the gradients autograd would send to scale / shift are ignored, they are
    that same dependence counted a second time."""

    @staticmethod
    def forward(ctx, raw, gamma, beta, stats, eps):
        n, c = raw.shape
        gamma = gamma.contiguous(); beta = beta.contiguous()
        f = _finalize(stats, n, c, eps, gamma, beta)
        ctx.save_for_backward(raw, gamma, f)
        scale, shift = f[2].clone(), f[3].clone()
        ctx.mark_non_differentiable(scale, shift)
        ctx.set_materialize_grads(False)            # (autograd would hand backward() two freshly zero-filled (c,) tensors for them)
        return raw, scale, shift                    # the same rows: the consumer applies the affine map itself

    @staticmethod
    def backward(ctx, d_out, _ds, _dh):
        if d_out is None:
            return None, None, None, None, None
        raw, gamma, f = ctx.saved_tensors
        L = _l.load()
        n, c = raw.shape
        d_out = d_out.contiguous()
        slots = 64
        sums = dc.zeros_f64((1, slots, 2, c), raw.device)
        o = torch.empty((4, c), dtype=torch.float32, device=raw.device)
        _l.check(L.urn_dense_bn_act_bwd_reduce(d_out.data_ptr(), None, raw.data_ptr(), f[0].data_ptr(), f[1].data_ptr(), None, None,
                                               None, n, c, sums.data_ptr(), None, slots, _l.stream()), 'dense_bn_act_bwd_reduce')
        _l.check(L.urn_dense_bn_bwd_finalize(sums.data_ptr(), 1, slots, n, c, o.data_ptr(), _l.stream()), 'dense_bn_bwd_finalize')
        d_raw = torch.empty_like(raw)
        _l.check(L.urn_dense_bn_act_bwd_apply(d_out.data_ptr(), None, raw.data_ptr(), gamma.data_ptr(), f[0].data_ptr(),
                                              f[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), None, None, None, None, None, None,
                                              d_raw.data_ptr(), None, n, c, _l.stream()), 'dense_bn_act_bwd_apply')
        return d_raw, o[0], o[1], None, None


def _conv_raw(x, w, b, stride, pad):
    """convolution with the statistics of its output: (raw rows, stats slab or None, B, out_spatial).  x may be a
    DeferredBN (the raw output of the previous convolution, its BatchNorm pending): the BatchNorm is folded into this
    convolution's load and the normalised tensor is never written"""
    xf = None
    if isinstance(x, DeferredBN):
        rows, scale, shift = BNFoldFunction.apply(x.raw, x.gamma, x.beta, x.stats, x.eps)
        B, spatial = x.B, x.spatial
        xf = (scale, shift)
    else:
        rows, B, spatial = to_rows(x)
    cout = w.shape[0]
    fused = dc.stats_ok(cout)
    stats = dc.new_stats(cout + (-cout) % 16, rows.device) if fused else None
    lo, hi = (int(pad[0]), int(pad[1])) if len(pad) else (0, 0)
    y = dc.DenseConvFunction.apply(rows, w, b, B, tuple(spatial), stride, lo, hi, stats, False, xf)
    k = w.shape[2]
    out_spatial = tuple((s + lo + hi - k) // stride + 1 for s in spatial)
    return y, stats, B, out_spatial


def conv_bn_act(x, w, b, stride, pad, gamma, beta, eps, relu, residual=None, defer=False):
    y, stats, B, out_spatial = _conv_raw(x, w, b, stride, pad)
    if defer and stats is not None:
        return DeferredBN(y, stats, gamma, beta, eps, B, out_spatial)
    if stats is None:                       # channel counts the fused passes do not take
        if isinstance(residual, DeferredBN):
            residual = from_rows(_bn(residual.raw, residual.gamma, residual.beta, residual.eps, False), B, out_spatial)
        if residual is None:
            y = _bn(y, gamma, beta, eps, relu)
        else:
            r, _, _ = to_rows(residual)
            y = _bn(y, gamma, beta, eps, False) + r
            if relu:
                y = torch.relu(y)
        return from_rows(y, B, out_spatial)
    if isinstance(residual, DeferredBN):
        y = BNActFunction.apply(y, gamma, beta, stats, eps, relu, residual.raw, residual.gamma, residual.beta, residual.stats,
                                residual.eps)
    elif residual is not None:
        r, _, _ = to_rows(residual)
        y = BNActFunction.apply(y, gamma, beta, stats, eps, relu, r, None, None, None, 0.0)
    else:
        y = BNActFunction.apply(y, gamma, beta, stats, eps, relu, None, None, None, None, 0.0)
    return from_rows(y, B, out_spatial)


def convT_bn_act(x, w, b, gamma, beta, eps, relu):
    rows, B, spatial = to_rows(x)
    cout = w.shape[1]
    fused = dc.stats_ok(cout)
    stats = dc.new_stats(cout + (-cout) % 16, rows.device) if fused else None
    y = dc.DenseConvTransposeFunction.apply(rows, w, b, B, tuple(spatial), stats, False)
    if fused:
        y = BNActFunction.apply(y, gamma, beta, stats, eps, relu, None, None, None, None, 0.0)
    else:
        y = _bn(y, gamma, beta, eps, relu)
    return from_rows(y, B, tuple(2 * s for s in spatial))


class DenseCEFunction(torch.autograd.Function):
    """DenseSegmentationLoss of one event (reference uresnet_dense.py:246-258) as one pass on the device: returns
    (loss, out) with out = [loss, accuracy] (not differentiable); no host synchronisation."""

    @staticmethod
    def forward(ctx, logits, label, data, weight):
        _l.require_gpu(logits)
        L = _l.load()
        logits = logits.contiguous()
        n, nc = logits.shape
        row_lse = torch.empty(n, dtype=torch.float32, device=logits.device)
        acc = torch.zeros(3, dtype=torch.float64, device=logits.device)
        out = torch.empty(2, dtype=torch.float32, device=logits.device)
        _l.check(L.urn_dense_ce_fwd(logits.data_ptr(), nc, label.data_ptr(), data.data_ptr(), _l.ptr(weight), n, nc,
                                    row_lse.data_ptr(), acc.data_ptr(), out.data_ptr(), _l.stream()), 'dense_ce_fwd')
        ctx.save_for_backward(logits, label, data, row_lse, acc)
        ctx.w = weight
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, gloss, _gout):
        if gloss is None:
            return None, None, None, None
        logits, label, data, row_lse, acc = ctx.saved_tensors
        L = _l.load()
        n, nc = logits.shape
        g = gloss.contiguous().float().reshape(1)
        dl = torch.empty_like(logits)
        _l.check(L.urn_dense_ce_bwd(logits.data_ptr(), nc, label.data_ptr(), data.data_ptr(), _l.ptr(ctx.w), row_lse.data_ptr(),
                                    acc.data_ptr(), g.data_ptr(), n, nc, dl.data_ptr(), _l.stream()), 'dense_ce_bwd')
        return dl, None, None, None


def segmentation_loss_event(seg, data, label, weight):
    """seg (nc, *spatial) logits of one event (a channels-last view of the row matrix: no copy), data / label / weight
    with one value per voxel -> (loss, [loss, accuracy])"""
    nd = seg.dim() - 1
    rows = seg.permute(*range(1, nd + 1), 0).reshape(-1, seg.shape[0])
    n = rows.shape[0]
    f = lambda t: None if t is None else t.reshape(-1).float().contiguous()
    d, lab, w = f(data), f(label), f(weight)
    assert d.numel() == n and lab.numel() == n and (w is None or w.numel() == n)
    return DenseCEFunction.apply(rows, lab, d, w)
