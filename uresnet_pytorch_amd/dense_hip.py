"""GPU route of the dense U-ResNet blocks (reference uresnet/models/uresnet_dense.py:29-226), on the
same HIP kernels as the sparse path: a dense volume is the special case "every site active".

Layout: activations are channels-last row matrices (B*D*H*W, C) fp32 inside this module (NC[D]HW at
the boundary).  Every convolution is the gather convolution of the C ABI with a precomputed index
table (cached per shape):
  * F.pad(mode='replicate') (reference :75-80,208,222-224)  -> urn_rows_gather with a clamp table
    (backward: urn_rows_scatter_add);
  * Conv k{1,3} s{1,2} on the padded volume               -> urn_gconv_fwd, table[o][out] = in index;
    input gradient through the inverse table (a function, because the padded conv is 'valid');
  * ConvTranspose k3 s2 p1 op1 (reference :164-172)         -> urn_gconv_fwd, table with -1 holes;
  * BatchNorm with batch statistics (+ReLU)                 -> urn_bn_relu_fwd/bwd over the rows.
Every convolution runs on the MFMA gather-conv kernels (channel counts are zero-padded to multiples of 16, inputs
wider than 112 channels are walked in chunks); activations stay channels-last row matrices between the layers.
fp32; a bf16 variant for BASELINE configs[1] is future work.
"""
import torch

from . import sparse_ops as so

_TABLES = {}


def _grid(shape, dev):
    """coordinates of a (B, *spatial) row grid, each as a flat int64 tensor"""
    axes = [torch.arange(s, device=dev) for s in shape]
    return [g.reshape(-1) for g in torch.meshgrid(*axes, indexing='ij')]


def _flat(coords, shape):
    idx = coords[0]
    for c, s in zip(coords[1:], shape[1:]):
        idx = idx * s + c
    return idx


def pad_table(B, spatial, pad, dev):
    """rows of the replicate-padded volume -> source rows.  pad = (lo, hi) per spatial dim, torch F.pad order
    (last dim first), as returned by the model's padding()."""
    key = ('pad', B, tuple(spatial), tuple(pad), str(dev))
    if key not in _TABLES:
        nd = len(spatial)
        lo = [pad[2 * (nd - 1 - i)] for i in range(nd)]
        hi = [pad[2 * (nd - 1 - i) + 1] for i in range(nd)]
        pshape = [B] + [s + a + b for s, a, b in zip(spatial, lo, hi)]
        g = _grid(pshape, dev)
        src = [g[0]] + [(g[1 + i] - lo[i]).clamp_(0, spatial[i] - 1) for i in range(nd)]
        _TABLES[key] = (_flat(src, [B] + list(spatial)).to(torch.int32).contiguous(), tuple(pshape[1:]))
    return _TABLES[key]


def conv_tables(B, in_spatial, k, stride, dev):
    """'valid' convolution on an (already padded) volume: forward table [K][n_out], inverse [K][n_in]."""
    key = ('conv', B, tuple(in_spatial), k, stride, str(dev))
    if key not in _TABLES:
        nd = len(in_spatial)
        out_spatial = [(s - k) // stride + 1 for s in in_spatial]
        og = _grid([B] + out_spatial, dev)
        ig = _grid([B] + list(in_spatial), dev)
        n_out, n_in = og[0].numel(), ig[0].numel()
        fwd = torch.empty((k ** nd, n_out), dtype=torch.int32, device=dev)
        inv = torch.full((k ** nd, n_in), -1, dtype=torch.int32, device=dev)
        for o in range(k ** nd):
            ks = [(o // (k ** (nd - 1 - i))) % k for i in range(nd)]
            src = [og[0]] + [og[1 + i] * stride + ks[i] for i in range(nd)]
            fwd[o] = _flat(src, [B] + list(in_spatial)).to(torch.int32)
            # inverse: input i at offset ks feeds output (i - ks)/stride when divisible and in range
            ok = torch.ones(n_in, dtype=torch.bool, device=dev)
            dst = [ig[0]]
            for i in range(nd):
                t = ig[1 + i] - ks[i]
                ok &= (t >= 0) & (t % stride == 0) & (t // stride < out_spatial[i])
                dst.append(t // stride)
            j = _flat(dst, [B] + out_spatial).to(torch.int32)
            inv[o] = torch.where(ok, j, torch.full_like(j, -1))
        _TABLES[key] = (fwd.contiguous(), inv.contiguous(), tuple(out_spatial))
    return _TABLES[key]


def convT_tables(B, in_spatial, dev):
    """ConvTranspose k3 s2 p1 op1: out size 2*in; out[i] += in[j] w[k] with i = 2j - 1 + k."""
    key = ('convT', B, tuple(in_spatial), str(dev))
    if key not in _TABLES:
        nd, k = len(in_spatial), 3
        out_spatial = [2 * s for s in in_spatial]
        og = _grid([B] + out_spatial, dev)
        ig = _grid([B] + list(in_spatial), dev)
        n_out, n_in = og[0].numel(), ig[0].numel()
        fwd = torch.empty((k ** nd, n_out), dtype=torch.int32, device=dev)
        inv = torch.empty((k ** nd, n_in), dtype=torch.int32, device=dev)
        for o in range(k ** nd):
            ks = [(o // (k ** (nd - 1 - i))) % k for i in range(nd)]
            ok = torch.ones(n_out, dtype=torch.bool, device=dev)
            src = [og[0]]
            for i in range(nd):
                t = og[1 + i] + 1 - ks[i]
                ok &= (t >= 0) & (t % 2 == 0) & (t // 2 < in_spatial[i])
                src.append(t // 2)
            j = _flat(src, [B] + list(in_spatial)).to(torch.int32)
            fwd[o] = torch.where(ok, j, torch.full_like(j, -1))
            ok2 = torch.ones(n_in, dtype=torch.bool, device=dev)
            dst = [ig[0]]
            for i in range(nd):
                t = 2 * ig[1 + i] - 1 + ks[i]
                ok2 &= (t >= 0) & (t < out_spatial[i])
                dst.append(t)
            i2 = _flat(dst, [B] + out_spatial).to(torch.int32)
            inv[o] = torch.where(ok2, i2, torch.full_like(i2, -1))
        _TABLES[key] = (fwd.contiguous(), inv.contiguous(), tuple(out_spatial))
    return _TABLES[key]


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
    """replicate-pad + conv on row matrices; w is the torch weight (Cout, Cin, *k)"""
    dev = rows.device
    k = w.shape[2]
    nd = len(spatial)
    if any(pad):
        ptab, pspatial = pad_table(B, spatial, pad, dev)
        rows = so.RowsGatherFunction.apply(rows, ptab, ptab.numel())
    else:
        pspatial = spatial
    fwd, inv, out_spatial = conv_tables(B, pspatial, k, stride, dev)
    n_out, n_in = fwd.shape[1], inv.shape[1]
    perm = list(range(2, 2 + nd)) + [1, 0]
    wk = w.permute(*perm).reshape(k ** nd, w.shape[1], w.shape[0])          # (K, Cin, Cout)
    y = _gconv(rows, wk, fwd, inv, n_out, n_in)
    if b is not None:
        y = y + b
    return y, out_spatial


def _gconv(rows, wk, fwd, inv, n_out, n_in):
    """GConvFunction takes one leading dimension for both tables: pad the narrower one"""
    ld = max(fwd.shape[1], inv.shape[1])
    key = ('ld', fwd.data_ptr(), inv.data_ptr())
    if key not in _TABLES:
        def widen(t):
            if t.shape[1] == ld:
                return t
            out = torch.full((t.shape[0], ld), -1, dtype=torch.int32, device=t.device)
            out[:, :t.shape[1]] = t
            return out
        _TABLES[key] = (widen(fwd), widen(inv))
    f2, i2 = _TABLES[key]
    # channel counts that are not multiples of 16 (the 1-channel input conv, the num_class-channel output conv) are
    # zero-padded to the next multiple so that they run on the MFMA kernels: measured at 128^3 the VALU fallback took
    # 4.3 ms per launch against 0.6 ms for a 16-channel MFMA launch
    cin, cout = wk.shape[1], wk.shape[2]
    pin, pout = (-cin) % 16, (-cout) % 16
    if pin or pout:
        wk = torch.nn.functional.pad(wk, (0, pout, 0, pin))
        if pin:
            rows = torch.nn.functional.pad(rows, (0, pin))
    y = so.GConvFunction.apply(rows, wk, None, f2, i2, 0, ld, n_out, n_in)
    return y[:, :cout].contiguous() if pout else y


def _bn(rows, gamma, beta, eps, relu):
    return so.BNReLUFunction.apply(rows, gamma, beta, None, None, eps, 0.0, bool(relu), True)


def conv_bn_act(x, w, b, stride, pad, gamma, beta, eps, relu, residual=None):
    rows, B, spatial = to_rows(x)
    y, out_spatial = _conv_rows(rows, B, spatial, w, b, stride, pad)
    if residual is None:
        y = _bn(y, gamma, beta, eps, relu)
    else:
        r, _, _ = to_rows(residual)
        y = _bn(y, gamma, beta, eps, False) + r
        if relu:
            y = torch.relu(y)
    return from_rows(y, B, out_spatial)


def convT_bn_act(x, w, b, gamma, beta, eps, relu):
    rows, B, spatial = to_rows(x)
    dev = rows.device
    nd = len(spatial)
    fwd, inv, out_spatial = convT_tables(B, spatial, dev)
    perm = list(range(2, 2 + nd)) + [0, 1]
    wk = w.permute(*perm).reshape(3 ** nd, w.shape[0], w.shape[1])            # (K, Cin, Cout); torch: (Cin, Cout, *k)
    y = _gconv(rows, wk, fwd, inv, fwd.shape[1], inv.shape[1])
    if b is not None:
        y = y + b
    y = _bn(y, gamma, beta, eps, relu)
    return from_rows(y, B, out_spatial)
