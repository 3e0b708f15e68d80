"""GPU route of the dense U-ResNet blocks (reference uresnet/models/uresnet_dense.py:29-226).

Layout: activations are channels-last row matrices (B*D*H*W, C) fp32 inside this module (NC[D]HW at the boundary, as
views).  Convolutions run on the dense implicit-GEMM kernels (csrc/urn_dense.hip via dense_conv.py):
  * F.pad(mode='replicate') + Conv k{1,3} s{1,2} (reference :38-67, 75-80, 128-134, 179-197): ONE kernel call, the clamp
    is part of its addressing; input gradient = the same kernel on the padded volume + urn_dense_fold; weight gradient =
    urn_dense_dw (two stages, deterministic);
  * ConvTranspose k3 s2 p1 op1 (reference :164-172): one call per output parity class (only the valid taps each);
  * BatchNorm with batch statistics (+ReLU): urn_bn_relu_fwd/bwd over the rows (the sparse path's row kernels).
No index tables, no padded copies (the first GPU route ran every conv as a gather convolution over [27][n] int32 tables:
226 MB per 128^3 conv, 8.4 GB per cfg2 step).  `-prec bf16` (BASELINE configs[1]) rounds the MFMA operands to bf16 in LDS;
tensors in HBM and the accumulation stay fp32.
"""
import torch

from . import dense_conv as dc
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
    y = dc.DenseConvTransposeFunction.apply(rows, w, b, B, tuple(spatial))
    y = _bn(y, gamma, beta, eps, relu)
    return from_rows(y, B, tuple(2 * s for s in spatial))
