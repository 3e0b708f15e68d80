"""The slice of the `sparseconvnet` module surface that the reference's sparse model
uses (reference uresnet/models/uresnet_sparse.py:9,19-24), re-implemented on the
MI355X-native C ABI:

    scn.Sequential().add(...), scn.InputLayer(dimension, spatial_size, mode=3),
    scn.SubmanifoldConvolution(dimension, nIn, nOut, filter_size, bias),
    scn.UNet(dimension, reps, nPlanes, residual_blocks=True, downsample=[2, 2]),
    scn.BatchNormReLU(nPlanes), scn.OutputLayer(dimension)

plus the building blocks scn.UNet is made of (Convolution, Deconvolution,
NetworkInNetwork, BatchNormLeakyReLU, ConcatTable, AddTable, JoinTable, Identity).
Module nesting and parameter names/shapes follow the published library so a
state_dict carries the same keys (weight (K, nIn, nOut), no bias).  Only what the
reference's call sites need is supported: dimension 3, filter 3 (submanifold),
filter 2 / stride 2 (strided), leakiness 0.
"""
import math

import torch
from torch import nn

from . import sparse_ops as so


class SparseConvNetTensor:
    """features (n_l, C) + the geometry they live on + the level index."""

    def __init__(self, features, geometry, level=0):
        self.features = features
        self.geometry = geometry
        self.level = level

    def with_features(self, f, level=None):
        return SparseConvNetTensor(f, self.geometry, self.level if level is None else level)


class Sequential(nn.Sequential):
    def add(self, module):
        self.add_module(str(len(self._modules)), module)
        return self

    def forward(self, x):
        for m in self._modules.values():
            x = m(x)
        return x


class Identity(nn.Module):
    def forward(self, x):
        return x


class ConcatTable(nn.Module):
    """fuse_add: the table feeds an AddTable and its last branch ends in a
    SubmanifoldConvolution -- the shortcut is then added in that kernel's epilogue."""
    fuse_add = False

    def add(self, module):
        self.add_module(str(len(self._modules)), module)
        return self

    def forward(self, x):
        mods = list(self._modules.values())
        if self.fuse_add and len(mods) == 2 and isinstance(mods[1], Sequential):
            tail = list(mods[1]._modules.values())
            if isinstance(tail[-1], SubmanifoldConvolution):
                shortcut = mods[0](x)
                t = x
                for m in tail[:-1]:
                    t = m(t)
                return [tail[-1](t, res=shortcut.features)]
        return [m(x) for m in mods]


class AddTable(nn.Module):
    def forward(self, xs):
        out = xs[0].features
        for t in xs[1:]:
            out = out + t.features
        return xs[0].with_features(out)


class JoinTable(nn.Module):
    def forward(self, xs):
        return xs[0].with_features(torch.cat([t.features for t in xs], dim=1))


class InputLayer(nn.Module):
    """mode 3: duplicate coordinates are summed.  `num_levels` is a hint (set by the
    enclosing model) so that all strided levels are built in the same integer phase."""

    def __init__(self, dimension, spatial_size, mode=3):
        super().__init__()
        if dimension != 3:
            raise ValueError('only dimension 3 is supported by the HIP path')
        if mode != 3:
            raise ValueError('only InputLayer mode 3 (sum duplicates) is supported')
        self.dimension = dimension
        self.spatial_size = int(spatial_size if not hasattr(spatial_size, '__len__') else spatial_size[0])
        self.mode = mode
        self.num_levels = 1

    def forward(self, inp):
        coords, features = inp
        c = coords.to(torch.int32) if coords.dtype != torch.int32 else coords   # float -> int: truncation
        geo = so.SparseGeometry(c, self.spatial_size, self.num_levels)
        return SparseConvNetTensor(so.input_features(geo, features), geo, 0)


class OutputLayer(nn.Module):
    def __init__(self, dimension):
        super().__init__()
        self.dimension = dimension

    def forward(self, x):
        g = x.geometry
        return so.RowsGatherFunction.apply(x.features, g.row2site, g.n_rows)


def _init_conv(weight, fan):
    with torch.no_grad():
        weight.normal_(0, math.sqrt(2.0 / fan))


class SubmanifoldConvolution(nn.Module):
    def __init__(self, dimension, nIn, nOut, filter_size, bias):
        super().__init__()
        if dimension != 3 or filter_size != 3 or bias:
            raise ValueError('supported: dimension 3, filter_size 3, bias False')
        self.nIn, self.nOut = nIn, nOut
        self.weight = nn.Parameter(torch.empty(27, nIn, nOut))
        _init_conv(self.weight, nIn * 27)

    def forward(self, x, res=None):
        g, l = x.geometry, x.level
        y = so.GConvFunction.apply(x.features, self.weight, res, g.nbr[l], g.nbr[l], 1, g.ld, g.n[l], g.n[l],
                                   g.pairs['nbr'][l], g.pairs['nbr'][l])
        return x.with_features(y)


class Convolution(nn.Module):
    """filter 2 / stride 2: level l -> l+1"""

    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias):
        super().__init__()
        if dimension != 3 or filter_size != 2 or filter_stride != 2 or bias:
            raise ValueError('supported: dimension 3, filter_size 2, stride 2, bias False')
        self.nIn, self.nOut = nIn, nOut
        self.weight = nn.Parameter(torch.empty(8, nIn, nOut))
        _init_conv(self.weight, nIn * 8)

    def forward(self, x):
        g, l = x.geometry, x.level
        if l + 1 >= g.num_levels:
            raise RuntimeError('geometry was built with %d levels; InputLayer.num_levels too small' % g.num_levels)
        y = so.GConvFunction.apply(x.features, self.weight, None, g.chd[l], g.up[l], 0, g.ld, g.n[l + 1], g.n[l],
                                   g.pairs['chd'][l], g.pairs['up'][l])
        return x.with_features(y, l + 1)


class Deconvolution(nn.Module):
    """filter 2 / stride 2 transpose: level l+1 -> l (reuses the strided tables)"""

    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias):
        super().__init__()
        if dimension != 3 or filter_size != 2 or filter_stride != 2 or bias:
            raise ValueError('supported: dimension 3, filter_size 2, stride 2, bias False')
        self.nIn, self.nOut = nIn, nOut
        self.weight = nn.Parameter(torch.empty(8, nIn, nOut))
        _init_conv(self.weight, nIn * 8)

    def forward(self, x):
        g, l = x.geometry, x.level - 1
        y = so.GConvFunction.apply(x.features, self.weight, None, g.up[l], g.chd[l], 0, g.ld, g.n[l], g.n[l + 1],
                                   g.pairs['up'][l], g.pairs['chd'][l])
        return x.with_features(y, l)


class NetworkInNetwork(nn.Module):
    """1x1 linear on rows == gather-conv with the centre (identity) table only."""

    def __init__(self, nIn, nOut, bias):
        super().__init__()
        if bias:
            raise ValueError('bias not supported')
        self.nIn, self.nOut = nIn, nOut
        self.weight = nn.Parameter(torch.empty(nIn, nOut))
        _init_conv(self.weight, nIn)

    def forward(self, x):
        g, l = x.geometry, x.level
        ident = g.nbr[l][13:14]       # centre offset of the submanifold table = identity map
        y = so.GConvFunction.apply(x.features, self.weight.unsqueeze(0), None, ident, ident, 0, g.ld, g.n[l], g.n[l],
                                   so.IDENT_PAIRS, so.IDENT_PAIRS)
        return x.with_features(y)


class BatchNormLeakyReLU(nn.Module):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, leakiness=0):
        super().__init__()
        if leakiness != 0:
            raise ValueError('only leakiness 0 is supported')
        self.nPlanes, self.eps, self.momentum = nPlanes, eps, momentum
        self.weight = nn.Parameter(torch.ones(nPlanes))
        self.bias = nn.Parameter(torch.zeros(nPlanes))
        self.register_buffer('running_mean', torch.zeros(nPlanes))
        self.register_buffer('running_var', torch.ones(nPlanes))

    def forward(self, x):
        y = so.BNReLUFunction.apply(x.features, self.weight, self.bias, self.running_mean, self.running_var,
                                    self.eps, self.momentum, True, self.training)
        return x.with_features(y)


class BatchNormReLU(BatchNormLeakyReLU):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9):
        super().__init__(nPlanes, eps, momentum, 0)


def UNet(dimension, reps, nPlanes, residual_blocks=False, downsample=(2, 2), leakiness=0):
    """Same recursive construction as the published scn.UNet (residual_blocks=True path)."""
    if not residual_blocks:
        raise ValueError('only residual_blocks=True (the reference configuration) is supported')

    def block(m, a, b):
        table = (ConcatTable()
                 .add(Identity() if a == b else NetworkInNetwork(a, b, False))
                 .add(Sequential()
                      .add(BatchNormLeakyReLU(a, leakiness=leakiness))
                      .add(SubmanifoldConvolution(dimension, a, b, 3, False))
                      .add(BatchNormLeakyReLU(b, leakiness=leakiness))
                      .add(SubmanifoldConvolution(dimension, b, b, 3, False))))
        table.fuse_add = True
        m.add(table).add(AddTable())

    def U(planes):
        m = Sequential()
        for _ in range(reps):
            block(m, planes[0], planes[0])
        if len(planes) > 1:
            m.add(ConcatTable()
                  .add(Identity())
                  .add(Sequential()
                       .add(BatchNormLeakyReLU(planes[0], leakiness=leakiness))
                       .add(Convolution(dimension, planes[0], planes[1], downsample[0], downsample[1], False))
                       .add(U(planes[1:]))
                       .add(BatchNormLeakyReLU(planes[1], leakiness=leakiness))
                       .add(Deconvolution(dimension, planes[1], planes[0], downsample[0], downsample[1], False))))
            m.add(JoinTable())
            for i in range(reps):
                block(m, planes[0] * (2 if i == 0 else 1), planes[0])
        return m

    return U(list(nPlanes))
