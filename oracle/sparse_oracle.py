"""CPU oracle for the sparse U-ResNet path: ctypes wrappers over sparse_ref.c plus
an explicit forward/backward of the whole network in numpy.

TEST INFRASTRUCTURE ONLY (see sparse_ref.c header): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product.
PARITY STATUS: "parity unpinned" by the reference (sparseconvnet absent, no
reference tests); pinned by dense equivalence (oracle/dense_equiv.py).

Network graph restated from reference uresnet/models/uresnet_sparse.py:12-37
and the published body of scn.UNet:

  resblock(a,b): y = (a==b ? x : NiN(a->b)(x)) + SubM3(b,b)(BNReLU(b)(SubM3(a,b)(BNReLU(a)(x))))
  U(l): resblock(P_l,P_l) x reps; if not last:
           z = Deconv(BNReLU(U(l+1)(Conv_k2s2(BNReLU(x)))));  x = concat(x, z)
           resblock(2P_l,P_l); resblock(P_l,P_l) ...
  net: InputLayer(mode 3) -> SubM3(1->m) -> U(0) -> BNReLU(m) -> OutputLayer -> Linear(m->nc)
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
BN_EPS = 1e-4  # scn BatchNormalization default eps


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE, 'liboracle.so'])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.orc_sites_build.restype = ctypes.c_int64
        L.orc_rulebook_subm.restype = ctypes.c_int64
        L.orc_level_down.restype = ctypes.c_int64
        L.orc_num_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


I64 = ctypes.c_int64
CI = ctypes.c_int


# ------------------------------------------------------------ integer ops --
def sites_build(coords, feats, mode=3):
    coords = _i32(coords); feats = _f32(feats)
    N, nf = coords.shape[0], feats.shape[1]
    row2site = np.empty(N, np.int32)
    sc = np.empty((max(N, 1), 4), np.int32)
    sf = np.empty((max(N, 1), nf), np.float32)
    na = lib().orc_sites_build(_p(coords), I64(N), _p(feats), CI(nf), CI(mode), _p(row2site), _p(sc), _p(sf))
    return row2site, sc[:na].copy(), sf[:na].copy()


def rulebook_subm(site_coords, spatial):
    sc = _i32(site_coords); Na = sc.shape[0]
    nbr = np.empty((27, Na), np.int32)
    R = lib().orc_rulebook_subm(_p(sc), I64(Na), CI(spatial), _p(nbr))
    return nbr, int(R)


def level_down(fine_coords):
    fc = _i32(fine_coords); Nf = fc.shape[0]
    cc = np.empty((max(Nf, 1), 4), np.int32)
    parent = np.empty(Nf, np.int32); off = np.empty(Nf, np.int32)
    nc = lib().orc_level_down(_p(fc), I64(Nf), _p(cc), _p(parent), _p(off))
    chd = np.empty((8, nc), np.int32)
    lib().orc_children(_p(parent), _p(off), I64(Nf), I64(nc), _p(chd))
    up = np.empty((8, Nf), np.int32)
    lib().orc_up_table(_p(parent), _p(off), I64(Nf), _p(up))
    return cc[:nc].copy(), parent, off, chd, up


def invert_table(nbr, Nin):
    nbr = _i32(nbr); K, Nout = nbr.shape
    inv = np.empty((K, Nin), np.int32)
    lib().orc_invert_table(_p(nbr), CI(K), I64(Nout), I64(Nin), _p(inv))
    return inv


def canonical_triples(nbr):
    """Sorted (offset, in, out) int32 triples of a gather table [K][Nout]."""
    o, j = np.nonzero(nbr >= 0)
    t = np.stack([o, nbr[o, j], j], axis=1).astype(np.int32)
    order = np.lexsort((t[:, 2], t[:, 1], t[:, 0]))
    return t[order]


# -------------------------------------------------------------- float ops --
def conv_fwd(x, W, nbr):
    x = _f32(x); W = _f32(W); nbr = _i32(nbr)
    K, Nout = nbr.shape; Cin, Cout = W.shape[1], W.shape[2]
    y = np.empty((Nout, Cout), np.float32)
    lib().orc_conv_fwd(_p(x), _p(W), _p(nbr), CI(K), I64(Nout), CI(Cin), CI(Cout), _p(y))
    return y


def conv_bwd(x, W, nbr, dy, inv=None):
    x = _f32(x); W = _f32(W); nbr = _i32(nbr); dy = _f32(dy)
    K, Nout = nbr.shape; Cin, Cout = W.shape[1], W.shape[2]; Nin = x.shape[0]
    if inv is None:
        inv = invert_table(nbr, Nin)
    dx = np.empty((Nin, Cin), np.float32)
    lib().orc_conv_bwd_dx(_p(dy), _p(W), _p(inv), CI(K), I64(Nin), CI(Cin), CI(Cout), _p(dx))
    dW = np.empty_like(W)
    lib().orc_conv_bwd_dw(_p(x), _p(dy), _p(nbr), CI(K), I64(Nout), CI(Cin), CI(Cout), _p(dW))
    return dx, dW


def bn_relu_fwd(x, gamma, beta, relu=True, eps=BN_EPS):
    x = _f32(x); N, C = x.shape
    y = np.empty_like(x); mean = np.empty(C, np.float32); invstd = np.empty(C, np.float32)
    lib().orc_bn_relu_fwd(_p(x), I64(N), CI(C), _p(_f32(gamma)), _p(_f32(beta)), ctypes.c_double(eps),
                          CI(int(relu)), _p(y), _p(mean), _p(invstd))
    return y, mean, invstd


def bn_relu_bwd(x, y, dy, gamma, mean, invstd, relu=True):
    x = _f32(x); N, C = x.shape
    dx = np.empty_like(x); dg = np.empty(C, np.float32); db = np.empty(C, np.float32)
    lib().orc_bn_relu_bwd(_p(x), _p(_f32(y)), _p(_f32(dy)), I64(N), CI(C), _p(_f32(gamma)), _p(_f32(mean)),
                          _p(_f32(invstd)), CI(int(relu)), _p(dx), _p(dg), _p(db))
    return dx, dg, db


# ------------------------------------------------------- parameter naming --
def param_specs(m, num_strides, num_class, reps=2, nin=1):
    """Ordered (name, shape, kind, fan) for the whole model.  Names follow the nested
    scn.Sequential numbering the reference's state_dict would carry
    (reference uresnet_sparse.py:19-25)."""
    planes = [i * m for i in range(1, num_strides + 1)]
    specs = [('sparseModel.1.weight', (27, nin, m), 'conv', nin * 27)]

    def bn(prefix, c):
        return [(prefix + '.weight', (c,), 'bn_w', 0), (prefix + '.bias', (c,), 'bn_b', 0)]

    def block(prefix, idx, a, b):
        p = '%s.%d' % (prefix, idx)
        s = []
        if a != b:
            s.append((p + '.0.weight', (a, b), 'nin', a))
        s += bn(p + '.1.0', a)
        s.append((p + '.1.1.weight', (27, a, b), 'conv', a * 27))
        s += bn(p + '.1.2', b)
        s.append((p + '.1.3.weight', (27, b, b), 'conv', b * 27))
        return s

    def U(prefix, pl):
        s = []
        idx = 0
        for _ in range(reps):
            s += block(prefix, idx, pl[0], pl[0]); idx += 2
        if len(pl) > 1:
            p = '%s.%d.1' % (prefix, idx)
            s += bn(p + '.0', pl[0])
            s.append((p + '.1.weight', (8, pl[0], pl[1]), 'conv', pl[0] * 8))
            s += U(p + '.2', pl[1:])
            s += bn(p + '.3', pl[1])
            s.append((p + '.4.weight', (8, pl[1], pl[0]), 'conv', pl[1] * 8))
            idx += 2  # ConcatTable + JoinTable
            for i in range(reps):
                s += block(prefix, idx, pl[0] * (2 if i == 0 else 1), pl[0]); idx += 2
        return s

    specs += U('sparseModel.2', planes)
    specs += bn('sparseModel.3', m)
    specs += [('linear.weight', (num_class, m), 'lin_w', m), ('linear.bias', (num_class,), 'lin_b', m)]
    return specs


def init_params(m, num_strides, num_class, seed=0, reps=2):
    """Deterministic numpy init: conv/NiN normal(0, sqrt(2/fan)), BN weight 1 (+small
    jitter so tests see non-trivial affine), bias 0 (+jitter), linear uniform."""
    rng = np.random.default_rng(seed)
    P = {}
    for name, shape, kind, fan in param_specs(m, num_strides, num_class, reps):
        if kind in ('conv', 'nin'):
            P[name] = (rng.normal(size=shape) * np.sqrt(2.0 / fan)).astype(np.float32)
        elif kind == 'bn_w':
            P[name] = (1.0 + 0.1 * rng.normal(size=shape)).astype(np.float32)
        elif kind == 'bn_b':
            P[name] = (0.1 * rng.normal(size=shape)).astype(np.float32)
        else:
            b = 1.0 / np.sqrt(fan)
            P[name] = rng.uniform(-b, b, size=shape).astype(np.float32)
    return P


# ------------------------------------------------------------- the network --
class Geometry:
    """Integer side of one forward: sites per level, subm tables, down/up tables."""

    def __init__(self, coords, feats, spatial, num_levels, mode=3):
        self.row2site, sc, self.feats = sites_build(coords, feats, mode)
        self.coords = [sc]
        self.nbr, self.nbr_inv, self.R = [], [], []
        self.parent, self.off, self.chd, self.up, self.up_inv, self.chd_inv = [], [], [], [], [], []
        sp = spatial
        for l in range(num_levels):
            n, R = rulebook_subm(self.coords[l], sp)
            self.nbr.append(n); self.R.append(R)
            # subm table inverse is the mirrored table: inv[o] = nbr[26-o]
            self.nbr_inv.append(np.ascontiguousarray(n[::-1]))
            if l + 1 < num_levels:
                cc, parent, off, chd, up = level_down(self.coords[l])
                self.coords.append(cc)
                self.parent.append(parent); self.off.append(off); self.chd.append(chd); self.up.append(up)
                # down conv gathers fine->coarse via chd; its inverse (per fine row) is `up`
                self.chd_inv.append(up); self.up_inv.append(chd)
                sp = (sp + 1) // 2

    @property
    def n(self):
        return [len(c) for c in self.coords]


class SparseUResNetOracle:
    def __init__(self, params, m, num_strides, num_class, spatial, reps=2):
        self.P = params; self.m = m; self.L = num_strides; self.nc = num_class
        self.spatial = spatial; self.reps = reps
        self.planes = [i * m for i in range(1, num_strides + 1)]

    # -- primitive steps that push closures onto the tape ---------------
    def _conv(self, name, x, nbr, inv):
        W = self.P[name]
        y = conv_fwd(x, W, nbr)

        def back(dy):
            dx, dW = conv_bwd(x, W, nbr, dy, inv)
            self.G[name] = self.G.get(name, 0) + dW
            return dx
        return y, back

    def _nin(self, name, x):
        W = self.P[name]
        y = (x.astype(np.float64) @ W.astype(np.float64)).astype(np.float32)

        def back(dy):
            d = dy.astype(np.float64)
            self.G[name] = self.G.get(name, 0) + (x.astype(np.float64).T @ d).astype(np.float32)
            return (d @ W.astype(np.float64).T).astype(np.float32)
        return y, back

    def _bn(self, prefix, x):
        g, b = self.P[prefix + '.weight'], self.P[prefix + '.bias']
        masks = getattr(self, 'masks', None)
        if masks is not None and prefix in masks:
            # ReLU mask pinned from outside (the GPU's): y = mask * (affine-normalised x).  A pre-activation within fp32
            # rounding of zero flips between any two evaluation orders; with the mask pinned, forward and backward of
            # both sides take the same branch and what is left to compare is arithmetic.
            y, mean, invstd = bn_relu_fwd(x, g, b, False)
            y = np.where(masks[prefix], y, np.float32(0)).astype(np.float32)
            # (the backward reads the mask from y > 0: make kept-but-nonpositive entries count as kept)
            ymask = masks[prefix].astype(np.float32)
        else:
            y, mean, invstd = bn_relu_fwd(x, g, b, True)
            ymask = y
        if getattr(self, 'keep_acts', False):
            self.acts[prefix] = y
            # the pre-activation as well: tests bound |pre| at every entry whose ReLU branch differs from the GPU's
            self.pre[prefix] = bn_relu_fwd(x, g, b, False)[0]

        def back(dy):
            dx, dg, db = bn_relu_bwd(x, ymask, dy, g, mean, invstd, True)
            self.G[prefix + '.weight'] = self.G.get(prefix + '.weight', 0) + dg
            self.G[prefix + '.bias'] = self.G.get(prefix + '.bias', 0) + db
            return dx
        return y, back

    def _block(self, prefix, idx, a, b, x, l):
        p = '%s.%d' % (prefix, idx)
        geo = self.geo
        backs = []
        if a != b:
            sc, bk_sc = self._nin(p + '.0.weight', x)
        else:
            sc, bk_sc = x, None
        t, bk0 = self._bn(p + '.1.0', x)
        t, bk1 = self._conv(p + '.1.1.weight', t, geo.nbr[l], geo.nbr_inv[l])
        t, bk2 = self._bn(p + '.1.2', t)
        t, bk3 = self._conv(p + '.1.3.weight', t, geo.nbr[l], geo.nbr_inv[l])
        y = sc + t

        def back(dy):
            d = bk0(bk1(bk2(bk3(dy))))
            return d + (bk_sc(dy) if bk_sc else dy)
        return y, back

    def _U(self, prefix, l, x):
        pl = self.planes[l:]
        geo = self.geo
        backs = []
        idx = 0
        for _ in range(self.reps):
            x, bk = self._block(prefix, idx, pl[0], pl[0], x, l); backs.append(bk); idx += 2
        if len(pl) > 1:
            p = '%s.%d.1' % (prefix, idx)
            skip = x
            t, b0 = self._bn(p + '.0', x)
            t, b1 = self._conv(p + '.1.weight', t, geo.chd[l], geo.chd_inv[l])
            t, b2 = self._U(p + '.2', l + 1, t)
            t, b3 = self._bn(p + '.3', t)
            t, b4 = self._conv(p + '.4.weight', t, geo.up[l], geo.up_inv[l])
            x = np.concatenate([skip, t], axis=1)
            c0 = pl[0]

            def back_join(dy, b0=b0, b1=b1, b2=b2, b3=b3, b4=b4, c0=c0):
                dskip = dy[:, :c0]
                dz = np.ascontiguousarray(dy[:, c0:])
                return dskip + b0(b1(b2(b3(b4(dz)))))
            backs.append(back_join)
            idx += 2
            for i in range(self.reps):
                x, bk = self._block(prefix, idx, pl[0] * (2 if i == 0 else 1), pl[0], x, l)
                backs.append(bk); idx += 2

        def back(dy):
            for bk in reversed(backs):
                dy = bk(dy)
            return dy
        return x, back

    def forward(self, point_cloud):
        """point_cloud (N, 5) [x,y,z,batch,value] -> logits (N, nc).  Keeps the tape."""
        pc = np.asarray(point_cloud)
        coords = pc[:, :4].astype(np.float32).astype(np.int64).astype(np.int32)
        feats = pc[:, 4:5].astype(np.float32)
        self.geo = geo = Geometry(coords, feats, self.spatial, self.L, mode=3)
        self.G = {}
        self.acts = {}
        self.pre = {}
        x, b_stem = self._conv('sparseModel.1.weight', geo.feats, geo.nbr[0], geo.nbr_inv[0])
        x, b_u = self._U('sparseModel.2', 0, x)
        x, b_bn = self._bn('sparseModel.3', x)
        rows = x[geo.row2site]                               # OutputLayer
        Wl, bl = self.P['linear.weight'], self.P['linear.bias']
        logits = (rows.astype(np.float64) @ Wl.astype(np.float64).T + bl).astype(np.float32)
        na = x.shape[0]

        def backward(dlogits):
            d = dlogits.astype(np.float64)
            self.G['linear.weight'] = (d.T @ rows.astype(np.float64)).astype(np.float32)
            self.G['linear.bias'] = d.sum(0).astype(np.float32)
            drows = d @ Wl.astype(np.float64)
            dx = np.zeros((na, self.m), np.float64)
            np.add.at(dx, geo.row2site, drows)
            dfeat = b_stem(b_u(b_bn(dx.astype(np.float32))))
            return dfeat
        self._backward = backward
        self.features = x
        return logits

    def backward(self, dlogits):
        """Returns (grads dict, d/d(site features))."""
        dfeat = self._backward(np.asarray(dlogits, np.float32))
        return self.G, dfeat


def segmentation_loss(logits, data, label, weight=None):
    """Restates reference uresnet_sparse.py:46-82 for ONE gpu entry: sum over events
    (batch id = column -2) of mean voxel CE (optionally weighted), accuracy sum.
    Returns (loss, acc, dloss/dlogits)."""
    lg = logits.astype(np.float64)
    bid = np.asarray(data)[:, -2]
    lab = np.asarray(label).reshape(-1).astype(np.int64)
    z = lg - lg.max(1, keepdims=True)
    lse = np.log(np.exp(z).sum(1, keepdims=True))
    logp = z - lse
    ce = -logp[np.arange(len(lab)), lab]
    p = np.exp(logp)
    dl = np.zeros_like(lg)
    total, acc = 0.0, 0.0
    for b in np.unique(bid):
        idx = np.nonzero(bid == b)[0]
        w = np.ones(len(idx)) if weight is None else np.asarray(weight).reshape(-1)[idx].astype(np.float64)
        total += float(np.mean(ce[idx] * w))
        g = p[idx].copy()
        g[np.arange(len(idx)), lab[idx]] -= 1.0
        dl[idx] = g * (w / len(idx))[:, None]
        acc += float((lg[idx].argmax(1) == lab[idx]).sum()) / float(len(idx))
    return total, acc, dl.astype(np.float32)
