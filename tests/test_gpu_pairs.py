"""GPU parity tests of the compacted rule lists ("pair lists") and of the gather convolution that walks them
(uresnet_pytorch_amd/csrc/urn_gconv_pairs.hip), through the C ABI, against the CPU oracle.

Lists: bit-exact against a numpy restatement of the layout documented in include/uresnet_hip.h.
Convolutions: forward, input gradient and weight gradient within 1e-5 relative (norm-wise, the tolerance of
BASELINE.json's north_star) -- on small clouds for every channel shape, and on the REAL BASELINE configs[2] geometry
(512^3, 50,000 voxels, 5 levels) for every (level, cin, cout) the network runs there.  Results are bitwise
reproducible from run to run (no float atomics across waves)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import sparse_oracle as orc
from uresnet_pytorch_amd.iotools.synthetic import generate_event, make_sparse_blob

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    from uresnet_pytorch_amd import lib
    lib.load()
    return torch.device('cuda:0')


def cloud(seed, S, n, nbatch):
    cs = []
    for b in range(nbatch):
        c, _, _ = generate_event(seed * 10 + b, S, n)
        cs.append(np.concatenate([c, np.full((len(c), 1), b, np.int32)], 1))
    c = np.ascontiguousarray(np.concatenate(cs, 0), np.int32)
    f = np.random.default_rng(seed).normal(size=(len(c), 1)).astype(np.float32)
    return c, f


def pairs_ref(tbl, n, T):
    """numpy restatement of the list layout: per tile [nblk | 28 bytes: first block of every table row | t of every block | 16 words per block]"""
    K = tbl.shape[0]
    maxb = K * (T // 16)
    out = []
    for tile in range((n + T - 1) // T):
        rows = np.arange(tile * T, min(n, (tile + 1) * T))
        ts, words = [], []
        for t in range(K):
            v = tbl[t, rows]
            ok = v >= 0
            w = (v[ok].astype(np.int64) | ((rows[ok] - tile * T).astype(np.int64) << 24))
            pad = (-len(w)) % 16
            w = np.concatenate([w, np.full(pad, T << 24, np.int64)])
            ts += [t] * (len(w) // 16)
            words.append(w)
        out.append((len(ts), np.array(ts, np.int64), np.concatenate(words) if words else np.zeros(0, np.int64), maxb))
    return out


def check_pairs(lst, tile, tbl, n, K):
    maxb = K * (tile // 16)
    tpad = (maxb + 3) // 4 * 4
    words = 16 + tpad + maxb * 16
    raw = lst.cpu().numpy()
    got = raw.astype(np.int64) & 0xFFFFFFFF
    for i, (nb, ts, w, _) in enumerate(pairs_ref(tbl, n, tile)):
        base = i * words
        assert got[base] == nb, (i, got[base], nb)
        assert np.array_equal(got[base + 16:base + 16 + nb], ts)
        assert np.array_equal(got[base + 16 + tpad:base + 16 + tpad + 16 * nb], w & 0xFFFFFFFF)
        # header bytes 4 .. 4 + K: the first block of every table row
        start = raw[base + 1:base + 8].view(np.uint8)[:K]
        first = np.searchsorted(ts, np.arange(K), side='left')
        assert np.array_equal(start, first.astype(np.uint8)), (i, start, first)


@pytest.mark.parametrize('seed,S,n,nb,L', [(1, 24, 400, 2, 3), (2, 64, 3000, 3, 4), (3, 16, 50, 1, 2), (4, 8, 1, 1, 2)])
def test_pair_lists_bit_exact(dev, seed, S, n, nb, L):
    from uresnet_pytorch_amd import sparse_ops as so
    c, f = cloud(seed, S, n, nb)
    geo = so.SparseGeometry(torch.from_numpy(c).to(dev), S, L)
    ref = orc.Geometry(c, f, S, L)
    for l in range(L):
        lst, tile = geo.pairs['nbr'][l]
        check_pairs(lst, tile, ref.nbr[l], ref.n[l], 27)
        if l + 1 < L:
            lst, tile = geo.pairs['chd'][l]
            check_pairs(lst, tile, ref.chd[l], ref.n[l + 1], 8)
            lst, tile = geo.pairs['up'][l]
            check_pairs(lst, tile, ref.up[l], ref.n[l], 8)


def run_conv(dev, geo_t, ref_t, ref_inv, n_out, n_in, cin, cout, pairs_f, pairs_b, flip_b, ld, with_res, seed, tbl_b=None,
             check_dw=True):
    """forward + both gradients of one gather convolution on the pair-list kernel vs the oracle; returns y, dx for
    determinism checks"""
    from uresnet_pytorch_amd import sparse_ops as so
    K = ref_t.shape[0]
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(n_in, cin)).astype(np.float32)
    W = (rng.normal(size=(K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)
    res = rng.normal(size=(n_out, cout)).astype(np.float32) if with_res else None
    dy = rng.normal(size=(n_out, cout)).astype(np.float32)
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    Wt = torch.from_numpy(W).to(dev).requires_grad_(True)
    rt = torch.from_numpy(res).to(dev) if with_res else None
    y = so.GConvFunction.apply(xt, Wt, rt, geo_t, geo_t if tbl_b is None else tbl_b, flip_b, ld, n_out, n_in, pairs_f, pairs_b)
    y.backward(torch.from_numpy(dy).to(dev))
    y_ref = orc.conv_fwd(x, W, ref_t) + (res if with_res else 0.0)
    dx_ref, dW_ref = orc.conv_bwd(x, W, ref_t, dy, ref_inv)
    ey, ex = rel(y.detach().cpu().numpy(), y_ref), rel(xt.grad.cpu().numpy(), dx_ref)
    assert ey < TOL and ex < TOL, (cin, cout, ey, ex)
    if check_dw:
        ew = rel(Wt.grad.cpu().numpy(), dW_ref)
        assert ew < TOL, (cin, cout, ew)
    return y.detach(), xt.grad.detach(), Wt.grad.detach()


SHAPES = [(16, 16), (16, 32), (32, 16), (32, 32), (48, 48), (96, 48), (64, 64), (128, 64), (64, 128), (80, 80), (160, 80),
          (80, 160), (64, 80), (192, 96), (224, 224)]


@pytest.mark.parametrize('cin,cout', SHAPES)
def test_pairs_subm_conv(dev, cin, cout):
    from uresnet_pytorch_amd import sparse_ops as so
    S = 32
    c, f = cloud(7, S, 1500, 2)
    geo = so.SparseGeometry(torch.from_numpy(c).to(dev), S, 1)
    ref = orc.Geometry(c, f, S, 1)
    n = ref.n[0]
    p = geo.pairs['nbr'][0]
    for kind in ('slabs', 'pairs'):    # both two-stage weight gradients: partial slabs of the table kernel, pair-list kernel
        so.set_deterministic_dw(True, kind)
        try:
            a = run_conv(dev, geo.nbr[0], ref.nbr[0], ref.nbr_inv[0], n, n, cin, cout, p, p, 1, geo.ld, True, cin * 100 + cout)
            b = run_conv(dev, geo.nbr[0], ref.nbr[0], ref.nbr_inv[0], n, n, cin, cout, p, p, 1, geo.ld, True, cin * 100 + cout)
        finally:
            so.set_deterministic_dw(False)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), 'forward / input gradient not bitwise reproducible'
        assert torch.equal(a[2], b[2]), 'weight gradient not bitwise reproducible (two-stage sum, no atomics): ' + kind
    # ... and the default weight-gradient kernel (dense table, fp32 atomics) against the oracle as well
    run_conv(dev, geo.nbr[0], ref.nbr[0], ref.nbr_inv[0], n, n, cin, cout, p, p, 1, geo.ld, True, cin * 100 + cout)


@pytest.mark.parametrize('cin,cout', [(16, 32), (64, 80), (48, 32), (32, 48)])
def test_pairs_strided_and_nin(dev, cin, cout):
    from uresnet_pytorch_amd import sparse_ops as so
    S = 32
    c, f = cloud(8, S, 1500, 2)
    geo = so.SparseGeometry(torch.from_numpy(c).to(dev), S, 2)
    ref = orc.Geometry(c, f, S, 2)
    nf, nc = ref.n
    # Convolution(k2, s2): fine -> coarse; its input gradient walks the fine<-parent list
    run_conv(dev, geo.chd[0], ref.chd[0], ref.chd_inv[0], nc, nf, cin, cout, geo.pairs['chd'][0], geo.pairs['up'][0], 0,
             geo.ld, False, 1, tbl_b=geo.up[0])
    # Deconvolution(k2, s2): coarse -> fine
    run_conv(dev, geo.up[0], ref.up[0], ref.up_inv[0], nf, nc, cout, cin, geo.pairs['up'][0], geo.pairs['chd'][0], 0,
             geo.ld, False, 2, tbl_b=geo.chd[0])
    # NetworkInNetwork: the identity list
    ident_ref = np.arange(nf, dtype=np.int32)[None, :]
    run_conv(dev, geo.nbr[0][13:14], ident_ref, ident_ref, nf, nf, cin, cout, so.IDENT_PAIRS, so.IDENT_PAIRS, 0, geo.ld,
             False, 3)


@pytest.fixture(scope='module')
def cfg3(dev):
    """BASELINE configs[2] geometry: seed-0 event, 512^3, 50,000 voxels, 5 levels"""
    from uresnet_pytorch_amd import sparse_ops as so
    blob = make_sparse_blob([0], 512, 50000)
    c = np.ascontiguousarray(blob['data'][:, :4].astype(np.int32))
    geo = so.SparseGeometry(torch.from_numpy(c).to(dev), 512, 5)
    ref = orc.Geometry(c, blob['data'][:, 4:5], 512, 5)
    assert geo.n == ref.n
    return geo, ref


@pytest.mark.parametrize('level', [0, 1, 2, 3, 4])
def test_conv_ops_on_cfg3_geometry(dev, cfg3, level):
    """every convolution shape the cfg3 network runs at this level (uf 16: P = 16 (level + 1)), on the real geometry:
    SubM3 P->P and 2P->P, NiN 2P->P, Convolution P->P', Deconvolution P'->P -- forward, dX, dW <= 1e-5 vs the oracle"""
    from uresnet_pytorch_amd import sparse_ops as so
    geo, ref = cfg3
    so.set_deterministic_dw(level != 4, 'slabs' if level % 2 == 0 else 'pairs')   # all three weight-gradient kernels see the real geometry
    l, P = level, 16 * (level + 1)
    n = ref.n[l]
    p = geo.pairs['nbr'][l]
    shapes = [(P, P)] + ([(2 * P, P)] if l < 4 else [])
    for cin, cout in shapes:
        run_conv(dev, geo.nbr[l], ref.nbr[l], ref.nbr_inv[l], n, n, cin, cout, p, p, 1, geo.ld, True, 10 * l + cin)
    if l < 4:
        ident_ref = np.arange(n, dtype=np.int32)[None, :]
        run_conv(dev, geo.nbr[l][13:14], ident_ref, ident_ref, n, n, 2 * P, P, so.IDENT_PAIRS, so.IDENT_PAIRS, 0, geo.ld, False, 5)
        nc, P2 = ref.n[l + 1], P + 16
        run_conv(dev, geo.chd[l], ref.chd[l], ref.chd_inv[l], nc, n, P, P2, geo.pairs['chd'][l], geo.pairs['up'][l], 0,
                 geo.ld, False, 6, tbl_b=geo.up[l])
        run_conv(dev, geo.up[l], ref.up[l], ref.up_inv[l], n, nc, P2, P, geo.pairs['up'][l], geo.pairs['chd'][l], 0,
                 geo.ld, False, 7, tbl_b=geo.chd[l])
    so.set_deterministic_dw(False)


def test_pairs_fused_epilogues_match_tile_kernel(dev):
    """The fused pieces (input BatchNorm+ReLU fold, column statistics, BatchNorm-backward reduce, strided output) of the
    pair-list kernel against the same call on the dense-table kernel (which tests/test_gpu_sparse.py pins to the oracle):
    y within 1e-6, statistics slabs (fp64 sums) within 1e-9 relative."""
    from uresnet_pytorch_amd import lib as _l, sparse_ops as so
    L = _l.load()
    S = 32
    c, f = cloud(9, S, 2000, 2)
    geo = so.SparseGeometry(torch.from_numpy(c).to(dev), S, 1)
    n = geo.n[0]
    g = torch.Generator(device='cpu').manual_seed(0)
    for cin, cout, epi in [(32, 32, 1), (64, 32, 1), (32, 64, 2), (16, 16, 2), (96, 48, 1)]:
        x = torch.randn(n, cin, generator=g).to(dev)
        wt = (torch.randn(27, cout, cin, generator=g) / (27 * cin) ** 0.5).to(dev)
        sc = (torch.rand(cin, generator=g) + 0.5).to(dev); sh = (torch.randn(cin, generator=g) * 0.3).to(dev)
        res = torch.randn(n, cout, generator=g).to(dev)
        ex = torch.randn(n, cout, generator=g).to(dev)
        esc = (torch.rand(cout, generator=g) + 0.5).to(dev); esh = (torch.randn(cout, generator=g) * 0.3).to(dev)
        emu = torch.randn(cout, generator=g).to(dev); eis = (torch.rand(cout, generator=g) + 0.5).to(dev)
        outs = []
        for use_pairs in (False, True):
            ldy = cout + 16
            y = torch.zeros(n, ldy, device=dev)
            part = torch.zeros(L.urn_gconv_part_bytes(n, cout) // 8, dtype=torch.float64, device=dev)
            a = _l.GConvArgs()
            a.x = x.data_ptr(); a.wt = wt.data_ptr(); a.tbl = geo.nbr[0].data_ptr(); a.ld = geo.ld; a.K = 27; a.flip = 0
            a.n_out = n; a.cin = cin; a.cout = cout; a.res = res.data_ptr(); a.y = y.data_ptr(); a.ldy = ldy
            a.xf_scale = sc.data_ptr(); a.xf_shift = sh.data_ptr()
            a.epilogue = epi; a.part = part.data_ptr()
            if epi == 2:
                a.e_x = ex.data_ptr(); a.e_scale = esc.data_ptr(); a.e_shift = esh.data_ptr()
                a.e_mean = emu.data_ptr(); a.e_invstd = eis.data_ptr()
            if use_pairs:
                a.pairs = geo.pairs['nbr'][0][0].data_ptr(); a.pairs_tile = geo.pairs['nbr'][0][1]
            npart = ctypes.c_int(0)
            _l.check(L.urn_gconv_fwd_ex(ctypes.byref(a), ctypes.byref(npart), _l.stream()), 'gconv_fwd_ex')
            sums = part[:npart.value * 2 * cout].reshape(npart.value, 2, cout).sum(0)
            outs.append((y.cpu().numpy(), sums.cpu().numpy()))
        assert rel(outs[1][0], outs[0][0]) < 1e-6, (cin, cout, epi)
        assert np.array_equal(outs[1][0][:, cout:], np.zeros((n, 16), np.float32)), 'wrote outside its column block'
        assert rel(outs[1][1], outs[0][1]) < 1e-6, (cin, cout, epi, outs[1][1][:, :4], outs[0][1][:, :4])


@pytest.mark.parametrize('cin,cout', [(16, 16), (32, 48), (64, 32), (80, 80), (160, 80)])
def test_pairs_fragment_ordered_weights_give_the_same_bits(dev, cin, cout):
    """urn_weight_fragments + urn_gconv_args.wt_frag: the same operands in another memory order -- identical results"""
    from uresnet_pytorch_amd import lib as _l, sparse_ops as so
    L = _l.load()
    S = 32
    c, f = cloud(11, S, 2500, 2)
    geo = so.SparseGeometry(torch.from_numpy(c).to(dev), S, 1)
    n = geo.n[0]
    g = torch.Generator(device='cpu').manual_seed(cin + cout)
    x = torch.randn(n, cin, generator=g).to(dev)
    wt = (torch.randn(27, cout, cin, generator=g) * 0.1).to(dev)
    wf = torch.empty_like(wt)
    _l.check(L.urn_weight_fragments(wt.data_ptr(), 27, cout, cin, wf.data_ptr(), _l.stream()), 'weight_fragments')
    # the layout the header states
    o, cb, kb, q, r, i = 5, cout // 16 - 1, cin // 16 - 1, 2, 7, 3
    assert float(wf.reshape(-1)[((o * (cout // 16) + cb) * (cin // 16) + kb) * 256 + (q * 16 + r) * 4 + i]) == float(wt[o, 16 * cb + r, 16 * kb + 4 * q + i])
    pl = geo.pairs['nbr'][0]
    outs = []
    L.urn_set_option(b'pairs_max_cin', 999)      # both calls on the pair-list kernel (without fragments cin > 80 takes the tile kernel)
    L.urn_set_option(b'pairs_v3', 0)             # ... and on the same loop (the strip variant needs the fragments and accumulates in another order)
    for frag in (None, wf):
        y = torch.empty(n, cout, device=dev)
        a = _l.GConvArgs()
        a.x = x.data_ptr(); a.wt = wt.data_ptr(); a.tbl = geo.nbr[0].data_ptr(); a.ld = geo.ld; a.K = 27; a.flip = 0; a.n_out = n
        a.cin = cin; a.cout = cout; a.y = y.data_ptr(); a.pairs = pl[0].data_ptr(); a.pairs_tile = pl[1]
        a.wt_frag = None if frag is None else frag.data_ptr()
        _l.check(L.urn_gconv_fwd_ex(ctypes.byref(a), None, _l.stream()), 'gconv_fwd_ex')
        outs.append(y)
    L.urn_set_option(b'pairs_max_cin', 80)
    L.urn_set_option(b'pairs_v3', 0x17E)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('prec,tol', [(1, 6e-3), (2, 8e-4)])
@pytest.mark.parametrize('cin,cout', [(16, 16), (32, 48), (64, 64), (96, 32), (128, 128), (224, 96)])
def test_pairs_reduced_precision_on_16bit_fragments(dev, cin, cout, prec, tol):
    """bf16 / fp16 operands on the pair lists (BASELINE configs[1] / configs[4] operand types): urn_weight_fragments16 +
    urn_gconv_args.wt_frag_prec; one, two or four column blocks per wave by shape.  Against the fp64 product of the
    unrounded operands within the operand type's bound (same bounds as the tile kernel's test in test_gpu_sparse.py),
    and really on the pair-list kernel: with the lists withheld the call takes the 2-D tile kernel and differs in the last
    bits (another summation order), with fp32 fragments it must fall back to it and give the tile kernel's bits."""
    from uresnet_pytorch_amd import lib as _l, sparse_ops as so
    L = _l.load()
    S = 32
    c, f = cloud(13, S, 2500, 2)
    geo = so.SparseGeometry(torch.from_numpy(c).to(dev), S, 1)
    n = geo.n[0]
    g = torch.Generator(device='cpu').manual_seed(cin * 7 + cout)
    x = torch.randn(n, cin, generator=g).to(dev)
    wt = (torch.randn(27, cout, cin, generator=g) * 0.1).to(dev)
    wf16 = torch.empty(27 * cout * cin, dtype=torch.int16, device=dev)
    _l.check(L.urn_weight_fragments16(wt.data_ptr(), 27, cout, cin, prec, wf16.data_ptr(), _l.stream()), 'weight_fragments16')
    # the layout the header states, with 16-bit elements: 8 bytes per lane and block; an even number of 16-channel groups
    # pairs the blocks (kb, kb + 1) in one kilobyte, a lane's 16 bytes = [its 8 of kb | its 8 of kb + 1]
    kbn = cin // 16
    for o, cb, kb, q, r, i in ((5, cout // 16 - 1, kbn - 1, 2, 7, 3), (26, 0, 0, 3, 15, 0), (11, cout // 32, kbn // 2, 1, 0, 2)):
        blk, lane = (o * (cout // 16) + cb) * kbn + kb, q * 16 + r
        slot = blk * 64 + lane if kbn % 2 else ((blk & ~1) << 6) + 2 * lane + (blk & 1)
        got = wf16.view(torch.bfloat16 if prec == 1 else torch.float16)[slot * 4 + i]
        assert float(got) == float(wt[o, 16 * cb + r, 16 * kb + 4 * q + i].to(got.dtype))
    wf32 = torch.empty_like(wt)
    _l.check(L.urn_weight_fragments(wt.data_ptr(), 27, cout, cin, wf32.data_ptr(), _l.stream()), 'weight_fragments')
    pl = geo.pairs['nbr'][0]

    def call(frag, frag_prec, with_lists):
        y = torch.empty(n, cout, device=dev)
        a = _l.GConvArgs()
        a.x = x.data_ptr(); a.wt = wt.data_ptr(); a.tbl = geo.nbr[0].data_ptr(); a.ld = geo.ld; a.K = 27; a.flip = 0; a.n_out = n
        a.cin = cin; a.cout = cout; a.y = y.data_ptr(); a.precision = prec + 1
        if with_lists:
            a.pairs = pl[0].data_ptr(); a.pairs_tile = pl[1]
        a.wt_frag = None if frag is None else frag.data_ptr(); a.wt_frag_prec = frag_prec
        _l.check(L.urn_gconv_fwd_ex(ctypes.byref(a), None, _l.stream()), 'gconv_fwd_ex')
        return y
    y_pairs = call(wf16, prec, True)
    y_tile = call(None, 0, False)
    y_fallback = call(wf32, 0, True)
    nbr = geo.nbr[0][:, :n].cpu().numpy()
    xd, wd = x.double().cpu().numpy(), wt.double().cpu().numpy()
    ref = np.zeros((n, cout))
    for k in range(27):
        m = nbr[k] >= 0
        ref[m] += xd[nbr[k][m]] @ wd[k].T
    e = rel(y_pairs.double().cpu().numpy(), ref)
    assert 1e-6 < e < tol, e
    assert rel(y_tile.double().cpu().numpy(), ref) < tol
    assert torch.equal(y_fallback, y_tile)
    assert not torch.equal(y_pairs, y_tile)


@pytest.mark.parametrize('cin,cout', [(16, 16), (32, 32), (48, 48), (64, 64), (80, 80), (96, 48), (128, 64), (32, 16), (16, 32)])
def test_pairs_strip_variant(dev, cin, cout):
    """urn_set_option("pairs_v3", bits), the default loop of the pair-list kernel for one-chunk inputs: the pair words of a wave's
    share staged in a wave-private LDS strip, the next offset's weight block requested with the next rows, the old slab values
    as the MFMA's C operand.  Another summation order than the per-block index loop (old + (acc + acc2) there), so: within
    1e-6 of it, within 1e-5 of the fp64 product, bitwise reproducible from run to run, with and without the folded input
    BatchNorm, on 64-row tiles with a share per wave of one to many blocks (pairs_split 1 / 8)."""
    from uresnet_pytorch_amd import lib as _l, sparse_ops as so
    L = _l.load()
    S = 32
    c, f = cloud(19, S, 2500, 2)
    geo = so.SparseGeometry(torch.from_numpy(c).to(dev), S, 1)
    n = geo.n[0]
    g = torch.Generator(device='cpu').manual_seed(cin * 5 + cout)
    x = torch.randn(n, cin, generator=g).to(dev)
    wt = (torch.randn(27, cout, cin, generator=g) * 0.1).to(dev)
    sc = (torch.rand(cin, generator=g) + 0.5).to(dev); sh = (torch.randn(cin, generator=g) * 0.1).to(dev)
    wf = torch.empty_like(wt)
    _l.check(L.urn_weight_fragments(wt.data_ptr(), 27, cout, cin, wf.data_ptr(), _l.stream()), 'weight_fragments')
    pl = geo.pairs['nbr'][0]

    def call(xf):
        y = torch.empty(n, cout, device=dev)
        a = _l.GConvArgs()
        a.x = x.data_ptr(); a.wt = wt.data_ptr(); a.tbl = geo.nbr[0].data_ptr(); a.ld = geo.ld; a.K = 27; a.flip = 0; a.n_out = n
        a.cin = cin; a.cout = cout; a.y = y.data_ptr(); a.pairs = pl[0].data_ptr(); a.pairs_tile = pl[1]; a.wt_frag = wf.data_ptr()
        if xf:
            a.xf_scale = sc.data_ptr(); a.xf_shift = sh.data_ptr()
        _l.check(L.urn_gconv_fwd_ex(ctypes.byref(a), None, _l.stream()), 'gconv_fwd_ex')
        return y
    nbr = geo.nbr[0][:, :n].cpu().numpy()
    wd = wt.double().cpu().numpy()
    refs = {}
    for xf in (False, True):
        xd = (torch.relu(x.double() * sc.double() + sh.double()) if xf else x.double()).cpu().numpy()
        ref = np.zeros((n, cout))
        for k in range(27):
            m = nbr[k] >= 0
            ref[m] += xd[nbr[k][m]] @ wd[k].T
        refs[xf] = ref
    try:
        L.urn_set_option(b'pairs_max_cin', 999)
        for split in (0, 1, 8):
            L.urn_set_option(b'pairs_split', split)
            for xf in (False, True):
                L.urn_set_option(b'pairs_v3', 0)
                y_loop = call(xf)
                L.urn_set_option(b'pairs_v3', 0x17E)
                y_a, y_b = call(xf), call(xf)
                assert torch.equal(y_a, y_b), 'strip variant not bitwise reproducible'
                assert rel(y_a.cpu().numpy(), refs[xf]) < TOL, (cin, cout, split, xf, rel(y_a.cpu().numpy(), refs[xf]))
                assert rel(y_a.cpu().numpy(), y_loop.cpu().numpy()) < 1e-6, (cin, cout, split, xf)
    finally:
        L.urn_set_option(b'pairs_split', 0); L.urn_set_option(b'pairs_v3', 0x17E); L.urn_set_option(b'pairs_max_cin', 80)
