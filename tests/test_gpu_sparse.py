"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.
Integer results are compared bit-exact; fp32 results within 1e-5 relative (norm-wise),
the tolerance BASELINE.json's north_star states."""
from types import SimpleNamespace

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


def cloud(seed, S, n, nbatch, dup):
    rng = np.random.default_rng(seed)
    cs = []
    for b in range(nbatch):
        c, _, _ = generate_event(seed * 10 + b, S, n)
        cs.append(np.concatenate([c, np.full((len(c), 1), b, np.int32)], 1))
    c = np.concatenate(cs, 0)
    if dup:
        c = np.concatenate([c, c[rng.integers(0, len(c), dup)]], 0)
        c = c[rng.permutation(len(c))]
    f = rng.normal(size=(len(c), 1)).astype(np.float32)
    return np.ascontiguousarray(c, np.int32), f


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    from uresnet_pytorch_amd import lib
    lib.load()
    return torch.device('cuda:0')


def check_geometry(c, f, S, L, dev):
    """Both routes of the integer phase against the oracle: all levels at once from the input rows
    (urn_sites_build_levels, the default) and level by level (urn_sites_build + urn_level_down_tables)."""
    from uresnet_pytorch_amd import sparse_ops as so
    ref = orc.Geometry(c, f, S, L)
    for per_level in (True, False):
        geo = so.SparseGeometry(torch.from_numpy(c).to(dev), S, L, per_level=per_level)
        assert geo.n == ref.n
        assert np.array_equal(geo.row2site[:len(c)].cpu().numpy(), ref.row2site)
        for l in range(L):
            assert np.array_equal(geo.export_coords(l), ref.coords[l]), 'site coords level %d' % l
            assert np.array_equal(geo.export_nbr(l), ref.nbr[l]), 'subm table level %d' % l
            assert geo.rules[l] == ref.R[l]
            # canonical (offset, in, out) triples, sorted
            assert np.array_equal(orc.canonical_triples(geo.export_nbr(l)), orc.canonical_triples(ref.nbr[l]))
            if l + 1 < L:
                n, nc = geo.n[l], geo.n[l + 1]
                assert np.array_equal(geo.parent[l][:n].cpu().numpy(), ref.parent[l])
                assert np.array_equal(geo.off[l][:n].cpu().numpy(), ref.off[l])
                assert np.array_equal(geo.chd[l][:, :nc].cpu().numpy(), ref.chd[l])
                assert np.array_equal(geo.up[l][:, :n].cpu().numpy(), ref.up[l])
    sf = so.input_features(geo, torch.from_numpy(f).to(dev)).cpu().numpy()
    assert np.array_equal(sf, ref.feats)       # fp64 accumulation, one rounding: bit-exact
    return geo, ref


@pytest.mark.parametrize('seed,S,n,nb,dup,L', [(1, 24, 400, 2, 37, 3), (2, 64, 3000, 3, 0, 4), (3, 16, 50, 1, 200, 2)])
def test_integer_phase_bit_exact(dev, seed, S, n, nb, dup, L):
    c, f = cloud(seed, S, n, nb, dup)
    check_geometry(c, f, S, L, dev)


def test_integer_phase_edges(dev):
    # volume border (no wrap-around), a single site, many duplicates of one site
    S = 8
    c = np.array([[0, 0, 0, 0], [7, 7, 7, 0], [0, 0, 1, 0], [7, 7, 6, 1], [7, 7, 6, 1], [7, 7, 6, 1]], np.int32)
    f = np.arange(len(c), dtype=np.float32)[:, None]
    check_geometry(c, f, S, 3, dev)
    c1 = np.array([[3, 4, 5, 0]], np.int32)
    check_geometry(c1, np.ones((1, 1), np.float32), S, 2, dev)
    # more levels than the multi-level entry points take (8): the per-level calls serve them
    c9, f9 = cloud(5, 512, 300, 1, 3)
    check_geometry(c9, f9, 512, 9, dev)


def test_integer_phase_full_size(dev):
    """BASELINE cfg3 size: 512^3, 50k active voxels, 5 levels -- bit-exact against the oracle."""
    blob = make_sparse_blob([0], 512, 50000)
    c = blob['data'][:, :4].astype(np.int32)
    f = blob['data'][:, 4:5].copy()
    geo, ref = check_geometry(c, f, 512, 5, dev)
    assert geo.n[0] == 50000


CONV_SHAPES = [(1, 16), (16, 16), (16, 32), (32, 48), (80, 80), (128, 64), (96, 48),
               (384, 192), (224, 224), (352, 176)]   # cfg5 widths: input channels walked in chunks


@pytest.mark.parametrize('cin,cout', CONV_SHAPES)
def test_subm_conv_fwd_bwd(dev, cin, cout):
    from uresnet_pytorch_amd import sparse_ops as so
    S = 32
    c, f = cloud(7, S, 1500, 2, 0)
    geo = so.SparseGeometry(torch.from_numpy(c).to(dev), S, 1)
    ref = orc.Geometry(c, f, S, 1)
    n = ref.n[0]
    rng = np.random.default_rng(cin * 100 + cout)
    x = rng.normal(size=(n, cin)).astype(np.float32)
    W = (rng.normal(size=(27, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    res = rng.normal(size=(n, cout)).astype(np.float32)
    dy = rng.normal(size=(n, cout)).astype(np.float32)
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    Wt = torch.from_numpy(W).to(dev).requires_grad_(True)
    rt = torch.from_numpy(res).to(dev).requires_grad_(True)
    y = so.GConvFunction.apply(xt, Wt, rt, geo.nbr[0], geo.nbr[0], 1, geo.ld, n, n)
    y.backward(torch.from_numpy(dy).to(dev))
    y_ref = orc.conv_fwd(x, W, ref.nbr[0]) + res
    dx_ref, dW_ref = orc.conv_bwd(x, W, ref.nbr[0], dy, ref.nbr_inv[0])
    assert rel(y.detach().cpu().numpy(), y_ref) < TOL
    assert rel(xt.grad.cpu().numpy(), dx_ref) < TOL
    assert rel(Wt.grad.cpu().numpy(), dW_ref) < TOL
    assert np.array_equal(rt.grad.cpu().numpy(), dy)


@pytest.mark.parametrize('cin,cout', [(16, 32), (64, 80), (48, 32)])
def test_strided_conv_deconv_fwd_bwd(dev, cin, cout):
    from uresnet_pytorch_amd import sparse_ops as so
    S = 32
    c, f = cloud(8, S, 1500, 2, 0)
    geo = so.SparseGeometry(torch.from_numpy(c).to(dev), S, 2)
    ref = orc.Geometry(c, f, S, 2)
    nf, nc = ref.n
    rng = np.random.default_rng(cin + cout)
    # down: fine -> coarse
    x = rng.normal(size=(nf, cin)).astype(np.float32)
    W = (rng.normal(size=(8, cin, cout)) / np.sqrt(8 * cin)).astype(np.float32)
    dy = rng.normal(size=(nc, cout)).astype(np.float32)
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    Wt = torch.from_numpy(W).to(dev).requires_grad_(True)
    y = so.GConvFunction.apply(xt, Wt, None, geo.chd[0], geo.up[0], 0, geo.ld, nc, nf)
    y.backward(torch.from_numpy(dy).to(dev))
    dx_ref, dW_ref = orc.conv_bwd(x, W, ref.chd[0], dy, ref.chd_inv[0])
    assert rel(y.detach().cpu().numpy(), orc.conv_fwd(x, W, ref.chd[0])) < TOL
    assert rel(xt.grad.cpu().numpy(), dx_ref) < TOL and rel(Wt.grad.cpu().numpy(), dW_ref) < TOL
    # up: coarse -> fine
    z = rng.normal(size=(nc, cout)).astype(np.float32)
    Wu = (rng.normal(size=(8, cout, cin)) / np.sqrt(8 * cout)).astype(np.float32)
    du = rng.normal(size=(nf, cin)).astype(np.float32)
    zt = torch.from_numpy(z).to(dev).requires_grad_(True)
    Wut = torch.from_numpy(Wu).to(dev).requires_grad_(True)
    u = so.GConvFunction.apply(zt, Wut, None, geo.up[0], geo.chd[0], 0, geo.ld, nf, nc)
    u.backward(torch.from_numpy(du).to(dev))
    dz_ref, dWu_ref = orc.conv_bwd(z, Wu, ref.up[0], du, ref.up_inv[0])
    assert rel(u.detach().cpu().numpy(), orc.conv_fwd(z, Wu, ref.up[0])) < TOL
    assert rel(zt.grad.cpu().numpy(), dz_ref) < TOL and rel(Wut.grad.cpu().numpy(), dWu_ref) < TOL


@pytest.mark.parametrize('n,c', [(1000, 16), (4097, 48), (300, 160), (1, 16), (50000, 80)])
def test_bn_relu_fwd_bwd(dev, n, c):
    from uresnet_pytorch_amd import sparse_ops as so
    rng = np.random.default_rng(n + c)
    x = (rng.normal(size=(n, c)) * 2 + 0.5).astype(np.float32)
    g = (1 + 0.1 * rng.normal(size=c)).astype(np.float32)
    b = (0.1 * rng.normal(size=c)).astype(np.float32)
    dy = rng.normal(size=(n, c)).astype(np.float32)
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    gt = torch.from_numpy(g).to(dev).requires_grad_(True)
    bt = torch.from_numpy(b).to(dev).requires_grad_(True)
    rm = torch.zeros(c, device=dev); rv = torch.ones(c, device=dev)
    y = so.BNReLUFunction.apply(xt, gt, bt, rm, rv, orc.BN_EPS, 0.9, True, True)
    y.backward(torch.from_numpy(dy).to(dev))
    y_ref, mean, invstd = orc.bn_relu_fwd(x, g, b, True)
    dx_ref, dg_ref, db_ref = orc.bn_relu_bwd(x, y_ref, dy, g, mean, invstd, True)
    assert rel(y.detach().cpu().numpy(), y_ref) < TOL
    if n > 1:
        assert rel(xt.grad.cpu().numpy(), dx_ref) < 5 * TOL
        assert rel(gt.grad.cpu().numpy(), dg_ref) < 5 * TOL
    assert rel(bt.grad.cpu().numpy(), db_ref) < TOL
    assert rel(rm.cpu().numpy(), 0.1 * mean) < 1e-4


def make_model(flags, P, dev):
    from uresnet_pytorch_amd.models import SparseUResNet
    m = SparseUResNet(flags)
    sd = m.state_dict()
    for k, v in P.items():
        assert tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v)
    m.load_state_dict(sd)
    return m.to(dev).train()


def run_network_parity(dev, S, m, L, nc, pc, lab, tol_fwd, tol_grad):
    from uresnet_pytorch_amd.models import SparseSegmentationLoss
    flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=m, URESNET_NUM_STRIDES=L, SPATIAL_SIZE=S, NUM_CLASS=nc)
    P = orc.init_params(m, L, nc, seed=1)
    net = make_model(flags, P, dev)
    data = torch.from_numpy(pc).to(dev)
    label = torch.from_numpy(lab).to(dev)
    out = net(data)
    loss, acc = SparseSegmentationLoss(flags)(out, [data], [label], None)
    loss.backward()
    ref = orc.SparseUResNetOracle(P, m, L, nc, S)
    logits_ref = ref.forward(pc)
    loss_ref, acc_ref, dl = orc.segmentation_loss(logits_ref, pc, lab)
    G, _ = ref.backward(dl)
    e_fwd = rel(out[0].detach().cpu().numpy(), logits_ref)
    assert e_fwd < tol_fwd, e_fwd
    assert abs(loss.item() - loss_ref) < 1e-5 * max(1.0, abs(loss_ref))
    assert abs(acc - acc_ref) < 1e-6
    worst = 0.0
    for k, p in net.named_parameters():
        e = rel(p.grad.cpu().numpy(), G[k])
        worst = max(worst, e)
        assert e < tol_grad, (k, e)
    return e_fwd, worst


def test_network_small(dev):
    S, m, L, nc = 32, 16, 3, 5
    c, f = cloud(6, S, 800, 2, 11)
    pc = np.concatenate([c.astype(np.float32), f], 1)
    lab = np.random.default_rng(3).integers(0, nc, size=(len(pc), 1)).astype(np.float32)
    run_network_parity(dev, S, m, L, nc, pc, lab, TOL, 5 * TOL)


def gpu_relu_masks(net, dev):
    """ReLU masks of the executor's last training forward, keyed by the BatchNorm's parameter prefix: mask = (x * scale +
    shift > 0) with the executor's own folded scale / shift and BatchNorm input (urn_net_bn_export).  Evaluated in fp64:
    the product of two fp32 values is exact there and rounding never crosses zero, so the sign equals that of the
    kernel's fmaf(x, scale, shift)."""
    import ctypes
    from uresnet_pytorch_amd import lib as _l
    L = _l.load()
    ex = net._executor
    h = ex.slots[0].handle
    names = {id(p): k for k, p in net.named_parameters()}
    by_off = {o: names[id(p)] for p, o in zip(ex.params, ex.offsets)}
    masks = {}
    for i in range(L.urn_net_num_bn(h)):
        w = ctypes.c_int64(); rows = ctypes.c_int64(); c = ctypes.c_int()
        _l.check(L.urn_net_bn_info(h, i, ctypes.byref(w), ctypes.byref(rows), ctypes.byref(c)))
        x = torch.empty((rows.value, c.value), dtype=torch.float32, device=dev)
        sc = torch.empty(c.value, dtype=torch.float32, device=dev); sh = torch.empty_like(sc)
        _l.check(L.urn_net_bn_export(h, i, x.data_ptr(), sc.data_ptr(), sh.data_ptr(), _l.stream()))
        pre = x.double() * sc.double() + sh.double()
        name = by_off[w.value]
        assert name.endswith('.weight')
        masks[name[:-len('.weight')]] = (pre > 0).cpu().numpy()
    return masks


def run_network_parity_pinned(dev, S, m, L, nc, pc, lab, tol_fwd, tol_grad, seeds=None):
    """Whole network through the EXECUTOR (the product path of a training step) against the oracle with its ReLU masks
    pinned to the GPU's.  Returns (logits error, worst gradient error, number of mask entries that differ from the
    oracle's own)."""
    from uresnet_pytorch_amd.models import SparseSegmentationLoss
    flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=m, URESNET_NUM_STRIDES=L, SPATIAL_SIZE=S, NUM_CLASS=nc)
    P = orc.init_params(m, L, nc, seed=1)
    net = make_model(flags, P, dev)
    data = torch.from_numpy(pc).to(dev)
    label = torch.from_numpy(lab).to(dev)
    out = net(data)
    loss, acc = SparseSegmentationLoss(flags)(out, [data], [label], None)
    loss.backward()
    masks = gpu_relu_masks(net, dev)
    assert len(masks) == sum(1 for k in P if k.endswith('.bias') and not k.startswith('linear')), 'one mask per BatchNorm'
    # the oracle on its OWN ReLU branches first: the GPU logits are compared with this unpinned run as well, and the GPU's
    # masks may differ from the oracle's own only in a handful of entries per layer, each with a pre-activation within
    # rounding of zero -- a wrong scale / shift export that flipped thousands would otherwise be copied into the reference
    free = orc.SparseUResNetOracle(P, m, L, nc, S)
    free.keep_acts = True
    logits_free = free.forward(pc)
    flips, worst_pre = 0, 0.0
    for k in masks:
        d = (free.acts[k] > 0) != masks[k]
        n_k = int(d.sum())
        flips += n_k
        assert n_k <= 8, (k, n_k)
        if n_k:
            pre = free.pre[k].astype(np.float64)
            r_k = float(np.abs(pre[d]).max() / max(np.sqrt((pre ** 2).mean()), 1e-30))
            worst_pre = max(worst_pre, r_k)
            assert r_k < 1e-4, (k, n_k, r_k)     # |pre-activation| / rms of the layer at every flipped entry
    assert flips <= 64, flips
    e_free = rel(out[0].detach().cpu().numpy(), logits_free)
    assert e_free < tol_fwd, ('unpinned logits', e_free)
    loss_free = orc.segmentation_loss(logits_free, pc, lab)[0]
    assert abs(loss.item() - loss_free) < 1e-5 * max(1.0, abs(loss_free))
    print('pinned parity: %d ReLU branches differ from the oracle\'s own (|pre|/rms <= %.1e), unpinned logits %.2e' % (flips, worst_pre, e_free))
    ref = orc.SparseUResNetOracle(P, m, L, nc, S)
    ref.masks = masks
    logits_ref = ref.forward(pc)
    loss_ref, acc_ref, dl = orc.segmentation_loss(logits_ref, pc, lab)
    G, _ = ref.backward(dl)
    e_fwd = rel(out[0].detach().cpu().numpy(), logits_ref)
    assert e_fwd < tol_fwd, e_fwd
    assert abs(loss.item() - loss_ref) < 1e-5 * max(1.0, abs(loss_ref))
    worst = 0.0
    for k, p in net.named_parameters():
        e = rel(p.grad.cpu().numpy(), G[k])
        worst = max(worst, e)
        assert e < tol_grad, (k, e, flips)
    return e_fwd, worst, flips


def test_network_cfg3_full_size(dev):
    """BASELINE configs[2]: -dd 3 -ss 512 ~50k voxels -nc 5 -uf 16 -uns 5, fp32, one training step through the executor
    vs the oracle: logits and loss to 1e-5, EVERY parameter gradient to 1e-5 (norm-wise; measured 1.2e-6).

    The oracle's ReLU masks are pinned to the GPU's (gpu_relu_masks): with 45 BatchNorm+ReLU layers over ~1e6 elements
    each, a handful of pre-activations lie within fp32 rounding of zero and would take different branches in any two
    evaluation orders (counted and printed: `flips`); one such element moves a per-channel gradient sum by ~1e-3
    relative, which says nothing about the arithmetic.  With the masks pinned the comparison is arithmetic only."""
    blob = make_sparse_blob([0], 512, 50000)
    e_fwd, e_grad, flips = run_network_parity_pinned(dev, 512, 16, 5, 5, blob["data"], blob["label"], TOL, TOL)
    print('cfg3 parity: logits rel err %.2e, worst grad rel err %.2e, %d mask entries pinned' % (e_fwd, e_grad, flips))


def test_network_cfg4_two_events_per_gpu(dev):
    """BASELINE configs[3] per-GPU workload (-bs 16 over 8 GPUs = two 50k-voxel events in one point cloud, batch ids 0/1):
    executor vs oracle with pinned masks -- logits 1e-5, loss (sum of the two per-event means), accuracy, gradients 1e-5 (measured 1.2e-6)."""
    blob = make_sparse_blob([3, 4], 512, 50000)
    assert set(np.unique(blob['data'][:, 3]).tolist()) == {0.0, 1.0}
    e_fwd, e_grad, flips = run_network_parity_pinned(dev, 512, 16, 5, 5, blob["data"], blob["label"], TOL, TOL)
    print('cfg4 per-GPU parity: logits rel err %.2e, worst grad rel err %.2e, %d mask entries pinned' % (e_fwd, e_grad, flips))


def test_executor_matches_per_layer_path(dev):
    """The C++ whole-network executor against the per-layer autograd path.
    Unfused executor (URN_NET_UNFUSED): the same kernels in the same order -> identical logits (bitwise),
    gradients equal up to the fp32 atomics order of the weight gradient.
    Fused executor (default; BatchNorm folded into the convolutions): same maths, different rounding
    (relu(x*scale+shift) instead of relu((x-mean)*a+beta)) -> 1e-5 relative."""
    from uresnet_pytorch_amd.models import SparseSegmentationLoss
    S, m, L, nc = 64, 16, 4, 5
    blob = make_sparse_blob([5, 6], S, 3000)
    flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=m, URESNET_NUM_STRIDES=L, SPATIAL_SIZE=S, NUM_CLASS=nc)
    P = orc.init_params(m, L, nc, seed=2)
    data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
    res = {}
    for name, use_ex, fl in (('fused', True, 0), ('slab', True, 4), ('unfused', True, 1), ('layers', False, 0)):
        net = make_model(flags, P, dev)
        net.use_executor = use_ex
        net.executor_flags = fl
        out = net(data)
        loss, _ = SparseSegmentationLoss(flags)(out, [data], [label], None)
        loss.backward()
        res[name] = (out[0].detach().cpu().numpy(),
                     {k: p.grad.detach().cpu().numpy().copy() for k, p in net.named_parameters()},
                     {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items() if 'running' in k})
    assert np.array_equal(res['unfused'][0], res['layers'][0])
    for k in res['layers'][1]:
        assert rel(res['unfused'][1][k], res['layers'][1][k]) < 1e-5, k
    for k in res['layers'][2]:
        assert np.array_equal(res['unfused'][2][k], res['layers'][2][k]), k
    assert rel(res['fused'][0], res['layers'][0]) < 1e-5
    for k in res['layers'][1]:
        assert rel(res['fused'][1][k], res['layers'][1][k]) < 5e-5, k
    for k in res['layers'][2]:
        assert rel(res['fused'][2][k], res['layers'][2][k]) < 1e-6, k
    # URN_NET_SLAB_STATS (per-workgroup slabs + finalize launches, fixed summation order) against the default
    # accumulated statistics: same fp64 sums up to their last bits
    assert rel(res['slab'][0], res['layers'][0]) < 1e-5 and rel(res['slab'][0], res['fused'][0]) < 2e-6
    for k in res['layers'][1]:
        assert rel(res['slab'][1][k], res['layers'][1][k]) < 5e-5, k
    for k in res['layers'][2]:
        assert rel(res['slab'][2][k], res['fused'][2][k]) < 1e-6, k


def test_tail_kernels_and_fused_head(dev):
    """urn_tail_fwd / urn_tail_bwd (last BatchNormReLU + OutputLayer + Linear as one kernel each way) against torch in fp64 on
    rows that share sites, and the model with the head inside the executor (urn_net_set_head, default) against the route with
    separate OutputLayer / Linear launches: logits within 1e-6, every parameter gradient within 2e-6."""
    import ctypes
    from uresnet_pytorch_amd import lib as _l
    from uresnet_pytorch_amd.models import SparseSegmentationLoss
    L = _l.load()
    g = torch.Generator(device='cpu').manual_seed(3)
    n0, N, m, nc, SLOTS = 3000, 3500, 16, 5, 8
    x = torch.randn(n0, m, generator=g).to(dev)
    r2s = torch.cat([torch.randperm(n0, generator=g), torch.randint(0, n0, (N - n0,), generator=g)]).to(torch.int32).to(dev)   # 500 rows share sites
    W = (torch.randn(nc, m, generator=g) * 0.3).to(dev); b = torch.randn(nc, generator=g).to(dev)
    gamma = (torch.rand(m, generator=g) + 0.5).to(dev); beta = (torch.randn(m, generator=g) * 0.2).to(dev)
    sums = torch.zeros(SLOTS, 2, m, dtype=torch.float64, device=dev)
    sums[1, 0] = x.double().sum(0); sums[6, 1] = (x.double() ** 2).sum(0)
    mean, invstd, scale, shift = [torch.empty(m, device=dev) for _ in range(4)]
    rm = torch.zeros(m, device=dev); rv = torch.ones(m, device=dev)
    logits = torch.empty(N, nc, device=dev)
    _l.check(L.urn_tail_fwd(x.data_ptr(), r2s.data_ptr(), N, m, nc, W.data_ptr(), b.data_ptr(), sums.data_ptr(), SLOTS, n0, 1e-4,
                            gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                            rm.data_ptr(), rv.data_ptr(), 0.9, logits.data_ptr(), _l.stream()), 'tail_fwd')
    xd = x.double()
    mu = xd.mean(0); var = xd.var(0, unbiased=False); isd = 1.0 / torch.sqrt(var + 1e-4)
    y = torch.relu((xd - mu) * isd * gamma.double() + beta.double())
    ref = y[r2s.long()] @ W.double().T + b.double()
    assert rel(logits.cpu().numpy(), ref.cpu().numpy()) < 1e-6
    assert rel(mean.cpu().numpy(), mu.cpu().numpy()) < 1e-6 and rel(invstd.cpu().numpy(), isd.cpu().numpy()) < 1e-6
    assert rel(rm.cpu().numpy(), (0.1 * mu).cpu().numpy()) < 1e-6 and rel(rv.cpu().numpy(), (0.9 + 0.1 * var).cpu().numpy()) < 1e-6
    dl_all = torch.randn(N, nc, generator=g).to(dev)
    xh = (xd - mean.double()) * invstd.double()
    mask = ((x * scale + shift) > 0).double()
    y32 = torch.relu(x * scale + shift).double()
    # rows that share sites (gradients added onto zeros), then one row per site (n_sites == n: stored, gsite not zeroed)
    for rows in (N, n0):
        dl = dl_all[:rows].contiguous(); rs = r2s[:rows].contiguous()
        gsite = torch.zeros(n0, m, device=dev) if rows != n0 else torch.full((n0, m), float('nan'), device=dev)
        dW = torch.zeros(nc, m, device=dev); db = torch.zeros(nc, device=dev)
        part = torch.zeros(SLOTS, 2, m, dtype=torch.float64, device=dev)
        _l.check(L.urn_tail_bwd(dl.data_ptr(), x.data_ptr(), rs.data_ptr(), rows, m, nc, W.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                mean.data_ptr(), invstd.data_ptr(), gsite.data_ptr(), n0, dW.data_ptr(), db.data_ptr(), part.data_ptr(),
                                SLOTS, _l.stream()), 'tail_bwd')
        g_rows = (dl.double() @ W.double()) * mask[rs.long()]
        g_ref = torch.zeros(n0, m, dtype=torch.float64, device=dev).index_add_(0, rs.long(), g_rows)
        assert rel(gsite.cpu().numpy(), g_ref.cpu().numpy()) < 1e-6
        assert rel(dW.cpu().numpy(), (dl.double().T @ y32[rs.long()]).cpu().numpy()) < 1e-5
        assert rel(db.cpu().numpy(), dl.double().sum(0).cpu().numpy()) < 1e-5
        s = part.sum(0)
        assert rel(s[0].cpu().numpy(), g_ref.sum(0).cpu().numpy()) < 1e-6 and rel(s[1].cpu().numpy(), (g_ref * xh).sum(0).cpu().numpy()) < 1e-6
    # the model with and without the head inside the executor
    S, mm, Lv = 64, 16, 3
    blob = make_sparse_blob([11, 12], S, 2500)
    flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=mm, URESNET_NUM_STRIDES=Lv, SPATIAL_SIZE=S, NUM_CLASS=nc)
    P = orc.init_params(mm, Lv, nc, seed=6)
    data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
    data = torch.cat([data, data[:300]])            # duplicated rows: several input rows per site (InputLayer mode 3)
    label = torch.cat([label, label[:300]])
    res = {}
    for fuse in (True, False):
        net = make_model(flags, P, dev)
        net.fuse_head = fuse
        out = net(data)
        loss, _ = SparseSegmentationLoss(flags)(out, [data], [label], None)
        loss.backward()
        res[fuse] = (out[0].detach().cpu().numpy(), {k: p.grad.detach().cpu().numpy().copy() for k, p in net.named_parameters()},
                     {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items() if 'running' in k})
    assert rel(res[True][0], res[False][0]) < 1e-6
    for k in res[False][1]:
        assert rel(res[True][1][k], res[False][1][k]) < 2e-6, k
    for k in res[False][2]:
        assert rel(res[True][2][k], res[False][2][k]) < 1e-6, k
    # inference without gradients (running statistics) takes the fused tail as well
    net = make_model(flags, P, dev).eval()
    with torch.no_grad():
        a = net(data)[0]
        net.fuse_head = False
        bb = net(data)[0]
    assert rel(a.cpu().numpy(), bb.cpu().numpy()) < 1e-6


def test_backward_callback_suffix_is_final(dev):
    """urn_net_backward_cb (the hook of parallel.OverlappedAllReduce): when the call-back fires, every kernel that writes
    [urn_net_suffix_offset, end) of the flat gradient buffer -- bottom level, decoder, last BatchNorm, head -- has been
    enqueued, on the caller's stream or on a side stream ordered behind it.  A copy of the suffix issued from inside the
    call-back BEHIND THE SIDE STREAM'S TAIL (where the collective goes) must therefore equal the suffix after the whole
    backward pass bit for bit, the prefix must still change afterwards, and the offset must split the parameters where the
    module tree says: everything from the deepest level's first block on."""
    from uresnet_pytorch_amd import lib as _l, parallel
    from uresnet_pytorch_amd.models import SparseSegmentationLoss
    S, m, Lv, nc = 64, 16, 4, 5
    blob = make_sparse_blob([21, 22], S, 4000)
    flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=m, URESNET_NUM_STRIDES=Lv, SPATIAL_SIZE=S, NUM_CLASS=nc)
    P = orc.init_params(m, Lv, nc, seed=8)
    net = make_model(flags, P, dev)
    grads = parallel.FlatGradients(net)
    data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
    out = net(data)
    loss, _ = SparseSegmentationLoss(flags)(out, [data], [label], None)
    seen = {}

    def hook(offset, side_ptr):
        assert 'offset' not in seen, 'the call-back fired twice'
        seen['offset'] = offset
        assert side_ptr, 'the executor has a side stream on this box'
        with torch.cuda.stream(torch.cuda.ExternalStream(side_ptr, device=dev)):
            seen['suffix'] = grads.flat[offset:].clone()
            seen['prefix'] = grads.flat[:offset].clone()
    grads.zero()
    net._executor.suffix_hook = hook
    loss.backward()
    net._executor.suffix_hook = None
    torch.cuda.synchronize()
    off = seen['offset']
    assert torch.equal(seen['suffix'], grads.flat[off:]), 'something wrote into the suffix after the call-back'
    assert not torch.equal(seen['prefix'], grads.flat[:off]), 'the encoder gradients were already complete?'
    assert float(grads.flat[off:].abs().sum()) > 0
    # where the offset falls in the module tree: the first parameter at or behind it belongs to the deepest level's first block
    names = [k for k, _ in net.named_parameters()]
    sizes = [p.numel() for _, p in net.named_parameters()]
    starts = np.cumsum([0] + sizes[:-1])
    i_first = int(np.searchsorted(starts, off))
    assert int(starts[i_first]) == off
    deepest = 'sparseModel.2' + '.4.1.2' * (Lv - 1) + '.'      # scn.UNet nests the next level at [4][1][2] of its Sequential
    assert names[i_first].startswith(deepest) and not any(k.startswith(deepest) for k in names[:i_first]), names[i_first]
    assert 0.3 < (grads.flat.numel() - off) / grads.flat.numel() < 0.9


def test_fused_conv_pieces_vs_oracle(dev):
    """urn_gconv_fwd_ex: BatchNormReLU folded into the load, column statistics epilogue, BatchNorm-backward
    reduce epilogue; urn_gconv_bwd_dw_ex with the same input transform -- each against the oracle."""
    import ctypes
    from uresnet_pytorch_amd import lib as L_, sparse_ops as so
    L = L_.load()
    S, cin, cout = 32, 32, 48
    c, f = cloud(9, S, 1500, 2, 0)
    geo = so.SparseGeometry(torch.from_numpy(c).to(dev), S, 1)
    ref = orc.Geometry(c, f, S, 1)
    n = ref.n[0]
    rng = np.random.default_rng(5)
    x = rng.normal(size=(n, cin)).astype(np.float32)
    W = (rng.normal(size=(27, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    g_, b_ = (1 + 0.1 * rng.normal(size=cin)).astype(np.float32), (0.1 * rng.normal(size=cin)).astype(np.float32)
    u_ref, mean, invstd = orc.bn_relu_fwd(x, g_, b_, True)
    scale = (g_ * invstd).astype(np.float32); shift = (b_ - mean * scale).astype(np.float32)
    y_ref = orc.conv_fwd(u_ref, W, ref.nbr[0])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xt, Wt, sc, sh = t(x), t(W), t(scale), t(shift)
    wt = so._transpose_w(Wt)
    y = torch.empty((n, cout), device=dev)
    part = torch.zeros(L.urn_gconv_part_bytes(n, cout) // 8, dtype=torch.float64, device=dev)

    A = L_.GConvArgs
    npart = ctypes.c_int()
    a = A(x=xt.data_ptr(), wt=wt.data_ptr(), tbl=geo.nbr[0].data_ptr(), ld=geo.ld, K=27, flip=0, n_out=n, cin=cin, cout=cout,
          y=y.data_ptr(), xf_scale=sc.data_ptr(), xf_shift=sh.data_ptr(), epilogue=1, part=part.data_ptr())
    L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a), ctypes.byref(npart), L_.stream()))
    assert rel(y.cpu().numpy(), y_ref) < TOL
    pp = part[:npart.value * 2 * cout].view(npart.value, 2, cout).sum(0).cpu().numpy()
    assert rel(pp[0], y_ref.astype(np.float64).sum(0)) < 1e-6 and rel(pp[1], (y_ref.astype(np.float64) ** 2).sum(0)) < 1e-6
    # weight gradient with the transform
    dy = rng.normal(size=(n, cout)).astype(np.float32)
    dW = torch.zeros_like(Wt)
    dyt = t(dy)
    L_.check(L.urn_gconv_bwd_dw_ex(xt.data_ptr(), sc.data_ptr(), sh.data_ptr(), dyt.data_ptr(), geo.nbr[0].data_ptr(),
                                   geo.ld, 27, n, cin, cout, dW.data_ptr(), L_.stream()))
    du_ref, dW_ref = orc.conv_bwd(u_ref, W, ref.nbr[0], dy, ref.nbr_inv[0])
    assert rel(dW.cpu().numpy(), dW_ref) < TOL
    # input gradient with the BatchNorm-backward reduce epilogue, then finalize + apply == oracle BN backward
    gbuf = torch.empty((n, cin), device=dev)
    part2 = torch.zeros(L.urn_gconv_part_bytes(n, cin) // 8, dtype=torch.float64, device=dev)
    mt, it = t(mean), t(invstd)
    a2 = A(x=dyt.data_ptr(), wt=Wt.data_ptr(), tbl=geo.nbr[0].data_ptr(), ld=geo.ld, K=27, flip=1, n_out=n, cin=cout, cout=cin,
           y=gbuf.data_ptr(), epilogue=2, part=part2.data_ptr(), e_x=xt.data_ptr(), e_scale=sc.data_ptr(),
           e_shift=sh.data_ptr(), e_mean=mt.data_ptr(), e_invstd=it.data_ptr())
    L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a2), ctypes.byref(npart), L_.stream()))
    dg = torch.zeros(cin, device=dev); db = torch.zeros(cin, device=dev); coef = torch.empty(2 * cin, device=dev)
    L_.check(L.urn_bn_finalize_bwd(part2.data_ptr(), npart.value, n, cin, dg.data_ptr(), db.data_ptr(), coef.data_ptr(),
                                   coef.data_ptr() + 4 * cin, L_.stream()))
    extra = rng.normal(size=(n, cin)).astype(np.float32)
    gt, bt, et = t(g_), t(b_), t(extra)   # keep the device copies alive across the launches
    dx = torch.empty((n, cin), device=dev)
    L_.check(L.urn_bn_bwd_apply(xt.data_ptr(), gbuf.data_ptr(), et.data_ptr(), n, cin, gt.data_ptr(), mt.data_ptr(),
                                it.data_ptr(), coef.data_ptr(), coef.data_ptr() + 4 * cin, dx.data_ptr(), L_.stream()))
    dx_ref, dg_ref, db_ref = orc.bn_relu_bwd(x, u_ref, du_ref, g_, mean, invstd, True)
    assert rel(dx.cpu().numpy(), dx_ref + extra) < 5 * TOL
    assert rel(dg.cpu().numpy(), dg_ref) < 5 * TOL and rel(db.cpu().numpy(), db_ref) < 5 * TOL

    # --- accumulated statistics (part_slots / xs_*): the same three steps without finalize launches -------------
    SL = 8
    # (a) producer: the column sums of x itself are produced by a 1x1 gather conv with identity weights
    eye = t(np.eye(cin, dtype=np.float32)[None])
    x2 = torch.empty((n, cin), device=dev)
    sums_x = torch.zeros((SL, 2, cin), dtype=torch.float64, device=dev)
    a3 = A(x=xt.data_ptr(), wt=eye.data_ptr(), tbl=geo.nbr[0].data_ptr() + 13 * geo.ld * 4, ld=geo.ld, K=1, flip=0, n_out=n,
           cin=cin, cout=cin, y=x2.data_ptr(), epilogue=1, part=sums_x.data_ptr(), part_slots=SL)
    L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a3), ctypes.byref(npart), L_.stream()))
    assert npart.value == SL and torch.equal(x2, xt)
    tot = sums_x.sum(0).cpu().numpy()
    assert rel(tot[0], x.astype(np.float64).sum(0)) < 1e-12 and rel(tot[1], (x.astype(np.float64) ** 2).sum(0)) < 1e-12
    # (b) consumer derives scale/shift from the sums (two slabs = a channel concat), stores mean/invstd/scale/shift
    h = cin // 2
    sa = sums_x[:, :, :h].contiguous(); sb = sums_x[:, :, h:].contiguous()
    outs = [torch.zeros(cin, device=dev) for _ in range(4)]
    rm, rv = torch.zeros(cin, device=dev), torch.ones(cin, device=dev)
    y2 = torch.empty((n, cout), device=dev)
    sums_y = torch.zeros((SL, 2, cout), dtype=torch.float64, device=dev)
    a4 = A(x=xt.data_ptr(), wt=wt.data_ptr(), tbl=geo.nbr[0].data_ptr(), ld=geo.ld, K=27, flip=0, n_out=n, cin=cin, cout=cout,
           y=y2.data_ptr(), epilogue=1, part=sums_y.data_ptr(), part_slots=SL, fin_eps=1e-4, fin_momentum=0.99,
           xs_slots=SL, xs_split=h, xs_n=n, xs_gamma=gt.data_ptr(), xs_beta=bt.data_ptr(),
           xs_mean=outs[0].data_ptr(), xs_invstd=outs[1].data_ptr(), xs_scale=outs[2].data_ptr(), xs_shift=outs[3].data_ptr(),
           xs_running_mean=rm.data_ptr(), xs_running_var=rv.data_ptr())
    a4.xs_ld[0], a4.xs_ld[1] = h, cin - h
    a4.xs_sums[0], a4.xs_sums[1] = sa.data_ptr(), sb.data_ptr()
    L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a4), ctypes.byref(npart), L_.stream()))
    assert rel(y2.cpu().numpy(), y_ref) < TOL
    assert rel(outs[0].cpu().numpy(), mean) < 1e-6 and rel(outs[1].cpu().numpy(), invstd) < 1e-6
    assert rel(outs[2].cpu().numpy(), scale) < 1e-6 and rel(outs[3].cpu().numpy(), shift) < 1e-5
    var = x.astype(np.float64).var(0)
    assert rel(rm.cpu().numpy(), 0.01 * mean) < 1e-5 and rel(rv.cpu().numpy(), 0.99 + 0.01 * var) < 1e-6
    ty = sums_y.sum(0).cpu().numpy()
    assert rel(ty[0], y_ref.astype(np.float64).sum(0)) < 1e-6
    # (c) backward reduce accumulated, apply straight from the slab
    g2 = torch.empty((n, cin), device=dev)
    sums_g = torch.zeros((SL, 2, cin), dtype=torch.float64, device=dev)
    a5 = A(x=dyt.data_ptr(), wt=Wt.data_ptr(), tbl=geo.nbr[0].data_ptr(), ld=geo.ld, K=27, flip=1, n_out=n, cin=cout, cout=cin,
           y=g2.data_ptr(), epilogue=2, part=sums_g.data_ptr(), part_slots=SL, e_x=xt.data_ptr(), e_scale=sc.data_ptr(),
           e_shift=sh.data_ptr(), e_mean=mt.data_ptr(), e_invstd=it.data_ptr())
    L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a5), ctypes.byref(npart), L_.stream()))
    assert torch.equal(g2, gbuf)
    dg2 = torch.zeros(cin, device=dev); db2 = torch.zeros(cin, device=dev); dx2 = torch.empty((n, cin), device=dev)
    L_.check(L.urn_bn_bwd_apply_sums(xt.data_ptr(), g2.data_ptr(), et.data_ptr(), 0, n, cin, gt.data_ptr(), mt.data_ptr(),
                                     it.data_ptr(), sums_g.data_ptr(), SL, dg2.data_ptr(), db2.data_ptr(), dx2.data_ptr(),
                                     L_.stream()))
    assert rel(dx2.cpu().numpy(), dx_ref + extra) < 5 * TOL and rel(dx2.cpu().numpy(), dx.cpu().numpy()) < 1e-6
    assert rel(dg2.cpu().numpy(), dg_ref) < 5 * TOL and rel(db2.cpu().numpy(), db_ref) < 5 * TOL

    # --- operands that are column blocks of wider row matrices (the halves of a channel concat) ----------------------
    wide_y = torch.full((n, cout + 16), 7.0, device=dev)
    a6 = A(x=xt.data_ptr(), wt=wt.data_ptr(), tbl=geo.nbr[0].data_ptr(), ld=geo.ld, K=27, flip=0, n_out=n, cin=cin, cout=cout,
           y=wide_y.data_ptr() + 4 * 16, ldy=cout + 16, xf_scale=sc.data_ptr(), xf_shift=sh.data_ptr())
    L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a6), None, L_.stream()))
    assert torch.equal(wide_y[:, 16:], y) and bool((wide_y[:, :16] == 7.0).all())
    wide_x = torch.randn((n, cin + 32), device=dev); wide_x[:, 32:] = xt
    y3 = torch.empty((n, cout), device=dev)
    a7 = A(x=wide_x.data_ptr() + 4 * 32, wt=wt.data_ptr(), tbl=geo.nbr[0].data_ptr(), ld=geo.ld, K=27, flip=0, n_out=n, cin=cin,
           cout=cout, y=y3.data_ptr(), ldx=cin + 32, xf_scale=sc.data_ptr(), xf_shift=sh.data_ptr())
    L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a7), None, L_.stream()))
    assert torch.equal(y3, y)
    wide_dy = torch.randn((n, cout + 16), device=dev); wide_dy[:, 16:] = dyt
    dW3 = torch.zeros_like(Wt)
    L_.check(L.urn_gconv_bwd_dw_strided(xt.data_ptr(), sc.data_ptr(), sh.data_ptr(), wide_dy.data_ptr() + 4 * 16, cout + 16,
                                        geo.nbr[0].data_ptr(), geo.ld, 27, n, cin, cout, dW3.data_ptr(), L_.stream()))
    assert rel(dW3.cpu().numpy(), dW.cpu().numpy()) < 1e-6
    wide_e = torch.randn((n, cin + 16), device=dev); wide_e[:, :cin] = et
    dg3 = torch.zeros(cin, device=dev); db3 = torch.zeros(cin, device=dev); dx3 = torch.empty((n, cin), device=dev)
    L_.check(L.urn_bn_bwd_apply_sums(xt.data_ptr(), g2.data_ptr(), wide_e.data_ptr(), cin + 16, n, cin, gt.data_ptr(), mt.data_ptr(),
                                     it.data_ptr(), sums_g.data_ptr(), SL, dg3.data_ptr(), db3.data_ptr(), dx3.data_ptr(),
                                     L_.stream()))
    assert torch.equal(dx3, dx2)


def test_executor_two_forwards_before_backward(dev):
    """Gradient accumulation over two events run as two forwards then one backward (trainval sub-steps):
    each forward gets its own executor slot, so neither tape is overwritten."""
    from uresnet_pytorch_amd.models import SparseSegmentationLoss
    S, m, L, nc = 32, 16, 3, 5
    flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=m, URESNET_NUM_STRIDES=L, SPATIAL_SIZE=S, NUM_CLASS=nc)
    P = orc.init_params(m, L, nc, seed=3)
    blobs = [make_sparse_blob([s], S, 700) for s in (1, 2)]
    crit = SparseSegmentationLoss(flags)
    res = []
    for use_ex in (True, False):
        net = make_model(flags, P, dev)
        net.use_executor = use_ex
        total = 0
        for b in blobs:
            d = torch.from_numpy(b['data']).to(dev); lab = torch.from_numpy(b['label']).to(dev)
            total = total + crit(net(d), [d], [lab], None)[0]
        total.backward()
        res.append({k: p.grad.detach().cpu().numpy().copy() for k, p in net.named_parameters()})
    for k in res[0]:
        assert rel(res[0][k], res[1][k]) < 1e-5, k


def test_head_and_loss_kernels_vs_torch(dev):
    """urn_head_* (OutputLayer+Linear) and urn_ce_* (per-event mean CE, accuracy) against torch / the oracle."""
    from uresnet_pytorch_amd import sparse_ops as so
    rng = np.random.default_rng(11)
    n, m, nc = 5000, 16, 5
    rows = rng.normal(size=(n, m)).astype(np.float32)
    W = (rng.normal(size=(nc, m)) * 0.3).astype(np.float32); b = rng.normal(size=nc).astype(np.float32)
    data = np.zeros((n, 5), np.float32); data[:, 3] = rng.integers(0, 3, size=n) * 2     # batch ids 0, 2, 4 (gaps)
    lab = rng.integers(0, nc, size=(n, 1)).astype(np.float32)
    wgt = rng.uniform(0.5, 2.0, size=(n, 1)).astype(np.float32)
    for use_w in (False, True):
        rt = torch.from_numpy(rows).to(dev).requires_grad_(True)
        Wt = torch.from_numpy(W).to(dev).requires_grad_(True); bt = torch.from_numpy(b).to(dev).requires_grad_(True)
        logits = so.HeadFunction.apply(rt, Wt, bt)
        wt_ = torch.from_numpy(wgt).to(dev) if use_w else None
        loss, out = so.SegmentationCEFunction.apply(logits, torch.from_numpy(data).to(dev), torch.from_numpy(lab).to(dev), wt_)
        (2.5 * loss).backward()
        # references: float64 torch on the CPU + the oracle's loss
        r64 = torch.tensor(rows, dtype=torch.float64, requires_grad=True)
        W64 = torch.tensor(W, dtype=torch.float64, requires_grad=True); b64 = torch.tensor(b, dtype=torch.float64, requires_grad=True)
        l64 = r64 @ W64.t() + b64
        assert rel(logits.detach().cpu().numpy(), l64.detach().numpy()) < TOL
        loss_ref, acc_ref, _ = orc.segmentation_loss(l64.detach().numpy().astype(np.float32), data, lab, wgt if use_w else None)
        assert abs(loss.item() - loss_ref) < 1e-5 * abs(loss_ref) and abs(out[1].item() - acc_ref) < 1e-5
        tl = 0
        bid = torch.tensor(data[:, 3]); labt = torch.tensor(lab[:, 0]).long()
        for e in bid.unique():
            mk = bid == e
            ce = torch.nn.functional.cross_entropy(l64[mk], labt[mk], reduction='none')
            if use_w:
                ce = ce * torch.tensor(wgt[:, 0], dtype=torch.float64)[mk]
            tl = tl + ce.mean()
        (2.5 * tl).backward()
        assert rel(rt.grad.cpu().numpy(), r64.grad.numpy()) < TOL
        assert rel(Wt.grad.cpu().numpy(), W64.grad.numpy()) < TOL and rel(bt.grad.cpu().numpy(), b64.grad.numpy()) < TOL


def test_trainval_sparse_gpu(dev, tmp_path):
    """The reference's trainer API on the GPU (reference uresnet/trainval.py:42-134): train_step with two
    sub-steps (gradient accumulation -> two executor slots), result dict, checkpoint, eval-mode forward."""
    from uresnet_pytorch_amd.trainval import trainval

    def flags(train, path=''):
        return SimpleNamespace(MODEL_NAME='uresnet_sparse', DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=3,
                               SPATIAL_SIZE=64, NUM_CLASS=5, BN_MOMENTUM=0.9, TRAIN=train, GPUS=[0], LEARNING_RATE=1e-3,
                               MODEL_PATH=path, WEIGHT_PREFIX=str(tmp_path / 'snap'))
    blobs = [make_sparse_blob([s], 64, 1200) for s in (1, 2)]
    data_blob = {'data': [[b['data']] for b in blobs], 'label': [[b['label']] for b in blobs]}
    torch.manual_seed(0)
    t = trainval(flags(True))
    assert t.initialize() == 0
    p0 = torch.cat([p.detach().flatten() for p in t._net.parameters()]).clone()
    res = t.train_step(data_blob, epoch=0., batch_size=2)
    assert set(res.keys()) == {'segmentation', 'softmax', 'accuracy', 'loss_seg'}
    assert len(res['segmentation']) == 2 and res['segmentation'][0].shape == (1200, 5)
    assert np.allclose(res['softmax'][1].sum(1), 1.0, atol=1e-5)
    assert np.isfinite(res['loss_seg']) and 0.0 <= res['accuracy'] <= 1.0
    p1 = torch.cat([p.detach().flatten() for p in t._net.parameters()])
    assert not torch.equal(p0, p1)                      # Adam stepped
    t.save_state(7)
    t2 = trainval(flags(False, str(tmp_path / 'snap-7.ckpt')))
    assert t2.initialize() == 8
    for (k1, v1), (k2, v2) in zip(t._net.state_dict().items(), t2._net.state_dict().items()):
        assert k1 == k2 and torch.equal(v1.cpu(), v2.cpu()), k1
    r2 = t2.forward(data_blob, epoch=0., batch_size=2)  # eval mode: running statistics, per-layer path
    assert r2['segmentation'][0].shape == (1200, 5) and np.isfinite(r2['loss_seg'])


def test_npz_reader_on_device_feeds_the_trainer(dev, tmp_path):
    """-io npz_sparse with -iod (the per-GPU concat of a step's events on the device): the blob assembled in
    HBM is bitwise the host-assembled one, and a trainer step on it gives bitwise the same loss and parameters."""
    from uresnet_pytorch_amd.trainval import trainval
    from uresnet_pytorch_amd.iotools import io_factory, array_io
    from uresnet_pytorch_amd.iotools.synthetic import generate_event
    events = []
    for seed in range(4):
        c, v, l = generate_event(seed, 64, 900 + 50 * seed)
        events.append({'voxels': c, 'feature': v, 'label': l})
    path = str(tmp_path / 'ev.npz')
    array_io.write_sparse_npz(path, events)

    def flags(on_device):
        return SimpleNamespace(MODEL_NAME='uresnet_sparse', IO_TYPE='npz_sparse', INPUT_FILE=[path], DATA_KEYS=['data', 'label'],
                               DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=3, SPATIAL_SIZE=64, NUM_CLASS=5,
                               BN_MOMENTUM=0.9, TRAIN=True, GPUS=[0], LEARNING_RATE=1e-3, MODEL_PATH='', WEIGHT_PREFIX='',
                               BATCH_SIZE=2, MINIBATCH_SIZE=2, SHUFFLE=0, LIMIT_NUM_SAMPLE=-1, IO_ON_DEVICE=on_device,
                               COMPUTE_WEIGHT=False, OUTPUT_FILE='')
    out = {}
    for on_device in (False, True):
        fl = flags(on_device)
        io = io_factory(fl)
        io.initialize()
        idx, blob = io.next()
        if on_device:
            assert torch.is_tensor(blob['data'][0]) and blob['data'][0].is_cuda
        torch.manual_seed(0)
        t = trainval(fl)
        t.initialize()
        res = t.train_step({'data': [blob['data']], 'label': [blob['label']]}, epoch=0., batch_size=2)
        host = lambda a: a.cpu().numpy() if torch.is_tensor(a) else a
        out[on_device] = (host(blob['data'][0]), host(blob['label'][0]), res['loss_seg'],
                          torch.cat([p.detach().flatten() for p in t._net.parameters()]).cpu().numpy())
    assert np.array_equal(out[False][0], out[True][0]) and np.array_equal(out[False][1], out[True][1])
    assert sorted(np.unique(out[True][0][:, 3]).tolist()) == [0.0, 1.0]
    assert out[False][2] == out[True][2] and np.isfinite(out[True][2])
    # parameters after the Adam step: equal up to the weight-gradient atomics' last bits -- which Adam's first step turns into
    # +-lr for the few parameters whose gradient is itself rounding noise, hence a bound of 2 lr and "almost all within 1e-6"
    diff = np.abs(out[False][3] - out[True][3])
    assert diff.max() <= 2.1e-3 and np.mean(diff > 1e-6) < 0.01, (diff.max(), np.mean(diff > 1e-6))


def test_empty_and_tiny_inputs(dev):
    """Edge cases: a single voxel; all rows duplicates of one site."""
    from uresnet_pytorch_amd.models import SparseUResNet
    flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=3, SPATIAL_SIZE=32, NUM_CLASS=5)
    net = SparseUResNet(flags).to(dev).train()
    one = torch.tensor([[5., 6., 7., 0., 0.3]], device=dev)
    out = net(one)[0]
    assert out.shape == (1, 5) and torch.isfinite(out).all()
    dup = torch.tensor([[5., 6., 7., 0., 0.3]] * 4, device=dev)
    out = net(dup)[0]
    out.sum().backward()
    assert out.shape == (4, 5) and torch.isfinite(out).all()
    assert torch.allclose(out[0], out[3])               # the same site feeds every duplicate row


def test_flat_adam_matches_torch_adam(dev):
    """parallel.FlatAdam (urn_adam_flat over contiguous segments) against torch.optim.Adam, incl. the optimizer
    state_dict round trip of the reference's checkpoint format (trainval.py:32-40,171-196)."""
    from uresnet_pytorch_amd import parallel
    torch.manual_seed(3)
    def make():
        torch.manual_seed(4)
        return torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5)).to(dev)
    a, b = make(), make()
    fg = parallel.FlatGradients(a)
    oa = parallel.FlatAdam(fg, lr=3e-3)
    ob = torch.optim.Adam(b.parameters(), lr=3e-3)
    x = torch.randn(64, 37, device=dev)
    for it in range(4):
        fg.zero(); ob.zero_grad()
        a(x).square().sum().backward(); b(x).square().sum().backward()
        oa.step(); ob.step()
        if it == 1:   # checkpoint round trip through torch.optim.Adam's state layout
            sd = oa.state_dict()
            assert set(sd['state'][0].keys()) == {'step', 'exp_avg', 'exp_avg_sq'}
            oa2 = parallel.FlatAdam(fg, lr=3e-3)
            oa2.load_state_dict(sd)
            oa = oa2
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert rel(pa.detach().cpu().numpy(), pb.detach().cpu().numpy()) < 2e-6
    sa, sb = oa.state_dict(), ob.state_dict()
    for k in sb['state']:
        assert rel(sa['state'][k]['exp_avg_sq'].cpu().numpy(), sb['state'][k]['exp_avg_sq'].cpu().numpy()) < 1e-6
        assert float(sa['state'][k]['step']) == float(sb['state'][k]['step'])


@pytest.mark.parametrize('prec,tol_op,tol_net', [('bf16', 6e-3, 5e-2), ('fp16', 8e-4, 8e-3)])
def test_reduced_precision_operands(dev, prec, tol_op, tol_net):
    """BASELINE configs[1] (bf16) / configs[4] (fp16): MFMA operands rounded to 16 bits while they are staged in LDS,
    fp32 tensors and accumulation.  Each operator against the fp64 oracle: bf16 has 8 significand bits (unit roundoff
    3.9e-3, observed norm-wise error 2.4e-3), fp16 has 11 (4.9e-4, observed 2.9e-4); the small network end to end."""
    from uresnet_pytorch_amd import lib as L_, sparse_ops as so
    from uresnet_pytorch_amd.models import SparseSegmentationLoss
    L = L_.load()
    try:
        L_.set_precision(prec)
        S, cin, cout = 32, 32, 48
        c, f = cloud(9, S, 1500, 2, 0)
        geo = so.SparseGeometry(torch.from_numpy(c).to(dev), S, 1)
        ref = orc.Geometry(c, f, S, 1)
        n = ref.n[0]
        rng = np.random.default_rng(5)
        x = rng.normal(size=(n, cin)).astype(np.float32)
        W = (rng.normal(size=(27, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
        dy = rng.normal(size=(n, cout)).astype(np.float32)
        xt = torch.from_numpy(x).to(dev).requires_grad_(True); Wt = torch.from_numpy(W).to(dev).requires_grad_(True)
        y = so.GConvFunction.apply(xt, Wt, None, geo.nbr[0], geo.nbr[0], 1, geo.ld, n, n)
        y.backward(torch.from_numpy(dy).to(dev))
        y_ref = orc.conv_fwd(x, W, ref.nbr[0])
        dx_ref, dW_ref = orc.conv_bwd(x, W, ref.nbr[0], dy, ref.nbr_inv[0])
        for got, want in ((y.detach(), y_ref), (xt.grad, dx_ref), (Wt.grad, dW_ref)):
            e = rel(got.cpu().numpy(), want)
            assert 1e-6 < e < tol_op, e      # > 1e-6: the reduced-precision kernels really ran
        # the weight gradient on the compacted rule lists (two stages, no atomics) with the same operand rounding
        # (k_dw_pairs<.., PREC>: the four rules a lane loads are the four contraction slots of one 16-bit MFMA), bitwise reproducible
        # a full-width weight-gradient tile (four input-channel blocks: wave w owns block w, urn_set_option "dw_rowmode") and
        # paired weight fragments with several chunks (cin / 16 even)
        for ci2, co2 in ((128, 80), (160, 32)):
            x3 = rng.normal(size=(n, ci2)).astype(np.float32)
            W3 = (rng.normal(size=(27, ci2, co2)) / np.sqrt(27 * ci2)).astype(np.float32)
            dy3 = rng.normal(size=(n, co2)).astype(np.float32)
            xt3 = torch.from_numpy(x3).to(dev).requires_grad_(True); Wt3 = torch.from_numpy(W3).to(dev).requires_grad_(True)
            p3 = geo.pairs['nbr'][0]
            y3 = so.GConvFunction.apply(xt3, Wt3, None, geo.nbr[0], geo.nbr[0], 1, geo.ld, n, n, p3, p3)
            y3.backward(torch.from_numpy(dy3).to(dev))
            dx3_ref, dW3_ref = orc.conv_bwd(x3, W3, ref.nbr[0], dy3, ref.nbr_inv[0])
            for got, want in ((y3.detach(), orc.conv_fwd(x3, W3, ref.nbr[0])), (xt3.grad, dx3_ref), (Wt3.grad, dW3_ref)):
                e = rel(got.cpu().numpy(), want)
                assert 1e-6 < e < tol_op, (ci2, co2, e)
        so.set_deterministic_dw(True, 'pairs')
        try:
            p0 = geo.pairs['nbr'][0]
            dws = []
            for _ in range(2):
                x2 = torch.from_numpy(x).to(dev).requires_grad_(True); W2 = torch.from_numpy(W).to(dev).requires_grad_(True)
                so.GConvFunction.apply(x2, W2, None, geo.nbr[0], geo.nbr[0], 1, geo.ld, n, n, p0, p0).backward(torch.from_numpy(dy).to(dev))
                dws.append(W2.grad.detach().clone())
            e = rel(dws[0].cpu().numpy(), dW_ref)
            assert 1e-6 < e < tol_op, e
            assert torch.equal(dws[0], dws[1])
        finally:
            so.set_deterministic_dw(False)
        # whole network (executor, fused BatchNorm) against the oracle
        S, m, Lv, nc = 32, 16, 3, 5
        c, f = cloud(6, S, 800, 2, 11)
        pc = np.concatenate([c.astype(np.float32), f], 1)
        lab = np.random.default_rng(3).integers(0, nc, size=(len(pc), 1)).astype(np.float32)
        flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=m, URESNET_NUM_STRIDES=Lv, SPATIAL_SIZE=S, NUM_CLASS=nc, PRECISION=prec)
        P = orc.init_params(m, Lv, nc, seed=1)
        net = make_model(flags, P, dev)
        data = torch.from_numpy(pc).to(dev); label = torch.from_numpy(lab).to(dev)
        out = net(data)
        loss, acc = SparseSegmentationLoss(flags)(out, [data], [label], None)
        loss.backward()
        oref = orc.SparseUResNetOracle(P, m, Lv, nc, S)
        logits_ref = oref.forward(pc)
        loss_ref, _, dl = orc.segmentation_loss(logits_ref, pc, lab)
        G, _ = oref.backward(dl)
        assert rel(out[0].detach().cpu().numpy(), logits_ref) < tol_net
        assert abs(loss.item() - loss_ref) < tol_net * abs(loss_ref)
        # gradients: norm-wise over ALL parameters (single small tensors -- a BatchNorm shift deep in the U -- see many
        # ReLU-mask flips at this precision and are individually much noisier)
        got = np.concatenate([p.grad.cpu().numpy().ravel() for k, p in net.named_parameters()])
        want = np.concatenate([np.asarray(G[k]).ravel() for k, p in net.named_parameters()])
        # (BatchNorm backward subtracts two means from the gradient: that cancellation amplifies the operand rounding;
        # observed 0.16 for bf16 and 0.054 for fp16 on this network -- fp16 additionally loses the smallest gradient
        # values, ~1e-6 here, to its narrow exponent range; no loss scaling is applied.  The strict bounds are the
        # per-operator ones above.)
        assert rel(got, want) < 8 * tol_net, rel(got, want)
    finally:
        L_.set_precision('fp32')


def test_network_cfg5_size_modes_agree(dev):
    """BASELINE configs[4] topology (-ss 768, 200k voxels, -uf 32 -uns 7: channels 32..224, inputs up to 448 wide walked
    in chunks) in fp32: too large for the oracle in seconds, so the check is a size-independent property -- the two
    statistics modes of the executor (accumulated sums vs per-workgroup slabs + finalize launches, different
    kernels' epilogues and prologues, different launch sequences) agree on logits, loss and gradients."""
    from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss
    S, m, Lv, nc = 768, 32, 7, 5
    blob = make_sparse_blob([3], S, 200000)
    flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=m, URESNET_NUM_STRIDES=Lv, SPATIAL_SIZE=S, NUM_CLASS=nc)
    data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
    res = []
    for fl in (0, 4):
        torch.manual_seed(0)
        net = SparseUResNet(flags).to(dev).train()
        net.executor_flags = fl
        out = net(data)
        loss, acc = SparseSegmentationLoss(flags)(out, [data], [label], None)
        loss.backward()
        g = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        assert torch.isfinite(out[0]).all() and torch.isfinite(g).all()
        res.append((out[0].detach().cpu().numpy(), float(loss), g.cpu().numpy()))
        del net, out, loss
    assert rel(res[0][0], res[1][0]) < 1e-5
    assert abs(res[0][1] - res[1][1]) < 1e-5 * abs(res[1][1])
    assert rel(res[0][2], res[1][2]) < 1e-3      # ReLU-mask flips between the two summation orders (see cfg3 test)


def test_network_cfg5_fp16_full_size_with_loss_scaling(dev):
    """BASELINE configs[4] at FULL size in fp16: -ss 768, 200k voxels, -uf 32 -uns 7, MFMA operands rounded to fp16 (fp32
    tensors and accumulation).  Too large for the oracle in seconds, so the fp32 run of the same executor is the
    reference (itself pinned to the oracle at cfg3 / cfg4 size above):
      * everything finite; logits within the fp16 operator tolerance of the fp32 run (8e-3 norm-wise, the bound of
        test_reduced_precision_operands for the small network);
      * gradients: WITHOUT loss scaling the gradient operands (1e-6 .. 1e-8 here) fall below fp16's range when they are
        rounded for the matrix cores; with the trainer's loss scale (flags -ls, 2^12) the norm-wise deviation of all
        parameter gradients from the fp32 run is bounded by 0.02 (measured 0.004; 0.023 without the scale)."""
    from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss
    S, m, Lv, nc = 768, 32, 7, 5
    blob = make_sparse_blob([5], S, 200000)
    data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)

    def run(prec, scale):
        flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=m, URESNET_NUM_STRIDES=Lv, SPATIAL_SIZE=S, NUM_CLASS=nc, PRECISION=prec)
        torch.manual_seed(0)
        net = SparseUResNet(flags).to(dev).train()
        out = net(data)
        loss, acc = SparseSegmentationLoss(flags)(out, [data], [label], None)
        (loss * scale).backward()
        g = torch.cat([p.grad.reshape(-1) for p in net.parameters()]) / scale
        return out[0].detach().cpu().numpy(), float(loss.detach()), g.cpu().numpy()
    try:
        lo32, l32, g32 = run('fp32', 1.0)
        lo16, l16, g16 = run('fp16', 1.0)
        lo16s, l16s, g16s = run('fp16', 4096.0)
    finally:
        from uresnet_pytorch_amd import lib as L_
        L_.set_precision('fp32')
    for a in (lo16, g16, lo16s, g16s):
        assert np.isfinite(a).all()
    e_logits = rel(lo16, lo32)
    assert 1e-6 < e_logits < 8e-3, e_logits                     # > 1e-6: the fp16 kernels really ran
    assert abs(l16 - l32) < 8e-3 * abs(l32)
    e_plain, e_scaled = rel(g16, g32), rel(g16s, g32)
    print('cfg5 fp16 full size: logits %.2e, gradients %.3f unscaled / %.3f with loss scale 2^12' % (e_logits, e_plain, e_scaled))
    assert e_scaled < 0.02, (e_plain, e_scaled)                 # measured 0.004 (0.023 without the loss scale)
    assert e_scaled <= e_plain * 1.05, (e_plain, e_scaled)


def test_executor_eval_mode_matches_per_layer_path(dev):
    """Inference (model.eval(), no autograd: trainval.forward with TRAIN=False) through the executor -- BatchNorm folded
    into the convolutions with the RUNNING statistics -- against the per-layer eval path."""
    S, m, L, nc = 64, 16, 4, 5
    blob = make_sparse_blob([5, 6], S, 3000)
    flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=m, URESNET_NUM_STRIDES=L, SPATIAL_SIZE=S, NUM_CLASS=nc)
    P = orc.init_params(m, L, nc, seed=2)
    data = torch.from_numpy(blob['data']).to(dev)
    net = make_model(flags, P, dev)
    net.train()
    for _ in range(2):      # two training forwards move the running statistics away from their initial values
        net(data)
    net.eval()
    running_before = {k: v.clone() for k, v in net.state_dict().items() if 'running' in k}
    with torch.no_grad():
        out_ex = net(data)[0]
        net.use_executor = False
        out_pl = net(data)[0]
    assert rel(out_ex.cpu().numpy(), out_pl.cpu().numpy()) < 1e-5
    for k, v in net.state_dict().items():
        if 'running' in k:
            assert torch.equal(v, running_before[k]), k      # eval does not touch the running statistics


def test_side_stream_handles_are_capped(dev, monkeypatch):
    """At most trunk.MAX_SIDE_STREAMS executor handles of a process own a side stream; further handles run
    single-stream and give the same results (and so do handles whose first backward picked another candidate side
    stream: the probe in urn_net.hip pick_side)."""
    import gc
    from uresnet_pytorch_amd import trunk
    from uresnet_pytorch_amd.models import SparseSegmentationLoss
    gc.collect()
    monkeypatch.setattr(trunk, 'MAX_SIDE_STREAMS', 2)
    base = trunk._SIDE_HANDLES
    S, m, L, nc = 32, 16, 3, 5
    flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=m, URESNET_NUM_STRIDES=L, SPATIAL_SIZE=S, NUM_CLASS=nc)
    P = orc.init_params(m, L, nc, seed=3)
    blob = make_sparse_blob([1], S, 700)
    d = torch.from_numpy(blob['data']).to(dev); lab = torch.from_numpy(blob['label']).to(dev)
    nets, outs = [], []
    for i in range(4):
        net = make_model(flags, P, dev)
        out = net(d)
        loss, _ = SparseSegmentationLoss(flags)(out, [d], [lab], None)
        loss.backward()
        nets.append(net); outs.append(out[0].detach().cpu().numpy())
    assert trunk._SIDE_HANDLES <= max(base, trunk.MAX_SIDE_STREAMS)
    assert sum(n._executor._side_handles for n in nets) <= trunk.MAX_SIDE_STREAMS
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])       # forward does not depend on the stream layout
    g0 = torch.cat([p.grad.reshape(-1) for p in nets[0].parameters()]).cpu().numpy()
    g3 = torch.cat([p.grad.reshape(-1) for p in nets[3].parameters()]).cpu().numpy()
    assert rel(g3, g0) < 1e-5
    del nets
    gc.collect()
    assert trunk._SIDE_HANDLES == base


def test_many_geometries_with_deferred_count_readback(dev):
    """the level counts travel through a ring of 16 pinned host buffers: a geometry whose slot has been handed out again
    before its sync() must still report its own counts"""
    from uresnet_pytorch_amd import sparse_ops as so
    geos = []
    for s in range(40):
        b = make_sparse_blob([s], 64, 500 + 10 * s)
        geos.append(so.SparseGeometry(torch.from_numpy(b['data'][:, :4].astype(np.int32)).to(dev), 64, 3, defer_sync=True))
    for g in geos:
        g.sync()
        assert g.n == g.counts.cpu().tolist()[:3]
