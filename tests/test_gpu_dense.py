"""GPU tests of the dense implicit-GEMM kernels (csrc/urn_dense.hip, through the C ABI) against torch on the CPU:
every convolution form of the dense U-ResNet (reference uresnet/models/uresnet_dense.py) -- replicate-padded Conv k3 s1 /
k3 s2 / k1 s2 / k1 s1 and ConvTranspose k3 s2 p1 op1, 2-D and 3-D -- forward, input gradient (incl. the gradient of the
replicate padding) and weight gradient.  fp32 operands: 1e-5 relative (norm-wise); bf16 operands: 1e-2."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().double().cpu().numpy(); b = b.detach().double().cpu().numpy()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    from uresnet_pytorch_amd import lib
    lib.load()
    return torch.device('cuda:0')


def ref_padding(kernel, stride, size):
    """reference uresnet_dense.py:19-26 (per dim)"""
    p = max(kernel - stride, 0) if size % stride == 0 else max(kernel - (size % stride), 0)
    return p // 2, p - p // 2


def to_rows(x):
    nd = x.dim() - 2
    return x.permute(0, *range(2, 2 + nd), 1).reshape(-1, x.shape[1]).contiguous()


def from_rows(rows, B, spatial):
    nd = len(spatial)
    return rows.reshape(B, *spatial, rows.shape[1]).permute(0, nd + 1, *range(1, nd + 1))


CASES = [  # dim, spatial, cin, cout, k, stride
    (2, (16, 16), 16, 16, 3, 1), (2, (20, 12), 8, 32, 3, 2), (2, (16, 16), 32, 16, 1, 2), (2, (9, 17), 1, 8, 3, 1),
    (3, (8, 8, 8), 16, 16, 3, 1), (3, (8, 12, 16), 16, 32, 3, 2), (3, (8, 8, 8), 32, 64, 1, 2), (3, (4, 4, 4), 64, 48, 3, 1),
    (3, (6, 10, 20), 1, 16, 3, 1), (3, (8, 8, 8), 16, 5, 3, 1), (3, (16, 16, 16), 80, 16, 1, 1), (3, (5, 7, 9), 16, 16, 3, 2),
    (3, (4, 4, 4), 64, 80, 3, 1),     # few slabs, >= 64 x 64 weights per tap: the LDS-transposing slab reduce (k_dense_dw_reduce_tt)
]


@pytest.mark.parametrize('prec,tol', [('fp32', 1e-5), ('bf16', 1e-2)])
@pytest.mark.parametrize('dim,spatial,cin,cout,k,stride', CASES)
def test_dense_conv_forward_backward(dev, dim, spatial, cin, cout, k, stride, prec, tol):
    from uresnet_pytorch_amd import dense_conv as dc
    dc.set_precision(prec)
    try:
        B = 2
        g = torch.Generator().manual_seed(cin * 7 + cout)
        x = torch.randn(B, cin, *spatial, generator=g)
        w = torch.randn(cout, cin, *([k] * dim), generator=g) / (cin * k ** dim) ** 0.5
        b = torch.randn(cout, generator=g)
        lo, hi = ref_padding(k, stride, spatial[-1])
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        xp = F.pad(xr, (lo, hi) * dim, mode='replicate') if (lo or hi) else xr
        y_ref = (F.conv3d if dim == 3 else F.conv2d)(xp, wr, br, stride=stride)
        dy = torch.randn(y_ref.shape, generator=g)
        y_ref.backward(dy)
        rows = to_rows(x).to(dev).requires_grad_(True)
        wg, bg = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        y = dc.DenseConvFunction.apply(rows, wg, bg, B, spatial, stride, lo, hi)
        assert tuple(y.shape) == (y_ref.numel() // cout, cout)
        y.backward(to_rows(dy).to(dev))
        assert rel(y, to_rows(y_ref)) < tol
        assert rel(rows.grad, to_rows(xr.grad)) < tol
        assert rel(wg.grad, wr.grad) < tol
        assert rel(bg.grad, br.grad) < 1e-5
    finally:
        dc.set_precision('fp32')


@pytest.mark.parametrize('prec,tol', [('fp32', 1e-5), ('bf16', 1e-2)])
@pytest.mark.parametrize('dim,spatial,cin,cout', [(2, (8, 8), 32, 16), (2, (5, 9), 16, 8), (3, (4, 4, 4), 32, 16), (3, (3, 5, 8), 64, 32),
                                                  (3, (8, 8, 8), 16, 16)])
def test_dense_conv_transpose_forward_backward(dev, dim, spatial, cin, cout, prec, tol):
    from uresnet_pytorch_amd import dense_conv as dc
    dc.set_precision(prec)
    try:
        B = 2
        g = torch.Generator().manual_seed(cin + cout)
        x = torch.randn(B, cin, *spatial, generator=g)
        w = torch.randn(cin, cout, *([3] * dim), generator=g) / (cin * 3 ** dim) ** 0.5
        b = torch.randn(cout, generator=g)
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y_ref = (F.conv_transpose3d if dim == 3 else F.conv_transpose2d)(xr, wr, br, stride=2, padding=1, output_padding=1)
        dy = torch.randn(y_ref.shape, generator=g)
        y_ref.backward(dy)
        rows = to_rows(x).to(dev).requires_grad_(True)
        wg, bg = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        y = dc.DenseConvTransposeFunction.apply(rows, wg, bg, B, spatial)
        y.backward(to_rows(dy).to(dev))
        assert rel(y, to_rows(y_ref)) < tol
        assert rel(rows.grad, to_rows(xr.grad)) < tol
        assert rel(wg.grad, wr.grad) < tol
        assert rel(bg.grad, br.grad) < 1e-5
    finally:
        dc.set_precision('fp32')


# ---- statistics in the convolution epilogue, fused BatchNorm + shortcut + ReLU row passes ---------------------------------
STAT_CASES = [  # dim, spatial, cin, cout, k, stride  (unsplit fast path, generic kernel, strided, split contraction, padded channels)
    (3, (16, 16, 16), 16, 16, 3, 1), (3, (8, 12, 16), 16, 32, 3, 2), (3, (8, 8, 8), 32, 64, 1, 2), (3, (4, 4, 4), 64, 128, 3, 1),
    (2, (20, 12), 8, 32, 3, 2), (2, (9, 17), 1, 8, 3, 1), (3, (8, 8, 8), 128, 256, 3, 1),
]


@pytest.mark.parametrize('dim,spatial,cin,cout,k,stride', STAT_CASES)
def test_dense_conv_epilogue_statistics(dev, dim, spatial, cin, cout, k, stride):
    """the [slots][2][cout] slab the epilogue fills = column sums / sums of squares of the output it wrote"""
    from uresnet_pytorch_amd import dense_conv as dc
    B = 2
    g = torch.Generator().manual_seed(cin + 3 * cout)
    x = torch.randn(B, cin, *spatial, generator=g)
    w = torch.randn(cout, cin, *([k] * dim), generator=g) / (cin * k ** dim) ** 0.5
    b = torch.randn(cout, generator=g)
    lo, hi = ref_padding(k, stride, spatial[-1])
    stats = dc.new_stats(cout + (-cout) % 16, dev)
    y = dc.DenseConvFunction.apply(to_rows(x).to(dev), w.to(dev), b.to(dev), B, spatial, stride, lo, hi, stats, False)
    s = stats.sum(0).cpu()
    yd = y.double().cpu()
    assert float((s[0, :cout] - yd.sum(0)).abs().max()) <= 1e-9 * float(yd.abs().sum(0).max())
    assert float((s[1, :cout] - (yd * yd).sum(0)).abs().max()) <= 1e-9 * float((yd * yd).sum(0).max())
    assert float(s[:, cout:].abs().max()) == 0.0 if s.shape[1] > cout else True


def test_dense_conv_transpose_epilogue_statistics(dev):
    from uresnet_pytorch_amd import dense_conv as dc
    B, cin, cout, spatial = 2, 32, 16, (4, 6, 8)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, cin, *spatial, generator=g)
    w = torch.randn(cin, cout, 3, 3, 3, generator=g) / 30.
    stats = dc.new_stats(cout, dev)
    y = dc.DenseConvTransposeFunction.apply(to_rows(x).to(dev), w.to(dev), None, B, spatial, stats, False)
    s = stats.sum(0).cpu()
    yd = y.double().cpu()
    assert float((s[0] - yd.sum(0)).abs().max()) <= 1e-9 * float(yd.abs().sum(0).max())
    assert float((s[1] - (yd * yd).sum(0)).abs().max()) <= 1e-9 * float((yd * yd).sum(0).max())


@pytest.mark.parametrize('c,n', [(16, 5000), (4, 777), (64, 1031), (512, 96)])
@pytest.mark.parametrize('relu', [False, True])
@pytest.mark.parametrize('res', ['none', 'identity', 'bn'])
def test_dense_bn_act_function_matches_torch(dev, c, n, relu, res):
    """BNActFunction (statistics slab in, one forward pass, reduce + apply backward) against the torch composite
    batch_norm(raw) [+ r | + batch_norm(r)] [-> relu] in fp64: forward 1e-6, every gradient 1e-5 (norm-wise)"""
    from uresnet_pytorch_amd import dense_hip as dh
    g = torch.Generator().manual_seed(c + n)
    raw = torch.randn(n, c, generator=g) * 2 + 0.5
    r = torch.randn(n, c, generator=g)
    ga, be, rga, rbe = (torch.randn(c, generator=g) for _ in range(4))
    dy = torch.randn(n, c, generator=g)
    eps = 1e-5

    def slab(t):
        s = torch.zeros(8, 2, c, dtype=torch.float64)
        td = t.double()
        s[3, 0] = td.sum(0); s[5, 1] = (td * td).sum(0)          # any split over the slots must do
        return s.to(dev)
    # torch fp64 reference
    P = [t.double().clone().requires_grad_(True) for t in (raw, ga, be, r, rga, rbe)]
    y = F.batch_norm(P[0], None, None, P[1], P[2], True, 0.0, eps)
    if res == 'identity':
        y = y + P[3]
    elif res == 'bn':
        y = y + F.batch_norm(P[3], None, None, P[4], P[5], True, 0.0, eps)
    if relu:
        y = F.relu(y)
    y.backward(dy.double())
    Q = [t.to(dev).requires_grad_(True) for t in (raw, ga, be, r, rga, rbe)]
    out = dh.BNActFunction.apply(Q[0], Q[1], Q[2], slab(raw), eps, relu, None if res == 'none' else Q[3],
                                 Q[4] if res == 'bn' else None, Q[5] if res == 'bn' else None, slab(r) if res == 'bn' else None, eps)
    out.backward(dy.to(dev))
    assert rel(out, y) < 1e-6
    names = ['raw', 'gamma', 'beta', 'res', 'res_gamma', 'res_beta']
    used = {'none': 3, 'identity': 4, 'bn': 6}[res]
    for i in range(used):
        assert rel(Q[i].grad, P[i].grad) < 1e-5, names[i]


@pytest.mark.parametrize('dim,S,B', [(2, 24, 3), (3, 12, 2)])
@pytest.mark.parametrize('weighted', [False, True])
def test_dense_segmentation_loss_on_device_matches_cpu_route(dev, dim, S, B, weighted):
    """DenseSegmentationLoss: the fused per-event kernel (GPU tensors) against the same class on CPU tensors (the
    reference's torch ops, uresnet_dense.py:235-260): loss, accuracy, gradient of the logits"""
    from types import SimpleNamespace
    from uresnet_pytorch_amd.models import DenseSegmentationLoss
    nc = 5
    g = torch.Generator().manual_seed(S + B)
    crit = DenseSegmentationLoss(SimpleNamespace())
    logits = torch.randn(B, nc, *([S] * dim), generator=g) * 2
    data = [(torch.rand(1, *([S] * dim), generator=g) * (torch.rand(1, *([S] * dim), generator=g) > 0.7)).float() for _ in range(B)]
    label = [torch.randint(0, nc, (1, *([S] * dim)), generator=g).float() for _ in range(B)]
    weight = [torch.rand(1, *([S] * dim), generator=g) + 0.5 for _ in range(B)] if weighted else None
    lc = logits.clone().requires_grad_(True)
    loss_c, acc_c = crit([lc[i] for i in range(B)], data, label, weight)
    loss_c.backward()
    # channels-last logits on the device, as the network hands them out
    rows = logits.permute(0, *range(2, 2 + dim), 1).reshape(-1, nc).to(dev).requires_grad_(True)
    lg = rows.reshape(B, *([S] * dim), nc).permute(0, dim + 1, *range(1, dim + 1))
    loss_g, acc_g = crit([lg[i] for i in range(B)], [d.to(dev) for d in data], [l.to(dev) for l in label],
                         None if weight is None else [w.to(dev) for w in weight])
    loss_g.backward()
    assert abs(float(loss_g.detach()) - float(loss_c.detach())) < 1e-5 * abs(float(loss_c.detach()))
    assert abs(float(acc_g) - float(acc_c)) < 1e-6
    gc = lc.grad.permute(0, *range(2, 2 + dim), 1).reshape(-1, nc)
    assert rel(rows.grad, gc) < 1e-5


def test_dense_step_replayed_from_a_graph_gives_the_eager_gradients(dev):
    """graphed.GraphedDenseStep: forward + loss + backward captured once (hipGraph) and replayed -- same loss, same
    parameter gradients as the eager step on the same inputs (bitwise: the same kernels in the same order), and a replay
    on NEW inputs equals the eager step on those"""
    from types import SimpleNamespace
    from uresnet_pytorch_amd.graphed import GraphedDenseStep
    from uresnet_pytorch_amd.iotools.synthetic import make_dense_blob
    from uresnet_pytorch_amd.models import DenseUResNet, DenseSegmentationLoss
    fl = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=2, SPATIAL_SIZE=16, NUM_CLASS=5, BN_MOMENTUM=0.9)
    torch.manual_seed(0)
    net = DenseUResNet(fl).to(dev).train()
    crit = DenseSegmentationLoss(fl)
    blobs = [make_dense_blob([s], 16, 3) for s in (0, 1)]
    D = [torch.from_numpy(b['data']).to(dev) for b in blobs]; Lb = [torch.from_numpy(b['label']).to(dev) for b in blobs]

    def eager(d, l):
        net.zero_grad(set_to_none=True)
        loss, _ = crit(net(d), d, l, None)
        loss.backward()
        return float(loss.detach()), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    gs = GraphedDenseStep(net, crit, D[0], Lb[0])
    for d, l in ((D[0], Lb[0]), (D[1], Lb[1])):
        loss_g, _ = gs(d, l)
        got = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
        lg = float(loss_g)
        le, ref = eager(d, l)
        assert abs(lg - le) <= 1e-6 * abs(le)
        assert set(got) == set(ref)
        for k in ref:
            assert rel(got[k], ref[k]) < 1e-6, k


@pytest.mark.parametrize('prec,tol', [('fp32', 1e-5), ('bf16', 1e-2)])
@pytest.mark.parametrize('dim,spatial,cin,cout,k,stride', [(3, (16, 16, 16), 16, 16, 3, 1), (3, (8, 12, 16), 32, 32, 3, 1), (3, (8, 8, 8), 16, 32, 3, 2),
                                                        (2, (20, 12), 8, 16, 3, 1), (3, (8, 8, 8), 64, 64, 1, 1)])
def test_dense_conv_with_folded_input_affine(dev, dim, spatial, cin, cout, k, stride, prec, tol):
    """xf = (scale, shift): the convolution and its weight gradient use x * scale + shift (the producer's BatchNorm folded into
    the load, urn_dense_conv / urn_dense_dw); the input gradient is the one w.r.t. that affine image (BNFoldFunction takes it
    through the BatchNorm).  Against torch on the CPU."""
    from uresnet_pytorch_amd import dense_conv as dc
    dc.set_precision(prec)
    try:
        B = 2
        g = torch.Generator().manual_seed(3 * cin + cout)
        x = torch.randn(B, cin, *spatial, generator=g)
        w = torch.randn(cout, cin, *([k] * dim), generator=g) / (cin * k ** dim) ** 0.5
        b = torch.randn(cout, generator=g)
        sc = torch.rand(cin, generator=g) + 0.5; sh = torch.randn(cin, generator=g)
        lo, hi = ref_padding(k, stride, spatial[-1])
        shape = (1, cin) + (1,) * dim
        u = (x * sc.view(shape) + sh.view(shape)).requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        up = F.pad(u, (lo, hi) * dim, mode='replicate') if (lo or hi) else u
        y_ref = (F.conv3d if dim == 3 else F.conv2d)(up, wr, b, stride=stride)
        dy = torch.randn(y_ref.shape, generator=g)
        y_ref.backward(dy)
        rows = to_rows(x).to(dev).requires_grad_(True)
        wg = w.to(dev).requires_grad_(True)
        y = dc.DenseConvFunction.apply(rows, wg, b.to(dev), B, spatial, stride, lo, hi, None, True, (sc.to(dev), sh.to(dev)))
        y.backward(to_rows(dy).to(dev))
        assert rel(y, to_rows(y_ref)) < tol
        assert rel(rows.grad, to_rows(u.grad)) < tol          # gradient w.r.t. the affine image
        assert rel(wg.grad, wr.grad) < tol
    finally:
        dc.set_precision('fp32')


def test_trainer_dense_graph_flag_matches_eager(dev):
    """trainval with flags -graph (the dense step replayed from a captured HIP graph) against the eager trainer: same
    initial weights, three train_step calls on changing batches -> same loss / accuracy, same parameters afterwards"""
    from types import SimpleNamespace
    from uresnet_pytorch_amd.iotools.synthetic import make_dense_blob
    from uresnet_pytorch_amd.trainval import trainval

    def flags(graph):
        return SimpleNamespace(MODEL_NAME='uresnet_dense', DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=2, SPATIAL_SIZE=16,
                               NUM_CLASS=5, BN_MOMENTUM=0.9, TRAIN=True, GPUS=[0], LEARNING_RATE=1e-3, MODEL_PATH='', WEIGHT_PREFIX='',
                               GRAPH=graph)
    blobs = []
    for s in range(3):
        b = make_dense_blob([2 * s, 2 * s + 1], 16, 3)
        blobs.append({'data': [[b['data'][0], b['data'][1]]], 'label': [[b['label'][0], b['label'][1]]]})
    out = {}
    for graph in (False, True):
        torch.manual_seed(0)
        t = trainval(flags(graph))
        t.initialize()
        res = [t.train_step(b, epoch=0., batch_size=2) for b in blobs]
        out[graph] = (res, torch.cat([p.detach().flatten() for p in t._net.parameters()]).cpu())
        assert (getattr(t, '_gstep', None) is not None) == graph
    for re, rg in zip(out[False][0], out[True][0]):
        assert abs(re['loss_seg'] - rg['loss_seg']) <= 1e-5 * abs(re['loss_seg'])
        assert abs(re['accuracy'] - rg['accuracy']) <= 1e-6
        assert len(rg['segmentation']) == 2 and rg['segmentation'][0].shape == re['segmentation'][0].shape
        assert rel(torch.from_numpy(rg['segmentation'][1]), torch.from_numpy(re['segmentation'][1])) < 1e-5
    assert rel(out[True][1], out[False][1]) < 1e-6


def test_dense_zero_bias_gradients_survive_accumulation_and_in_place_writes(dev):
    """The bias gradient of a convolution that a BatchNorm follows is exactly zero and comes out of ONE zero buffer
    (dense_conv._ZeroGrads) instead of a fill launch per convolution.  A .grad that outlives its step -- two passes without
    zero_grad, then zero_grad(set_to_none=False) -- must stay zero, never alias the statistics slabs of the next forward pass,
    and an in-place write into one .grad must not leak into the next step's gradients."""
    from types import SimpleNamespace
    from uresnet_pytorch_amd.iotools.synthetic import make_dense_blob
    from uresnet_pytorch_amd.models import DenseUResNet, DenseSegmentationLoss
    flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=2, SPATIAL_SIZE=16, NUM_CLASS=5, BN_MOMENTUM=0.9)
    torch.manual_seed(1)
    net = DenseUResNet(flags).to(dev).train()
    crit = DenseSegmentationLoss(flags)
    blob = make_dense_blob([0, 1], 16, 3)
    data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)

    def step():
        loss, _ = crit(net(data), data, label, None)
        loss.backward()
    # the conv biases in front of a BatchNorm: every conv bias except the last layer's
    names = [k for k, p in net.named_parameters() if k.endswith('.bias') and p.dim() == 1]
    step(); step()                                   # accumulation: the second pass adds onto the first pass' .grad
    grads = dict(net.named_parameters())
    zero_biases = [k for k in names if grads[k].grad is not None and float(grads[k].grad.abs().max()) == 0.0]
    assert len(zero_biases) >= 8, len(zero_biases)   # the convolutions that a BatchNorm follows
    w_ref = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    net.zero_grad(set_to_none=False)                 # the .grad tensors stay alive ...
    grads[zero_biases[0]].grad.add_(3.0)             # ... and someone writes into one in place
    net.zero_grad(set_to_none=True)
    step(); step()
    for k, p in net.named_parameters():
        if k in zero_biases:
            assert float(p.grad.abs().max()) == 0.0, k
        elif k in w_ref:
            assert rel(p.grad, w_ref[k]) < 1e-5, k


@pytest.mark.parametrize('prec,tol', [('fp32', 1e-5), ('bf16', 1e-2)])
def test_dense_big_strided_launch(dev, prec, tol):
    """A stride-2 convolution with >= 512 output tiles (4 x 64^3 x 16 -> 32^3 x 32): with bf16 operands the staged box of 16 row
    blocks fits and the launch takes them (urn_dense.hip: dense_strided_nrb), with fp32 operands it keeps 4 -- both against
    torch on the CPU, forward and both gradients."""
    from uresnet_pytorch_amd import dense_conv as dc
    dc.set_precision(prec)
    try:
        B, cin, cout, k, stride, spatial = 4, 16, 32, 3, 2, (64, 64, 64)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(B, cin, *spatial, generator=g)
        w = torch.randn(cout, cin, k, k, k, generator=g) / (cin * k ** 3) ** 0.5
        b = torch.randn(cout, generator=g)
        lo, hi = ref_padding(k, stride, spatial[-1])
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y_ref = F.conv3d(F.pad(xr, (lo, hi) * 3, mode='replicate'), wr, b, stride=stride)
        dy = torch.randn(y_ref.shape, generator=g)
        y_ref.backward(dy)
        rows = to_rows(x).to(dev).requires_grad_(True)
        wg = w.to(dev).requires_grad_(True)
        y = dc.DenseConvFunction.apply(rows, wg, b.to(dev), B, spatial, stride, lo, hi)
        y.backward(to_rows(dy).to(dev))
        assert rel(y, to_rows(y_ref)) < tol
        assert rel(rows.grad, to_rows(xr.grad)) < tol
        assert rel(wg.grad, wr.grad) < tol
    finally:
        dc.set_precision('fp32')
