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
