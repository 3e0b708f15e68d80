"""Test-side seams for pinning ReLU branches of the dense model (tests/test_dense_golden.py): nothing in the product module
knows about them.  A pre-activation within fp32 rounding of zero takes different branches in any two evaluation orders, and one
such element moves a per-channel gradient sum by ~1e-2 at these layer sizes -- which says nothing about the arithmetic."""
import contextlib

from uresnet_pytorch_amd import dense_ops as D


@contextlib.contextmanager
def cpu_relu(fn):
    """replace the CPU route's ReLU by fn(pre-activation) for the duration of the block"""
    orig = D._relu
    D._relu = fn
    try:
        yield
    finally:
        D._relu = orig


@contextlib.contextmanager
def gpu_relu_record(fn):
    """call fn(output) with every ReLU output of the GPU route (the fused kernels apply the ReLU themselves: the output is
    what can be observed) for the duration of the block"""
    impl = D._gpu()
    orig_c, orig_t = impl.conv_bn_act, impl.convT_bn_act

    def conv_bn_act(x, w, b, stride, pad, gamma, beta, eps, relu, residual=None, defer=False):
        y = orig_c(x, w, b, stride, pad, gamma, beta, eps, relu, residual, defer)
        if relu:
            fn(y)
        return y

    def convT_bn_act(x, w, b, gamma, beta, eps, relu):
        y = orig_t(x, w, b, gamma, beta, eps, relu)
        if relu:
            fn(y)
        return y
    impl.conv_bn_act, impl.convT_bn_act = conv_bn_act, convT_bn_act
    try:
        yield
    finally:
        impl.conv_bn_act, impl.convT_bn_act = orig_c, orig_t
