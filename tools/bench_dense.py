"""Micro-benchmark of the dense implicit-GEMM kernels at the cfg2 level shapes (one event): forward conv k3 s1 and its
weight gradient, fp32 and bf16 operands."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uresnet_pytorch_amd import lib as L_, dense_conv as dc
L = L_.load(); dev = torch.device('cuda:0')
def timeit(call, reps=5):
    for _ in range(2): call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for S, c in [(128, 16), (64, 32), (32, 64), (16, 128), (8, 256), (64, 16), (128, 32)]:
    n = S ** 3
    x = torch.randn(n, c, device=dev); w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
    wt = w.reshape(c, c, -1).permute(2, 0, 1).contiguous()
    y = torch.empty(n, c, device=dev); dy = torch.randn(n, c, device=dev)
    Out, fwd, bwd, _ = dc.conv_geoms((S, S, S), 3, 1, 1, 1)
    fl = 2.0 * n * 27 * c * c
    out = []
    for prec in ('fp32', 'bf16'):
        dc.set_precision(prec)
        t = timeit(lambda: dc._launch(x, c, c, wt, None, y, c, c, 1, fwd))
        tw = timeit(lambda: dc._dw_call(x, c, dy, c, 1, [S, S, S], [S, S, S], [3, 3, 3], [1, 1, 1], [1, 1, 1], 0, c, c, (c, c, 3, 3, 3)))
        out.append('%s fwd %.0f us (%.0f TF/s) dW %.0f us (%.0f TF/s)' % (prec, t, fl / t / 1e6, tw, fl / tw / 1e6))
    print('%3d^3 %3d->%3d  %s' % (S, c, c, ' | '.join(out)), flush=True)
