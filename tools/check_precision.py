"""bf16 / fp16 MFMA operands of the gather convolution against fp32: norm-wise relative error and time per launch."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uresnet_pytorch_amd import lib as L_, sparse_ops as so
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
L = L_.load(); dev = torch.device('cuda:0')
blob = make_sparse_blob([0], 512, 50000)
geo = so.SparseGeometry(torch.from_numpy(blob['data'][:, :4].astype(np.int32)).to(dev), 512, 5)
torch.manual_seed(0)
def run(a, reps=30):
    for _ in range(3): L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a), None, L_.stream()))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a), None, L_.stream()))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for lv, ci, co in ((0, 16, 16), (1, 32, 32), (2, 48, 48), (3, 64, 64), (4, 80, 80), (3, 128, 64), (4, 160, 80), (1, 16, 32)):
    n = geo.n[lv]
    x = torch.randn(n, ci, device=dev); wt = torch.randn(27, co, ci, device=dev) * 0.05
    sc = torch.rand(ci, device=dev) + 0.5; sh = torch.randn(ci, device=dev) * 0.1
    res = {}
    for name, prec in (('fp32', 1), ('bf16', 2), ('fp16', 3)):
        y = torch.empty(n, co, device=dev)
        a = L_.GConvArgs(x=x.data_ptr(), wt=wt.data_ptr(), tbl=geo.nbr[lv].data_ptr(), ld=geo.ld, K=27, flip=0, n_out=n, cin=ci, cout=co,
                         y=y.data_ptr(), xf_scale=sc.data_ptr(), xf_shift=sh.data_ptr(), precision=prec)
        t = min(run(a) for _ in range(3))
        res[name] = (y, t)
    ref = res['fp32'][0].double()
    print('L%d %3d->%3d  fp32 %5.1f us | bf16 %5.1f us rel err %.1e | fp16 %5.1f us rel err %.1e' % (
        lv, ci, co, res['fp32'][1], res['bf16'][1], float((res['bf16'][0].double() - ref).norm() / ref.norm()),
        res['fp16'][1], float((res['fp16'][0].double() - ref).norm() / ref.norm())), flush=True)
print('weight gradient:')
def run_dw(args, reps=20):
    for _ in range(2): L_.check(L.urn_gconv_bwd_dw(*args))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L_.check(L.urn_gconv_bwd_dw(*args))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for lv, ci, co in ((0, 16, 16), (1, 32, 32), (1, 16, 32), (2, 48, 48), (3, 64, 64), (4, 80, 80), (3, 128, 64), (4, 160, 80)):
    n = geo.n[lv]
    x = torch.randn(n, ci, device=dev); dy = torch.randn(n, co, device=dev)
    res = {}
    for name, prec in (('fp32', 0), ('bf16', 1), ('fp16', 2)):
        L.urn_set_option(b'gconv_precision', prec)
        dw = torch.zeros(27, ci, co, device=dev)
        args = (x.data_ptr(), dy.data_ptr(), geo.nbr[lv].data_ptr(), geo.ld, 27, n, ci, co, dw.data_ptr(), L_.stream())
        L_.check(L.urn_gconv_bwd_dw(*args)); torch.cuda.synchronize()
        first = dw.clone()
        res[name] = (first, min(run_dw(args) for _ in range(3)))
    L.urn_set_option(b'gconv_precision', 0)
    ref = res['fp32'][0].double()
    print('L%d %3d->%3d  fp32 %5.1f us | bf16 %5.1f us rel err %.1e | fp16 %5.1f us rel err %.1e' % (
        lv, ci, co, res['fp32'][1], res['bf16'][1], float((res['bf16'][0].double() - ref).norm() / ref.norm()),
        res['fp16'][1], float((res['fp16'][0].double() - ref).norm() / ref.norm())), flush=True)
