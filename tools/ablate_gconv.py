"""Timing-only ablations of the 2-D tile gather-conv kernel (probe instantiations <4,2,4> at level 3 64->64 and
<1,4,1> at level 0 16->16): which part of an offset step is the chain made of?  Results of ablated runs are garbage."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uresnet_pytorch_amd import lib as L_, sparse_ops as so
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
L = L_.load(); dev = torch.device('cuda:0')
blob = make_sparse_blob([0], 512, 50000)
geo = so.SparseGeometry(torch.from_numpy(blob['data'][:, :4].astype(np.int32)).to(dev), 512, 5)
def run(level, cin, cout, reps=30):
    n = geo.n[level]
    x = torch.randn(n, cin, device=dev); wt = torch.randn(27, cout, cin, device=dev) * 0.05; y = torch.empty(n, cout, device=dev)
    def call():
        L_.check(L.urn_gconv_fwd(x.data_ptr(), wt.data_ptr(), geo.nbr[level].data_ptr(), geo.ld, 27, 0, n, cin, cout, None, y.data_ptr(), L_.stream()))
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
names = {0: 'full', 1: 'no MFMA', 2: 'no global loads', 4: 'no LDS parking', 8: 'no barrier', 3: 'no MFMA, no loads', 6: 'no loads, no parking',
         7: 'barriers + fragment reads only', 14: 'MFMA + fragment reads only', 15: 'fragment reads only'}
L.urn_set_option(b'gconv_kernel', 6)
for lv, ci, co in ((3, 64, 64), (0, 16, 16)):
    for abl, nm in names.items():
        L.urn_set_option(b'gconv_dbg', abl << 8)
        t = min(run(lv, ci, co) for _ in range(3))
        print('L%d %d->%d  %-32s %6.1f us' % (lv, ci, co, nm, t))
    L.urn_set_option(b'gconv_dbg', 16)
    print('L%d %d->%d  %-32s %6.1f us' % (lv, ci, co, 'no offsets (prologue+epilogue)', min(run(lv, ci, co) for _ in range(3))))
    L.urn_set_option(b'gconv_dbg', 0)
