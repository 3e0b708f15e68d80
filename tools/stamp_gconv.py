"""Cycle stamps inside the LDS gather-conv kernel (diagnostic build, KS=4): where does an offset step spend its time?"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uresnet_pytorch_amd import lib as L_, sparse_ops as so
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
L = L_.load(); dev = torch.device('cuda:0')
blob = make_sparse_blob([0], 512, 50000)
geo = so.SparseGeometry(torch.from_numpy(blob['data'][:, :4].astype(np.int32)).to(dev), 512, 5)
for lv, cin, cout, rows_per_wg, waves in ((3, 64, 64, 32, 8), (0, 16, 16, 64, 4)):
    n = geo.n[lv]
    x = torch.randn(n, cin, device=dev); wt = torch.randn(27, cout, cin, device=dev) * 0.05; y = torch.empty(n, cout, device=dev)
    nwg = (n + rows_per_wg - 1) // rows_per_wg
    dbgbuf = torch.zeros(max(nwg * waves * 8, n * cout), dtype=torch.float32, device=dev)
    a = L_.GConvArgs(x=x.data_ptr(), wt=wt.data_ptr(), tbl=geo.nbr[lv].data_ptr(), ld=geo.ld, K=27, flip=0, n_out=n, cin=cin, cout=cout,
                     y=y.data_ptr(), res=dbgbuf.data_ptr())
    L.urn_set_option(b'gconv_kernel', 6); L.urn_set_option(b'gconv_dbg', 32)
    npart = ctypes.c_int()
    for _ in range(3):
        L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a), ctypes.byref(npart), L_.stream()))
    torch.cuda.synchronize()
    L.urn_set_option(b'gconv_dbg', 0)
    p = dbgbuf[:nwg * waves * 8].view(nwg * waves, 8).cpu().numpy()
    names = ['fetch issue', 'frag+mfma issue', 'park(+vm wait)', 'barrier']
    steps = np.maximum(p[:, 4], 1)
    print('level %d %d->%d: workgroups %d (%d waves), steps per wave %.1f' % (lv, cin, cout, nwg, waves, p[:, 4].mean()))
    for i, nm in enumerate(names):
        print('  %-18s %8.0f cycles per step (mean over waves)' % (nm, (p[:, i] / steps).mean()))
    print('  sum %.0f cycles per step' % (p[:, :4].sum(1) / steps).mean())
