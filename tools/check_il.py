"""Interleaved tile kernel against the plain one: same products, accumulator assignment differs (1e-6 relative expected)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uresnet_pytorch_amd import lib as L_, sparse_ops as so
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
L = L_.load(); dev = torch.device('cuda:0')
blob = make_sparse_blob([0], 512, 50000)
geo = so.SparseGeometry(torch.from_numpy(blob['data'][:, :4].astype(np.int32)).to(dev), 512, 5)
torch.manual_seed(0)
for lv, ci, co in ((2, 48, 48), (3, 64, 64), (4, 80, 80), (3, 128, 64), (2, 96, 48), (4, 160, 80), (3, 64, 128), (1, 64, 32)):
    n = geo.n[lv]
    x = torch.randn(n, ci, device=dev); wt = torch.randn(27, co, ci, device=dev) * 0.05
    sc = torch.rand(ci, device=dev) + 0.5; sh = torch.randn(ci, device=dev) * 0.1
    outs = []
    for il in (0, 1):
        for xf in (0, 1):
            L.urn_set_option(b'tile_il', il)
            y = torch.empty(n, co, device=dev)
            a = L_.GConvArgs(x=x.data_ptr(), wt=wt.data_ptr(), tbl=geo.nbr[lv].data_ptr(), ld=geo.ld, K=27, flip=0, n_out=n, cin=ci, cout=co,
                             y=y.data_ptr(), xf_scale=sc.data_ptr() if xf else None, xf_shift=sh.data_ptr() if xf else None)
            L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a), None, L_.stream()))
            outs.append(y)
    torch.cuda.synchronize()
    print('L%d %d->%d max rel diff plain vs interleaved: raw %.2e, xf %.2e' % (lv, ci, co, float((outs[0] - outs[2]).abs().max() / outs[0].abs().max()), float((outs[1] - outs[3]).abs().max() / outs[1].abs().max())), flush=True)
L.urn_set_option(b'tile_il', 1)
