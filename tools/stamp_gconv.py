"""Cycle stamps inside the LDS gather-conv kernel (diagnostic build, KS=4): where does an offset step spend its time?"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uresnet_pytorch_amd import lib as L_, sparse_ops as so
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
L = L_.load(); dev = torch.device('cuda:0')
blob = make_sparse_blob([0], 512, 50000)
geo = so.SparseGeometry(torch.from_numpy(blob['data'][:, :4].astype(np.int32)).to(dev), 512, 5)
lv, cin, cout = 3, 64, 64
n = geo.n[lv]
x = torch.randn(n, cin, device=dev); wt = torch.randn(27, cout, cin, device=dev) * 0.05; y = torch.empty(n, cout, device=dev)
nwg = (n + 63) // 64
part = torch.zeros(max(nwg * 8, L.urn_gconv_part_bytes(n, cout) // 8), dtype=torch.float64, device=dev)
a = L_.GConvArgs(x=x.data_ptr(), wt=wt.data_ptr(), tbl=geo.nbr[lv].data_ptr(), ld=geo.ld, K=27, flip=0, n_out=n, cin=cin, cout=cout,
                 y=y.data_ptr(), epilogue=1, part=part.data_ptr())
L.urn_set_option(b'gconv_kernel', 4); L.urn_set_option(b'gconv_dbg', 32)
npart = ctypes.c_int()
for _ in range(3):
    L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a), ctypes.byref(npart), L_.stream()))
torch.cuda.synchronize()
p = part[:nwg * 8].view(nwg, 8).cpu().numpy()
steps = p[:, 5].mean()
names = ['frag reads', 'fetch issue', 'mfma issue', 'park(+vm wait)', 'barrier']
print('workgroups %d, steps per workgroup %.1f' % (nwg, steps))
for i, nm in enumerate(names):
    print('%-16s %8.0f cycles per step' % (nm, (p[:, i] / np.maximum(p[:, 5], 1)).mean()))
print('sum %.0f cycles per step' % (p[:, :5].sum(1) / np.maximum(p[:, 5], 1)).mean())
