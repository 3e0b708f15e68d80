"""one pair-list convolution shape repeated (for rocprofv3 --pmc): python tools/bench_pairs_one.py <level> <cin> <cout> [dbg]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uresnet_pytorch_amd import lib as L_, sparse_ops as so
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
L = L_.load(); dev = torch.device('cuda:0')
lv, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if len(sys.argv) > 4: L.urn_set_option(b'gconv_dbg', int(sys.argv[4]))
blob = make_sparse_blob([0], 512, 50000)
geo = so.SparseGeometry(torch.from_numpy(blob['data'][:, :4].astype(np.int32)).to(dev), 512, 5)
n = geo.n[lv]; pl = geo.pairs['nbr'][lv]
x = torch.randn(n, cin, device=dev); wt = torch.randn(27, cout, cin, device=dev) * 0.05; y = torch.empty(n, cout, device=dev)
wf = torch.empty_like(wt)
L_.check(L.urn_weight_fragments(wt.data_ptr(), 27, cout, cin, wf.data_ptr(), L_.stream()))
a = L_.GConvArgs()
a.x = x.data_ptr(); a.wt = wt.data_ptr(); a.wt_frag = wf.data_ptr(); a.tbl = geo.nbr[lv].data_ptr(); a.ld = geo.ld; a.K = 27; a.n_out = n
a.cin = cin; a.cout = cout; a.y = y.data_ptr(); a.pairs = pl[0].data_ptr(); a.pairs_tile = pl[1]
for _ in range(20):
    L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a), None, L_.stream()))
torch.cuda.synchronize()
