"""One gather-conv shape, a few launches, for rocprofv3 --pmc runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uresnet_pytorch_amd import lib as L_, sparse_ops as so
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
L = L_.load(); dev = torch.device('cuda:0')
lv, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kern = int(sys.argv[4]) if len(sys.argv) > 4 else 4
L.urn_set_option(b'gconv_kernel', kern)
blob = make_sparse_blob([0], 512, 50000)
geo = so.SparseGeometry(torch.from_numpy(blob['data'][:, :4].astype(np.int32)).to(dev), 512, 5)
n = geo.n[lv]
x = torch.randn(n, cin, device=dev); wt = torch.randn(27, cout, cin, device=dev) * 0.05; y = torch.empty(n, cout, device=dev)
for _ in range(5):
    L_.check(L.urn_gconv_fwd(x.data_ptr(), wt.data_ptr(), geo.nbr[lv].data_ptr(), geo.ld, 27, 0, n, cin, cout, None, y.data_ptr(), L_.stream()))
torch.cuda.synchronize()
