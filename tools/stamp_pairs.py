"""Phase times inside the pair-list kernel (s_memtime stamps per wave): python tools/stamp_pairs.py <level> <cin> <cout> [xf]
Needs a library built with the stamps compiled in:
  touch uresnet_pytorch_amd/csrc/urn_gconv_pairs.hip && make -C uresnet_pytorch_amd/csrc EXTRA=-DURN_PAIRS_STAMP"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uresnet_pytorch_amd import lib as L_, sparse_ops as so
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
L = L_.load(); dev = torch.device('cuda:0')
lv, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
xf = len(sys.argv) > 4
blob = make_sparse_blob([0], 512, 50000)
geo = so.SparseGeometry(torch.from_numpy(blob['data'][:, :4].astype(np.int32)).to(dev), 512, 5)
n = geo.n[lv]; pl = geo.pairs['nbr'][lv]
x = torch.randn(n, cin, device=dev); wt = torch.randn(27, cout, cin, device=dev) * 0.05; y = torch.empty(n, cout, device=dev)
wf = torch.empty_like(wt)
L_.check(L.urn_weight_fragments(wt.data_ptr(), 27, cout, cin, wf.data_ptr(), L_.stream()))
sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev) * 0.1
a = L_.GConvArgs()
a.x = x.data_ptr(); a.wt = wt.data_ptr(); a.wt_frag = wf.data_ptr(); a.tbl = geo.nbr[lv].data_ptr(); a.ld = geo.ld; a.K = 27; a.n_out = n
a.cin = cin; a.cout = cout; a.y = y.data_ptr(); a.pairs = pl[0].data_ptr(); a.pairs_tile = pl[1]
if xf: a.xf_scale = sc.data_ptr(); a.xf_shift = sh.data_ptr()
for _ in range(5): L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a), None, L_.stream()))
st = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
L.urn_set_option(b'gconv_stamp_ptr', st.data_ptr())
L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a), None, L_.stream()))
torch.cuda.synchronize()
L.urn_set_option(b'gconv_stamp_ptr', 0)
s = st.cpu().numpy().reshape(-1, 8)
s = s[s[:, 0] != 0]
t0 = s[:, 0].min()
clk = 2.1e9   # __builtin_readcyclecounter = s_memtime: shader cycles (~2.1 GHz under this load; kernel span checked against HIP events)
us = lambda v: v / clk * 1e6
s = s[s[:, 4] != 0]
print('%d waves; entry spread %.2f us; kernel span %.2f us' % (len(s), us(s[:, 0].max() - t0), us(s[:, 4].max() - t0)))
names = ['prologue', 'block loop', 'barrier wait', 'epilogue']
for i, nm in enumerate(names):
    d = us(s[:, i + 1] - s[:, i])
    print('  %-13s mean %.2f  p10 %.2f  p90 %.2f  max %.2f us' % (nm, d.mean(), np.percentile(d, 10), np.percentile(d, 90), d.max()))
