"""Phase times inside k_dense_conv3 (s_memtime stamps per wave): python tools/stamp_dense.py <S> <C> <prec>
Needs: touch uresnet_pytorch_amd/csrc/urn_dense.hip && make -C uresnet_pytorch_amd/csrc EXTRA=-DURN_DENSE_STAMP"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uresnet_pytorch_amd import lib as L_, dense_conv as dc
L = L_.load(); dev = torch.device('cuda:0')
S, c, prec = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
n = S ** 3
x = torch.randn(n, c, device=dev); w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
wt = w.reshape(c, c, -1).permute(2, 0, 1).contiguous(); y = torch.empty(n, c, device=dev)
Out, fwd, bwd, _ = dc.conv_geoms((S, S, S), 3, 1, 1, 1)
dc.set_precision(prec)
for _ in range(3): dc._launch(x, c, c, wt, None, y, c, c, 1, fwd)
st = torch.zeros(1 << 22, dtype=torch.int64, device=dev)
L.urn_set_option(b'dense_stamp_ptr', st.data_ptr())
dc._launch(x, c, c, wt, None, y, c, c, 1, fwd)
torch.cuda.synchronize()
L.urn_set_option(b'dense_stamp_ptr', 0)
s = st.cpu().numpy().reshape(-1, 8)
s = s[(s[:, 0] != 0) & (s[:, 4] != 0)]
clk = 2.1e9
us = lambda v: v / clk * 1e6
print('%d waves; kernel span %.1f us' % (len(s), us(s[:, 4].max() - s[:, 0].min())))
for i, nm in enumerate(['box staged (loads + LDS writes)', 'first weight slice + barrier', 'tap loops (+2 slices)', 'epilogue stores']):
    d = us(s[:, i + 1] - s[:, i])
    print('  %-34s mean %.2f  p10 %.2f  p90 %.2f  max %.2f us' % (nm, d.mean(), np.percentile(d, 10), np.percentile(d, 90), d.max()))
d = us(s[:, 4] - s[:, 0]); print('  wave lifetime mean %.2f us' % d.mean())
