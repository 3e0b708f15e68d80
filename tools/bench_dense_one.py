"""one dense conv shape repeated (for rocprofv3 --pmc): python tools/bench_dense_one.py S C prec"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uresnet_pytorch_amd import lib as L_, dense_conv as dc
L = L_.load(); dev = torch.device('cuda:0')
S, c, prec = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
n = S ** 3
x = torch.randn(n, c, device=dev); w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
wt = w.reshape(c, c, -1).permute(2, 0, 1).contiguous(); y = torch.empty(n, c, device=dev); dy = torch.randn(n, c, device=dev)
Out, fwd, bwd, _ = dc.conv_geoms((S, S, S), 3, 1, 1, 1)
dc.set_precision(prec)
for _ in range(4):
    dc._launch(x, c, c, wt, None, y, c, c, 1, fwd)
    dc._dw_call(x, c, dy, c, 1, [S, S, S], [S, S, S], [3, 3, 3], [1, 1, 1], [1, 1, 1], 0, c, c, (c, c, 3, 3, 3))
torch.cuda.synchronize()
