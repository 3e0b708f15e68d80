"""Is the training step GPU-bound or launch-bound?  Enqueue time of a step (host, no sync) against its GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np, torch
from uresnet_pytorch_amd import parallel
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss
dev = torch.device('cuda:0')
flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=5, SPATIAL_SIZE=512, NUM_CLASS=5)
blob = make_sparse_blob([0], 512, 50000)
data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
torch.manual_seed(0)
net = SparseUResNet(flags).to(dev).train()
g = parallel.FlatGradients(net); opt = parallel.FlatAdam(g, lr=1e-3); crit = SparseSegmentationLoss(flags)
def step():
    g.zero(); out = net(data); loss, _ = crit(out, [data], [label], None); loss.backward(); opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue %.3f ms per step (host), total %.3f ms per step; the host is %s' % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3,
      'ahead of the GPU' if (t2 - t1) > 0.1 * (t2 - t0) else 'the bottleneck (GPU drains as fast as it is fed)'))
# host time of the pieces
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
