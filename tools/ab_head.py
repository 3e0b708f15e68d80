"""A/B in one process: the Linear head inside the executor (fused tail kernels) against separate launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np, torch
from uresnet_pytorch_amd import parallel
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss
dev = torch.device('cuda:0')
flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=5, SPATIAL_SIZE=512, NUM_CLASS=5)
torch.manual_seed(0)
model = SparseUResNet(flags).to(dev).train(); crit = SparseSegmentationLoss(flags)
blob = make_sparse_blob([0], 512, 50000)
data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
grads = parallel.FlatGradients(model); opt = parallel.FlatAdam(grads, lr=1e-3)
def step():
    grads.zero(); out = model(data); loss, _ = crit(out, [data], [label], None); loss.backward(); grads.all_reduce(); opt.step()
def run(n=30):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
res = {True: [], False: []}
for rnd in range(5):
    for fuse in (False, True):
        model.fuse_head = fuse
        res[fuse].append(run())
for fuse in (False, True):
    print('fuse_head=%s  min %.3f  median %.3f  all %s' % (fuse, min(res[fuse]), float(np.median(res[fuse])), ' '.join('%.3f' % v for v in res[fuse])))
