"""Host timeline of one training step around the level-count synchronisation (us, mean over steps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np, torch
from uresnet_pytorch_amd import parallel, sparse_ops as so, trunk, lib as _l
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss
dev = torch.device('cuda:0')
flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=5, SPATIAL_SIZE=512, NUM_CLASS=5)
blob = make_sparse_blob([0], 512, 50000)
data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
torch.manual_seed(0)
net = SparseUResNet(flags).to(dev).train()
g = parallel.FlatGradients(net); opt = parallel.FlatAdam(g, lr=1e-3); crit = SparseSegmentationLoss(flags)
T = {}
def mark(k): T.setdefault(k, []).append(time.perf_counter())
orig_sync = so.SparseGeometry.sync
def sync(self):
    mark('sync_enter'); r = orig_sync(self); mark('sync_exit'); return r
so.SparseGeometry.sync = sync
def step():
    mark('step'); g.zero(); out = net(data); mark('fwd_done'); loss, _ = crit(out, [data], [label], None); mark('loss_done')
    loss.backward(); mark('bwd_done'); opt.step(); mark('opt_done')
for _ in range(10): step()
torch.cuda.synchronize(); T.clear()
for _ in range(40): step()
torch.cuda.synchronize()
def d(a, b): return 1e6 * float(np.mean(np.array(T[b]) - np.array(T[a])))
print('step start -> sync enter  %7.1f us (hidden behind the previous step)' % d('step', 'sync_enter'))
print('sync wait                 %7.1f us' % d('sync_enter', 'sync_exit'))
print('sync exit -> forward done %7.1f us (exposed: executor call incl. its launches, head)' % d('sync_exit', 'fwd_done'))
print('loss                      %7.1f us' % d('fwd_done', 'loss_done'))
print('backward enqueue          %7.1f us' % d('loss_done', 'bwd_done'))
print('optimizer                 %7.1f us' % d('bwd_done', 'opt_done'))
print('step                      %7.1f us' % (1e6 * float(np.mean(np.diff(np.array(T['step']))))))
