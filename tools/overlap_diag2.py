"""Transient after the suffix hook is first used: per-10-step averages."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from uresnet_pytorch_amd import parallel
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=5, SPATIAL_SIZE=512, NUM_CLASS=5)
blob = make_sparse_blob([0], 512, 50000)
data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
torch.manual_seed(0)
net = SparseUResNet(flags).to(dev).train()
g = parallel.FlatGradients(net); opt = parallel.FlatAdam(g, lr=1e-3); crit = SparseSegmentationLoss(flags)
HOOK = [None]
def step():
    g.zero(); out = net(data); loss, _ = crit(out, [data], [label], None)
    net._executor.suffix_hook = HOOK[0]
    loss.backward()
    net._executor.suffix_hook = None
    opt.step()
def block(n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(10): step()
print('no hook :', ' '.join('%.2f' % block() for _ in range(4)), flush=True)
HOOK[0] = lambda off, side: None
print('hook    :', ' '.join('%.2f' % block() for _ in range(14)), flush=True)
HOOK[0] = None
print('no hook :', ' '.join('%.2f' % block() for _ in range(4)), flush=True)
HOOK[0] = lambda off, side: None
print('hook    :', ' '.join('%.2f' % block() for _ in range(6)), flush=True)
