"""Forward-only (inference, model.eval(), no autograd) time per 50k-voxel event: executor against the per-layer path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
from uresnet_pytorch_amd.models import SparseUResNet
dev = torch.device('cuda:0')
flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=5, SPATIAL_SIZE=512, NUM_CLASS=5)
blob = make_sparse_blob([0], 512, 50000)
data = torch.from_numpy(blob['data']).to(dev)
torch.manual_seed(0)
net = SparseUResNet(flags).to(dev).train()
net(data)                                   # one training forward: running statistics
net.eval()
for name, use in (('executor', True), ('per-layer', False)):
    net.use_executor = use
    with torch.no_grad():
        for _ in range(5): net(data)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): net(data)
        torch.cuda.synchronize()
    print('%-10s %.2f ms per event (%.1f M voxels/s)' % (name, (time.perf_counter() - t0) / 30 * 1e3, 50000 / ((time.perf_counter() - t0) / 30) / 1e6))
