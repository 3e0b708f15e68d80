"""Step time of the k-th executor alive in one process (every one with its own side stream): HIP maps streams onto
GPU_MAX_HW_QUEUES hardware queues (default 4).  Usage: [GPU_MAX_HW_QUEUES=8] python tools/many_executors.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from uresnet_pytorch_amd import parallel, trunk
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss
trunk.MAX_SIDE_STREAMS = 16
from uresnet_pytorch_amd import lib as _lib
_lib.load().urn_set_option(b"net_side_verbose", 1)
_lib.load().urn_set_option(b"net_side_probe", int(os.environ.get("PROBE", "4")))
dev = torch.device('cuda:0')
flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=5, SPATIAL_SIZE=512, NUM_CLASS=5)
blob = make_sparse_blob([0], 512, 50000)
data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
crit = SparseSegmentationLoss(flags)
steps = []
for k in range(6):
    torch.manual_seed(0)
    net = SparseUResNet(flags).to(dev).train()
    g = parallel.FlatGradients(net); opt = parallel.FlatAdam(g, lr=1e-3)
    def step(net=net, g=g, opt=opt):
        g.zero(); out = net(data); loss, _ = crit(out, [data], [label], None); loss.backward(); opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    print('executor %d: %.3f ms per step (GPU_MAX_HW_QUEUES=%s)' % (k + 1, (time.perf_counter() - t0) / 20 * 1e3, os.environ.get('GPU_MAX_HW_QUEUES', 'default')), flush=True)
    steps.append(step)
