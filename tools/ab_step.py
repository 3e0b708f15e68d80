"""A/B of executor switches in ONE process (interleaved rounds): fused vs unfused BatchNorm,
weight gradients on a side stream vs single stream.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np
import torch
from uresnet_pytorch_amd import parallel
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss

dev = torch.device('cuda:0')
flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=5, SPATIAL_SIZE=512, NUM_CLASS=5)
blob = make_sparse_blob([0], 512, 50000)
data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
crit = SparseSegmentationLoss(flags)
variants = {}
from uresnet_pytorch_amd import lib as _lib
L = _lib.load()
for name, fl in (('default', 0),):
    torch.manual_seed(0)
    net = SparseUResNet(flags).to(dev).train(); net.executor_flags = fl
    g = parallel.FlatGradients(net); opt = parallel.FlatAdam(g, lr=1e-3)
    def step(net=net, g=g, opt=opt):
        g.zero(); out = net(data); loss, _ = crit(out, [data], [label], None); loss.backward(); opt.step()
    variants[name] = step
for s in variants.values():
    for _ in range(5): s()
res = {k: [] for k in variants}
kernels = {'min wgs 100': 100, 'min wgs 150': 150, 'min wgs 200': 200, 'min wgs 250': 250, 'min wgs 300': 300}
res = {(k, kn): [] for k in variants for kn in kernels}
for rnd in range(4):
    for kn, kv in kernels.items():
        L.urn_set_option(b'tile_min_wgs', kv)
        for k, s in variants.items():
            for _ in range(2): s()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): s()
            torch.cuda.synchronize(); res[(k, kn)].append((time.perf_counter() - t0) / 20 * 1e3)
for k, v in res.items():
    print('%-28s median %.3f ms  min %.3f ms' % (str(k), float(np.median(v)), min(v)))
