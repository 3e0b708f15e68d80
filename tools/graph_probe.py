"""Feasibility probe: is the cfg3 training step faster replayed from a captured HIP graph than issued eagerly?  The step is
captured with the level counts of the (repeated) bench event handed over as a hint, so that nothing synchronises inside."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np, torch
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
def fb():
    g.zero(); out = net(data); loss, _ = crit(out, [data], [label], None); loss.backward(); return loss
def step():
    l = fb(); opt.step(); return l
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(5): step()
gc.collect(); gc.freeze()
print('eager: %.3f ms per step' % timeit(step), flush=True)
net._counts_hint = list(net._last_geo.n)
print('eager with the counts handed over (no read-back): %.3f ms per step' % timeit(step), flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): fb()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    loss_g = fb()
torch.cuda.synchronize()
def gstep():
    graph.replay(); opt.step()
print('graph replay: %.3f ms per step (loss %.5f)' % (timeit(gstep), float(loss_g)), flush=True)
net._counts_hint = None
print('eager again: %.3f ms per step' % timeit(step), flush=True)
