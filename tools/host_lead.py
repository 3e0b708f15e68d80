"""How far the host runs ahead of the GPU in the cfg3 training step: per step, the time at which the host has enqueued
everything against the time at which the GPU finishes it (positive lead = the GPU still has queued work when the host
starts on the next step; a lead below the host time of the next step's first launches means idle GPU time).
Also times the host side of the phases of one step (geometry, sync wait, forward enqueue, backward enqueue)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from uresnet_pytorch_amd import parallel
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss

dev = torch.device('cuda:0')
flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=5, SPATIAL_SIZE=512, NUM_CLASS=5)
torch.manual_seed(0)
model = SparseUResNet(flags).to(dev).train()
crit = SparseSegmentationLoss(flags)
blob = make_sparse_blob([0], 512, 50000)
data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
grads = parallel.FlatGradients(model); opt = parallel.FlatAdam(grads, lr=1e-3)
marks = []
def step():
    t = [time.perf_counter()]
    grads.zero(); out = model(data); t.append(time.perf_counter())
    loss, _ = crit(out, [data], [label], None); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    grads.all_reduce(); opt.step(); t.append(time.perf_counter())
    marks.append(t)
for _ in range(5): step()
torch.cuda.synchronize()
N = 30
marks.clear()
ev0 = torch.cuda.Event(enable_timing=True); ev0.record(); torch.cuda.synchronize(); t0 = time.perf_counter()
evs, hd = [], []
for _ in range(N):
    step()
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e); hd.append(time.perf_counter())
torch.cuda.synchronize()
tend = time.perf_counter()
print('ms/step %.3f' % ((tend - t0) * 1e3 / N))
lead = [(ev0.elapsed_time(e) - (h - t0) * 1e3) for e, h in zip(evs, hd)]
print('GPU finish - host enqueue-done, ms, per step:', ' '.join('%.2f' % v for v in lead))
import numpy as np
m = np.array(marks)
print('host ms per step: forward(model) %.3f  loss %.3f  backward %.3f  allreduce+adam %.3f  total %.3f' % tuple(
    list(np.median(np.diff(m, axis=1), axis=0) * 1e3) + [float(np.median(m[:, -1] - m[:, 0]) * 1e3)]))
