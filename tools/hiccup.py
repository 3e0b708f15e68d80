"""Sporadic long steps: per-step wall time (synchronised) over many steps, with the spikes listed; then the same with the
Python garbage collector disabled.  Also counts device allocations of the caching allocator."""
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
def step():
    g.zero(); out = net(data); loss, _ = crit(out, [data], [label], None); loss.backward(); opt.step()
def run(n, tag, sync_every=1):
    ts = []
    torch.cuda.synchronize()
    a0 = torch.cuda.memory_stats().get('num_device_alloc', 0)
    t0 = time.perf_counter()
    for i in range(n):
        step()
        if (i + 1) % sync_every == 0:
            torch.cuda.synchronize(); t1 = time.perf_counter(); ts.append((t1 - t0) / sync_every * 1e3); t0 = t1
    ts = np.array(ts)
    spikes = [(int(i * sync_every), round(float(v), 2)) for i, v in enumerate(ts) if v > 2 * np.median(ts)]
    print('%s: median %.3f ms, mean %.3f, %d spikes %s, device allocations during the run: %d, gc counts %s' % (
        tag, np.median(ts), ts.mean(), len(spikes), spikes[:12], torch.cuda.memory_stats().get('num_device_alloc', 0) - a0, gc.get_count()), flush=True)
for _ in range(10): step()
run(400, 'gc on, sync every step')
run(400, 'gc on, sync every 10', 10)
gc.collect(); gc.disable()
run(400, 'gc OFF, sync every 10', 10)
gc.enable()
gc.freeze()
run(400, 'gc on after gc.freeze(), sync every 10', 10)
