"""Where does the two-piece all-reduce spend its time at world size 1 (RCCL)?  Host time inside the suffix hook, step time with
the hook doing nothing, with the collective on the side stream's tail, and with it on the caller's stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
from types import SimpleNamespace
import torch, torch.distributed as dist
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
dist.init_process_group(backend='nccl', rank=0, world_size=1)
dist.all_reduce(g.flat)   # communicator warm-up
MODE = {'m': 'none'}
HOOK_T = []
works = []
def hook(offset, side_ptr):
    t0 = time.perf_counter()
    m = MODE['m']
    if m in ('side', 'side+prefix', 'side+prefix_sync'):
        with torch.cuda.stream(torch.cuda.ExternalStream(side_ptr, device=dev)):
            works.append(dist.all_reduce(g.flat[offset:], async_op=True))
    elif m == 'main':
        works.append(dist.all_reduce(g.flat[offset:], async_op=True))
    elif m == 'copy_side':     # a plain kernel on a third stream behind the side stream's tail, no RCCL
        s3 = STREAM3
        ev = torch.cuda.Event(); 
        with torch.cuda.stream(torch.cuda.ExternalStream(side_ptr, device=dev)):
            ev.record()
        s3.wait_event(ev)
        with torch.cuda.stream(s3):
            SCRATCH.copy_(g.flat[offset:])
    HOOK_T.append(time.perf_counter() - t0)
OFF = [0]
STREAM3 = torch.cuda.Stream()
SCRATCH = torch.empty_like(g.flat)[:0]
def step():
    g.zero(); out = net(data); loss, _ = crit(out, [data], [label], None)
    net._executor.suffix_hook = hook if MODE['m'] != 'off' else None
    loss.backward()
    net._executor.suffix_hook = None
    if MODE['m'] == 'side+prefix':
        works.append(dist.all_reduce(g.flat[:OFF[0]], async_op=True))
    if MODE['m'] == 'side+prefix_sync':
        dist.all_reduce(g.flat[:OFF[0]])
    for w in works: w.wait()
    works.clear()
    if MODE['m'] == 'copy_side': torch.cuda.current_stream().wait_stream(STREAM3)
    opt.step()
def timeit(n=40):
    for _ in range(5): step()
    torch.cuda.synchronize(); HOOK_T.clear(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
# size the scratch after the first backward told us the offset
MODE['m'] = 'none'; step(); torch.cuda.synchronize()
from uresnet_pytorch_amd import lib as _l
off = int(_l.load().urn_net_suffix_offset(net._executor.slots[0].handle))
SCRATCH = torch.empty(g.flat.numel() - off, device=dev)
OFF[0] = off
print('suffix: %d of %d floats' % (g.flat.numel() - off, g.flat.numel()))
for m in ('off', 'side', 'side+prefix', 'side+prefix_sync', 'main', 'none', 'off'):
    MODE['m'] = m
    ms = timeit()
    print('%-10s %.3f ms per step, host time in the hook %.1f us' % (m, ms, 1e6 * sum(HOOK_T) / max(len(HOOK_T), 1)), flush=True)
dist.destroy_process_group()
