"""Does an RCCL communicator (its internal streams) in the process disturb the executor's two-stream schedule?
One rank, backend nccl, one all-reduce of the flat gradient buffer per step -- against the same loop without it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
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
ov = parallel.OverlappedAllReduce(g, force=True)
def step(reduce):
    g.zero(); out = net(data); loss, _ = crit(out, [data], [label], None)
    if reduce == 2: ov.arm(net)          # the two-piece all-reduce: suffix from inside the backward pass, behind the side stream
    loss.backward()
    if reduce == 2: ov.finish()
    elif reduce: dist.all_reduce(g.flat, op=dist.ReduceOp.SUM)
    opt.step()
def timeit(reduce, n=40):
    for _ in range(5): step(reduce)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step(reduce)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('before init_process_group: %.3f ms per step' % timeit(False), flush=True)
dist.init_process_group(backend='nccl', rank=0, world_size=1)
print('communicator present, no collective: %.3f ms per step' % timeit(False), flush=True)
print('one all-reduce (11 MB, world 1) per step: %.3f ms per step' % timeit(True), flush=True)
print('two-piece all-reduce (suffix overlapped with the backward pass), world 1: %.3f ms per step' % timeit(2), flush=True)
print('again without: %.3f ms per step' % timeit(False), flush=True)
dist.destroy_process_group()
