"""Which framework-level operations (copies, fills, elementwise kernels) one cfg3 training step issues besides the
library's kernels: torch.profiler table of one step.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from torch.profiler import profile, ProfilerActivity
from uresnet_pytorch_amd import parallel
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss
dev = torch.device('cuda:0')
flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=5, SPATIAL_SIZE=512, NUM_CLASS=5)
blob = make_sparse_blob([0], 512, 50000)
data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
crit = SparseSegmentationLoss(flags)
torch.manual_seed(0)
net = SparseUResNet(flags).to(dev).train()
g = parallel.FlatGradients(net); opt = parallel.FlatAdam(g, lr=1e-3)
def step():
    g.zero(); out = net(data); loss, _ = crit(out, [data], [label], None); loss.backward(); g.all_reduce(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for _ in range(4): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cpu_time_total', row_limit=60, max_name_column_width=60))
