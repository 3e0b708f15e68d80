"""Which Python lines issue device copies / fills in the sparse training step?  torch.profiler with stacks, three steps."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from torch.profiler import profile, ProfilerActivity
from uresnet_pytorch_amd import parallel
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss
dev = torch.device('cuda:0')
flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=5, SPATIAL_SIZE=512, NUM_CLASS=5)
torch.manual_seed(0)
model = SparseUResNet(flags).to(dev).train(); crit = SparseSegmentationLoss(flags)
blob = make_sparse_blob([0], 512, 50000)
data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
grads = parallel.FlatGradients(model); opt = parallel.FlatAdam(grads, lr=1e-3)
def step():
    grads.zero(); out = model(data); loss, _ = crit(out, [data], [label], None); loss.backward(); grads.all_reduce(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(N): step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::clone', 'aten::_to_copy', 'aten::contiguous', 'aten::index_add_', 'aten::cat'):
        st = [s for s in (e.stack or []) if 'uresnet_pytorch_amd' in s or 'find_copies' in s]
        cnt[(e.name, (st[0] if st else ' | '.join((e.stack or ['?'])[:3])) + '  shapes ' + str(e.input_shapes))] += 1
for (name, where), c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print('%5.1f per step  %-16s %s' % (c / N, name, where))


# device-side memcpy / memset / tiny blit launches and the CPU op that issued them
cnt2 = collections.Counter()
for e in prof.events():
    ks = getattr(e, 'kernels', None) or []
    for k in ks:
        if 'emcpy' in k.name or 'emset' in k.name or 'copyBuffer' in k.name or 'fillBuffer' in k.name:
            st = [x for x in (e.stack or []) if 'uresnet_pytorch_amd' in x or 'find_copies' in x or 'bench' in x]
            cnt2[(k.name[:40], e.name, st[0] if st else ' | '.join((e.stack or ['?'])[:2]), str(e.input_shapes)[:60])] += 1
print('--- device copies by issuing op')
for k, c in sorted(cnt2.items(), key=lambda kv: -kv[1]):
    print('%5.1f per step  %s' % (c / N, ' || '.join(k)))
