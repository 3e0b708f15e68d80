"""Which Python lines issue the torch copy / add / fill / cat kernels of the dense cfg2 step (bf16 operands)?  torch.profiler with
stacks, two steps; kernels of >= 10 us listed with the issuing op, its shapes and the first package frame."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from torch.profiler import profile, ProfilerActivity
from uresnet_pytorch_amd.iotools.synthetic import make_dense_blob
from uresnet_pytorch_amd.models import DenseUResNet, DenseSegmentationLoss
dev = torch.device('cuda:0')
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=5, SPATIAL_SIZE=S, NUM_CLASS=5, BN_MOMENTUM=0.9, PRECISION='bf16')
torch.manual_seed(0)
net = DenseUResNet(flags).to(dev).train(); crit = DenseSegmentationLoss(flags)
blob = make_dense_blob([0], S, 3)
data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
def step():
    net.zero_grad(set_to_none=True); loss, _ = crit(net(data), data, label, None); loss.backward()
for _ in range(3): step()
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(N): step()
    torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    for k in (getattr(e, 'kernels', None) or []):
        if not any(t in k.name for t in ('at::native', 'rocclr', 'Cat')): continue
        st = [x for x in (e.stack or []) if 'uresnet_pytorch_amd' in x or 'find_copies' in x]
        key = (e.name, str(e.input_shapes)[:70], (st[0] if st else ' | '.join((e.stack or ['?'])[:2]))[-90:])
        rows[key][0] += 1; rows[key][1] += k.duration
for key, (c, us) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%7.1f us/step %5.1f/step  %-22s %-70s %s' % (us / N, c / N, key[0], key[1], key[2]))
