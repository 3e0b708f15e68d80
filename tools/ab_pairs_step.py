"""In-step A/B of the pair-list dispatch policy: whole cfg3 training steps, policies interleaved in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np, torch
from uresnet_pytorch_amd import lib as L_, parallel, sparse_ops as so
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss
L = L_.load(); dev = torch.device('cuda:0')
flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=5, SPATIAL_SIZE=512, NUM_CLASS=5)
torch.manual_seed(0)
model = SparseUResNet(flags).to(dev).train(); crit = SparseSegmentationLoss(flags)
blob = make_sparse_blob([0], 512, 50000)
data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
grads = parallel.FlatGradients(model); opt = parallel.FlatAdam(grads, lr=1e-3)
def step():
    grads.zero(); out = model(data); loss, _ = crit(out, [data], [label], None); loss.backward(); grads.all_reduce(); opt.step()
def run(n=20):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
policies = [('cin<=80 (default)', 80, 999, 0, 7), ('cin<=80, dW slabs', 80, 999, 0, 7, 'slabs'), ('cin<=80, dW pairs', 80, 999, 0, 7, 'pairs')]
res = {p[0]: [] for p in policies}
for rnd in range(3):
    for name, mi, mo, nin, k, *rest in policies:
        so.set_deterministic_dw(bool(rest), rest[0] if rest else 'slabs')
        L.urn_set_option(b'gconv_kernel', k); L.urn_set_option(b'pairs_max_cin', mi); L.urn_set_option(b'pairs_max_cout', mo); L.urn_set_option(b'pairs_nin', nin)
        res[name].append(run())
for name, *_ in policies:
    print('%-24s ms/step min %.3f  all %s' % (name, min(res[name]), ' '.join('%.3f' % v for v in res[name])), flush=True)
