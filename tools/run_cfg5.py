"""BASELINE configs[4]-like run: -ss 768, 200k voxels, -uf 32 -uns 7 (fp32 here): does the path scale, how long is a step?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np, torch
from uresnet_pytorch_amd import parallel, lib as L_
if os.environ.get('URN_LIB_PATH'): L_.LIB_PATH = os.environ['URN_LIB_PATH']   # diagnostic build (tools/build_diag_lib.sh)
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss
dev = torch.device('cuda:0')
PREC = ('fp32', 'bf16', 'fp16')[int(sys.argv[5])] if len(sys.argv) > 5 else 'fp32'   # MFMA operand precision (flags -prec)
S, n, m, Lv = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=m, URESNET_NUM_STRIDES=Lv, SPATIAL_SIZE=S, NUM_CLASS=5, PRECISION=PREC)
blob = make_sparse_blob([0], S, n)
data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
torch.manual_seed(0)
net = SparseUResNet(flags).to(dev).train()
print('params', sum(p.numel() for p in net.parameters()))
g = parallel.FlatGradients(net); opt = parallel.FlatAdam(g, lr=1e-3)
crit = SparseSegmentationLoss(flags)
def step():
    g.zero(); out = net(data); loss, _ = crit(out, [data], [label], None); loss.backward(); opt.step(); return loss
for _ in range(3): l = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print('S=%d n=%d uf=%d uns=%d: %.2f ms/step, %.2f M voxels/s, loss %.4f, mem %.2f GB' % (S, n, m, Lv, dt * 1e3, n / dt / 1e6, l.item(), torch.cuda.max_memory_allocated() / 1e9))
