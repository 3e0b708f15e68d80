"""BASELINE configs[1]-shaped run of the dense model on the GPU route (fp32 gather-conv kernels):
uresnet_dense -dd 3 -ss 128 -nc 5 -uf 16 -uns 5, one event per step; prints ms per fwd+loss+bwd step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np
import torch
from uresnet_pytorch_amd.iotools.synthetic import make_dense_blob
from uresnet_pytorch_amd import lib as L_
if os.environ.get('URN_LIB_PATH'): L_.LIB_PATH = os.environ['URN_LIB_PATH']   # an alternative build of the library (A/B of compile-time choices)
from uresnet_pytorch_amd.models import DenseUResNet, DenseSegmentationLoss

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
uns = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda:0')
PREC = ('fp32', 'bf16', 'fp16')[int(sys.argv[3])] if len(sys.argv) > 3 else 'fp32'   # MFMA operand precision (flags -prec)
flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=uns, SPATIAL_SIZE=S, NUM_CLASS=5, BN_MOMENTUM=0.9, PRECISION=PREC)
torch.manual_seed(0)
net = DenseUResNet(flags).to(dev).train()
crit = DenseSegmentationLoss(flags)
blob = make_dense_blob([0], S, 3)
data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
print('input', tuple(data.shape), 'params', sum(p.numel() for p in net.parameters()), flush=True)
def step():
    net.zero_grad(set_to_none=True)
    out = net(data)
    loss, acc = crit(out, data, label, None)
    loss.backward()
    return float(loss)
t0 = time.perf_counter(); l = step(); torch.cuda.synchronize(); print('first step %.1f s, loss %.4f' % (time.perf_counter() - t0, l), flush=True)
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print('S=%d: %.1f ms per fwd+loss+bwd step, %.2f M voxels/s, peak mem %.1f GB' % (S, dt * 1e3, S ** 3 / dt / 1e6, torch.cuda.max_memory_allocated() / 1e9))

# optional: the same step replayed from a captured graph (uresnet_pytorch_amd/graphed.py)
if os.environ.get('URN_GRAPH'):
    from uresnet_pytorch_amd.graphed import GraphedDenseStep
    gs = GraphedDenseStep(net, crit, data, label)
    gs(data, label); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): gl, _ = gs(data, label)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print('graph replay: %.1f ms per step, %.2f M voxels/s, loss %.4f' % (dt * 1e3, S ** 3 / dt / 1e6, float(gl)))
