"""Per-parameter gradient parity of the HIP path vs the oracle at cfg3 (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np
import torch
from oracle import sparse_oracle as orc
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss

S, m, L, nc = 512, 16, 5, 5
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dev = torch.device('cuda:0')
blob = make_sparse_blob([0], S, n)
flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=m, URESNET_NUM_STRIDES=L, SPATIAL_SIZE=S, NUM_CLASS=nc)
P = orc.init_params(m, L, nc, seed=1)
net = SparseUResNet(flags)
sd = net.state_dict(); sd.update({k: torch.from_numpy(v) for k, v in P.items()}); net.load_state_dict(sd)
net = net.to(dev).train()
data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
crit = SparseSegmentationLoss(flags)
for it in range(3):
    net.zero_grad()
    torch.cuda.synchronize(); t0 = time.time()
    out = net(data); loss, acc = crit(out, [data], [label], None); loss.backward()
    torch.cuda.synchronize(); print('step %d: %.2f ms' % (it, 1e3 * (time.time() - t0)))
# capture BN outputs of the GPU model to count ReLU-mask flips against the oracle
from uresnet_pytorch_amd import scn
gpu_acts = {}
hooks = [mod.register_forward_hook(lambda mo, i, o, name=name: gpu_acts.__setitem__(name, o.features.detach()))
         for name, mod in net.named_modules() if isinstance(mod, scn.BatchNormLeakyReLU)]
net.zero_grad(); out = net(data); loss, acc = crit(out, [data], [label], None); loss.backward()
for h in hooks: h.remove()
ref = orc.SparseUResNetOracle(P, m, L, nc, S); ref.keep_acts = True
t0 = time.time(); logits = ref.forward(blob['data'])
loss_ref, _, dl = orc.segmentation_loss(logits, blob['data'], blob['label'])
G, _ = ref.backward(dl); print('oracle fwd+bwd %.2f s, threads %d' % (time.time() - t0, orc.lib().orc_num_threads()))
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))
flips = 0
for k, y in ref.acts.items():
    f = int(((gpu_acts[k].cpu().numpy() > 0) != (y > 0)).sum()); flips += f
    if f: print('ReLU mask flips in %s: %d of %d' % (k, f, y.size))
print('total ReLU mask flips GPU vs oracle: %d' % flips)
print('logits', rel(out[0].detach().cpu().numpy(), logits), 'loss', loss.item(), loss_ref)
for k, p in net.named_parameters():
    print('%-40s %-16s %.2e  |g|=%.3e' % (k, tuple(p.shape), rel(p.grad.cpu().numpy(), G[k]), np.linalg.norm(G[k])))
