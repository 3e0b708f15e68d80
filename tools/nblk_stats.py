import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from uresnet_pytorch_amd import lib as L_, sparse_ops as so
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
L = L_.load(); dev = torch.device('cuda:0')
blob = make_sparse_blob([0], 512, 50000)
geo = so.SparseGeometry(torch.from_numpy(blob['data'][:, :4].astype(np.int32)).to(dev), 512, 5)
for l in range(5):
    t, T = geo.pairs['nbr'][l]
    words = L.urn_pairs_bytes(geo.cap, 27, T) // 4 // ((geo.cap + T - 1) // T)
    ntiles = (geo.n[l] + T - 1) // T
    nb = t.view(-1, words)[:ntiles, 0].cpu().numpy()
    print('level %d: tiles %d  blocks per tile min %d  mean %.1f  median %d  p90 %d  max %d' % (l, ntiles, nb.min(), nb.mean(), np.median(nb), np.percentile(nb, 90), nb.max()))
