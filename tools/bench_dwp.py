"""Weight gradient on the real cfg3 geometry: dense-table kernel with atomics (k_gconv_dw2) vs the two-stage pair-list kernel."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uresnet_pytorch_amd import lib as L_, sparse_ops as so
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
L = L_.load(); dev = torch.device('cuda:0')
blob = make_sparse_blob([0], 512, 50000)
geo = so.SparseGeometry(torch.from_numpy(blob['data'][:, :4].astype(np.int32)).to(dev), 512, 5)
def timeit(call, reps=20):
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def run(kind, level, cin, cout, variants):
    if kind == 'nbr': tbl, K, n_out, n_in, pl = geo.nbr[level], 27, geo.n[level], geo.n[level], geo.pairs['nbr'][level]
    elif kind == 'chd': tbl, K, n_out, n_in, pl = geo.chd[level], 8, geo.n[level + 1], geo.n[level], geo.pairs['chd'][level]
    else: tbl, K, n_out, n_in, pl = geo.up[level], 8, geo.n[level], geo.n[level + 1], geo.pairs['up'][level]
    x = torch.randn(n_in, cin, device=dev); dy = torch.randn(n_out, cout, device=dev); dw = torch.zeros(K, cin, cout, device=dev)
    st = L_.stream()
    out = ['old %.0f' % min(timeit(lambda: L_.check(L.urn_gconv_bwd_dw(x.data_ptr(), dy.data_ptr(), tbl.data_ptr(), geo.ld, K, n_out, cin, cout, dw.data_ptr(), st))) for _ in range(2))]
    for smax, waves, dbg in variants:
        L.urn_set_option(b'dwp_smax', smax); L.urn_set_option(b'dwp_waves', waves); L.urn_set_option(b'dwp_cap', dbg)
        sb = L.urn_gconv_dw_pairs_scratch_bytes(n_out, pl[1], K, cin, cout)
        scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
        t = min(timeit(lambda: L_.check(L.urn_gconv_bwd_dw_pairs(x.data_ptr(), 0, None, None, dy.data_ptr(), 0, pl[0].data_ptr(), pl[1], K, n_out, cin, cout, dw.data_ptr(), scratch.data_ptr(), sb, st))) for _ in range(2))
        out.append('S<=%d/w%d/cap%d %.0f' % (smax, waves, dbg, t))
    print('%s L%d %3d->%3d  %s' % (kind, level, cin, cout, ' | '.join(out)), flush=True)
variants = [(16, 4096, 1), (16, 4096, 2), (16, 4096, 4), (64, 4096, 1), (64, 4096, 2), (64, 8192, 2), (128, 8192, 2)]
for sh in [('nbr', 0, 16, 16), ('nbr', 1, 32, 32), ('nbr', 2, 48, 48), ('nbr', 3, 64, 64), ('nbr', 4, 80, 80), ('nbr', 3, 128, 64), ('nbr', 1, 64, 32),
           ('chd', 0, 16, 32), ('up', 0, 32, 16), ('chd', 2, 48, 64)]:
    run(*sh, variants)
