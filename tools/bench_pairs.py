"""Micro-benchmark on the real cfg3 geometry: dense-table 2-D tile kernel vs the pair-list kernel, variants interleaved
in one process (launches back to back on one stream, HIP events around 30 launches)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uresnet_pytorch_amd import lib as L_, sparse_ops as so
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
L = L_.load(); dev = torch.device('cuda:0')
blob = make_sparse_blob([0], 512, 50000)
geo = so.SparseGeometry(torch.from_numpy(blob['data'][:, :4].astype(np.int32)).to(dev), 512, 5)
print('n', geo.n, 'rules', geo.rules)


def run(kind, level, cin, cout, pairs, reps=30, xf=False, frag=False):
    if kind == 'nbr':
        tbl, K, n_out, n_in, pl = geo.nbr[level], 27, geo.n[level], geo.n[level], geo.pairs['nbr'][level]
    elif kind == 'chd':
        tbl, K, n_out, n_in, pl = geo.chd[level], 8, geo.n[level + 1], geo.n[level], geo.pairs['chd'][level]
    elif kind == 'up':
        tbl, K, n_out, n_in, pl = geo.up[level], 8, geo.n[level], geo.n[level + 1], geo.pairs['up'][level]
    else:
        tbl, K, n_out, n_in, pl = geo.nbr[level][13:14], 1, geo.n[level], geo.n[level], so.IDENT_PAIRS
    x = torch.randn(n_in, cin, device=dev); wt = torch.randn(K, cout, cin, device=dev) * 0.05; y = torch.empty(n_out, cout, device=dev)
    sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev) * 0.1
    a = L_.GConvArgs()
    a.x = x.data_ptr(); a.wt = wt.data_ptr(); a.tbl = tbl.data_ptr(); a.ld = geo.ld; a.K = K; a.flip = 0; a.n_out = n_out
    a.cin = cin; a.cout = cout; a.y = y.data_ptr()
    if xf:
        a.xf_scale = sc.data_ptr(); a.xf_shift = sh.data_ptr()
    if pairs:
        a.pairs = None if pl[0] is None else pl[0].data_ptr(); a.pairs_tile = pl[1]
    if frag:
        wf = torch.empty_like(wt)
        L_.check(L.urn_weight_fragments(wt.data_ptr(), K, cout, cin, wf.data_ptr(), L_.stream()))
        a.wt_frag = wf.data_ptr()
    st = L_.stream()

    def call():
        L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a), None, st))
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


shapes = [('nbr', 0, 16, 16), ('nbr', 1, 32, 32), ('nbr', 2, 48, 48), ('nbr', 3, 64, 64), ('nbr', 4, 80, 80), ('nbr', 3, 128, 64),
          ('nbr', 2, 96, 48), ('nbr', 1, 64, 32), ('nbr', 0, 32, 16), ('chd', 0, 16, 32), ('chd', 2, 48, 64), ('up', 0, 32, 16),
          ('up', 2, 64, 48), ('nin', 0, 32, 16), ('nin', 3, 128, 64)]
mode = sys.argv[1] if len(sys.argv) > 1 else 'auto'
for kind, lv, ci, co in shapes:
    out = []
    L.urn_set_option(b'pairs_nc', 0); L.urn_set_option(b'pairs_split', 0); L.urn_set_option(b'pairs_max_cin', 999); L.urn_set_option(b'pairs_max_cout', 999); L.urn_set_option(b'pairs_nin', 1)
    t_tile = min(run(kind, lv, ci, co, False) for _ in range(3))
    out.append('tile %.0f' % t_tile)
    variants = [('auto', 0, 0)]
    if mode == 'abl':
        variants = [('dbg%d' % d, -d, 0) for d in (0, 32, 256, 512, 768)]
    if mode == 'cbg':
        variants = [('cbg%d/nc%d/G%d' % (cb, nc, G), nc, G, cb) for cb in (0, 1, 2) for nc in (1, 2) for G in (2, 8)]
    if mode == 'deep':
        variants = [('one set', 0, 0, 0, 0), ('three sets', 0, 0, 0, 14)]
    if mode == 'v3':
        variants = [('loop', 0, 0, 0, 2, 0), ('strip', 0, 0, 0, 2, 0x17E), ('strip nc2', 2, 0, 0, 2, 0x17E)]
    if mode == 'sweep':
        variants += [('nc%d/G%d' % (nc, G), nc, G) for nc in (1, 2) for G in (2, 4, 8)]
    use_frag = mode in ('sweep', 'abl', 'v3')
    for name, nc, G, *rest in variants:
        L.urn_set_option(b'pairs_cbg', rest[0] if rest else 0)
        L.urn_set_option(b'pairs_v3', rest[2] if len(rest) > 2 else 0x17E)
        L.urn_set_option(b'gconv_dbg', -nc if nc < 0 else 0)
        L.urn_set_option(b'pairs_nc', max(nc, 0)); L.urn_set_option(b'pairs_split', G)
        out.append('%s %.0f' % (name, min(run(kind, lv, ci, co, True, frag=use_frag) for _ in range(3))))
    L.urn_set_option(b'pairs_nc', 0); L.urn_set_option(b'pairs_split', 0); L.urn_set_option(b'gconv_dbg', 0); L.urn_set_option(b'pairs_cbg', 0); L.urn_set_option(b'pairs_v3', 0x17E)
    out.append('frag %.0f' % min(run(kind, lv, ci, co, True, frag=True) for _ in range(3)))
    out.append('xf: tile %.0f pairs %.0f' % (min(run(kind, lv, ci, co, False, xf=True) for _ in range(2)),
                                             min(run(kind, lv, ci, co, True, xf=True) for _ in range(2))))
    print('%s L%d %3d->%3d  %s' % (kind, lv, ci, co, ' | '.join(out)), flush=True)
