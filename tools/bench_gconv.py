"""Micro-benchmark of the gather-conv kernel on the real cfg3 geometry, variants interleaved in one process."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uresnet_pytorch_amd import lib as L_, sparse_ops as so
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
L = L_.load(); dev = torch.device('cuda:0')
blob = make_sparse_blob([0], 512, 50000)
geo = so.SparseGeometry(torch.from_numpy(blob['data'][:, :4].astype(np.int32)).to(dev), 512, 5)
print('n', geo.n, 'rules', geo.rules)
def run(level, cin, cout, reps=30):
    n = geo.n[level]
    x = torch.randn(n, cin, device=dev); wt = torch.randn(27, cout, cin, device=dev) * 0.05; y = torch.empty(n, cout, device=dev)
    def call():
        L_.check(L.urn_gconv_fwd(x.data_ptr(), wt.data_ptr(), geo.nbr[level].data_ptr(), geo.ld, 27, 0, n, cin, cout, None, y.data_ptr(), L_.stream()))
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def run_dw(level, cin, cout, reps=30):
    n = geo.n[level]
    x = torch.randn(n, cin, device=dev); dy = torch.randn(n, cout, device=dev); dw = torch.zeros(27, cin, cout, device=dev)
    def call():
        L_.check(L.urn_gconv_bwd_dw(x.data_ptr(), dy.data_ptr(), geo.nbr[level].data_ptr(), geo.ld, 27, n, cin, cout, dw.data_ptr(), L_.stream()))
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
shapes = [(0, 16, 16), (1, 32, 32), (2, 48, 48), (3, 64, 64), (4, 80, 80), (3, 128, 64), (2, 96, 48), (1, 64, 32), (0, 32, 16), (1, 16, 32), (2, 32, 48), (3, 48, 64), (4, 64, 80)]
variants = [('auto', (0, 0, 0))]
if len(sys.argv) > 1 and sys.argv[1] == 'il':
    variants = [('plain', (0, 0, 0, 0, 0)), ('interleaved', (0, 0, 0, 0, 1))]
    shapes = [(0, 16, 16), (1, 32, 32), (0, 32, 16), (0, 16, 32), (1, 16, 32), (1, 32, 64), (2, 32, 48)] + shapes[2:5]
if len(sys.argv) > 1 and sys.argv[1] == 'narrow':
    shapes = [(0, 16, 16), (0, 32, 16), (0, 16, 32), (1, 32, 32), (1, 16, 32), (1, 64, 32), (1, 32, 64)]
    variants = [('auto', (0, 0, 0))] + [('%dx%d' % (r, c), (r, c, 0)) for r in (1, 2, 4) for c in (1, 2, 4)]
if len(sys.argv) > 1 and sys.argv[1] == 'xcd':
    variants = [('xcd-aware', (0, 0, 0, 0, 1, 0)), ('round-robin', (0, 0, 0, 0, 1, 64))]
if len(sys.argv) > 1 and sys.argv[1] == 'depth':
    variants = [('d%d' % d, (0, 0, 0, d)) for d in (1, 2, 4)]
if len(sys.argv) > 1 and sys.argv[1] == 'sweep':
    shapes = [(1, 32, 32), (2, 48, 48), (3, 64, 64), (4, 80, 80), (3, 128, 64), (2, 96, 48), (1, 64, 32), (3, 64, 128), (2, 48, 96), (1, 32, 64), (4, 160, 80), (4, 80, 160)]
    variants = [('auto', (0, 0, 0))] + [('%dx%d/k%d' % (r, c, k), (r, c, k)) for k in (2, 3, 4, 5) for r in (1, 2, 4) for c in (1, 2, 3, 4, 5)]
for lv, ci, co in shapes:
    out = []
    for name, mw in variants:
        L.urn_set_option(b'gconv_kernel', 6); L.urn_set_option(b'tile_rb', mw[0]); L.urn_set_option(b'tile_cb', mw[1]); L.urn_set_option(b'tile_kc', mw[2])
        L.urn_set_option(b'tile_depth', mw[3] if len(mw) > 3 else 0)
        L.urn_set_option(b'tile_il', mw[4] if len(mw) > 4 else 1)
        L.urn_set_option(b'gconv_dbg', mw[5] if len(mw) > 5 else 0)
        if (mw[1] and (co // 16) % mw[1]) or (mw[2] and ((ci // 16) % mw[2] or mw[0] * mw[1] > 12 or (mw[0] + mw[1]) * 128 * (mw[2] * 16 + 4) > 98304)):
            continue
        t = min(run(lv, ci, co) for _ in range(3))
        out.append('%s %.0f' % (name, t))
    fl = 2.0 * geo.rules[lv] * ci * co
    tdw = min(run_dw(lv, ci, co) for _ in range(3))
    if len(sys.argv) > 1 and sys.argv[1] == 'dw':
        outs = []
        for nb in (512, 1024, 2048, 4096, 8192):
            L.urn_set_option(b'dw_blocks', nb)
            outs.append('%d: %.0f' % (nb, min(run_dw(lv, ci, co) for _ in range(3))))
        L.urn_set_option(b'dw_blocks', 2048)
        print('L%d %3d->%3d dW us by target blocks  %s' % (lv, ci, co, ' | '.join(outs)))
        continue
    L.urn_set_option(b'tile_rb', 0); L.urn_set_option(b'tile_cb', 0); L.urn_set_option(b'tile_kc', 0); L.urn_set_option(b'tile_il', 1); L.urn_set_option(b'gconv_dbg', 0)
    print('L%d %3d->%3d n=%6d fwd %s us | dW %.0f us (%.1f TF)' % (lv, ci, co, geo.n[lv], ' | '.join(out), tdw, fl / tdw / 1e6))
