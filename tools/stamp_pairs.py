"""Phase times inside the pair-list kernel (s_memtime stamps per wave), for the launch forms of a training step:
  python tools/stamp_pairs.py <level> <cin> <cout> [plain|xf|fwd|bwd]
    plain: no fusion;  xf: folded input BatchNorm with given coefficients;
    fwd:   as the forward launches of the executor: coefficients derived from an accumulated-statistics slab in the prologue
           + column statistics of the output accumulated in the epilogue (epilogue 1, 8 slots);
    bwd:   as the input-gradient launches: BatchNorm-backward reduce with ReLU mask in the epilogue (epilogue 2, 8 slots)
Needs a library with the stamps compiled in (tools/build_diag_lib.sh -> uresnet_pytorch_amd/liburesnet_hip_diag.so);
URN_LIB_PATH selects it."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uresnet_pytorch_amd import lib as L_
if os.environ.get('URN_LIB_PATH'):
    L_.LIB_PATH = os.environ['URN_LIB_PATH']
from uresnet_pytorch_amd import sparse_ops as so
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
L = L_.load(); dev = torch.device('cuda:0')
lv, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else 'plain'
blob = make_sparse_blob([0], 512, 50000)
geo = so.SparseGeometry(torch.from_numpy(blob['data'][:, :4].astype(np.int32)).to(dev), 512, 5)
n = geo.n[lv]; pl = geo.pairs['nbr'][lv]
x = torch.randn(n, cin, device=dev); wt = torch.randn(27, cout, cin, device=dev) * 0.05; y = torch.empty(n, cout, device=dev)
wf = torch.empty_like(wt)
L_.check(L.urn_weight_fragments(wt.data_ptr(), 27, cout, cin, wf.data_ptr(), L_.stream()))
sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev) * 0.1
keep = []
a = L_.GConvArgs()
a.x = x.data_ptr(); a.wt = wt.data_ptr(); a.wt_frag = wf.data_ptr(); a.tbl = geo.nbr[lv].data_ptr(); a.ld = geo.ld; a.K = 27; a.n_out = n
a.cin = cin; a.cout = cout; a.y = y.data_ptr(); a.pairs = pl[0].data_ptr(); a.pairs_tile = pl[1]
if mode == 'xf':
    a.xf_scale = sc.data_ptr(); a.xf_shift = sh.data_ptr()
if mode == 'fwd':
    SLOTS = 8
    xd = x.double()
    sums = torch.zeros(SLOTS, 2, cin, dtype=torch.float64, device=dev)
    sums[0, 0] = xd.sum(0); sums[0, 1] = (xd * xd).sum(0)
    gam = torch.rand(cin, device=dev) + 0.5; bet = torch.randn(cin, device=dev) * 0.1
    outs = [torch.empty(cin, device=dev) for _ in range(4)]
    part = torch.zeros(SLOTS, 2, cout, dtype=torch.float64, device=dev)
    res = torch.randn(n, cout, device=dev)
    keep += [sums, gam, bet, outs, part, res]
    a.xs_slots = SLOTS; a.xs_n = n; a.xs_sums[0] = sums.data_ptr(); a.xs_ld[0] = cin; a.xs_split = cin
    a.xs_gamma = gam.data_ptr(); a.xs_beta = bet.data_ptr()
    a.xs_mean, a.xs_invstd, a.xs_scale, a.xs_shift = [o.data_ptr() for o in outs]
    a.fin_eps = 1e-4; a.fin_momentum = 0.9
    a.epilogue = 1; a.part = part.data_ptr(); a.part_slots = SLOTS; a.res = res.data_ptr()
if mode == 'bwd':
    SLOTS = 8
    part = torch.zeros(SLOTS, 2, cout, dtype=torch.float64, device=dev)
    ex = torch.randn(n, cout, device=dev)
    esc = torch.rand(cout, device=dev) + 0.5; esh = torch.randn(cout, device=dev) * 0.3
    emu = torch.randn(cout, device=dev); eis = torch.rand(cout, device=dev) + 0.5
    keep += [part, ex, esc, esh, emu, eis]
    a.flip = 1
    a.epilogue = 2; a.part = part.data_ptr(); a.part_slots = SLOTS
    a.e_x = ex.data_ptr(); a.e_scale = esc.data_ptr(); a.e_shift = esh.data_ptr(); a.e_mean = emu.data_ptr(); a.e_invstd = eis.data_ptr()
npart = ctypes.c_int(0)
def call():
    L_.check(L.urn_gconv_fwd_ex(ctypes.byref(a), ctypes.byref(npart), L_.stream()))
for _ in range(5): call()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): call()
e1.record(); torch.cuda.synchronize()
print('%s L%d %d->%d: %.1f us per launch (30 back to back)' % (mode, lv, cin, cout, e0.elapsed_time(e1) / 30 * 1e3))
st = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
if L.urn_set_option(b'gconv_stamp_ptr', st.data_ptr()) == 0:
    call()
    torch.cuda.synchronize()
    L.urn_set_option(b'gconv_stamp_ptr', 0)
    s = st.cpu().numpy().reshape(-1, 8)
    s = s[(s[:, 0] != 0) & (s[:, 4] != 0)]
    if len(s):
        t0 = s[:, 0].min()
        clk = 2.1e9   # __builtin_readcyclecounter = s_memtime: shader cycles (~2.1 GHz under this load)
        us = lambda v: v / clk * 1e6
        end = np.where(s[:, 6] != 0, s[:, 6], s[:, 4])
        print('  %d waves; entry spread %.2f us; kernel span %.2f us' % (len(s), us(s[:, 0].max() - t0), us(end.max() - t0)))
        segs = [('  header wait', 0, 7), ('  rest of it', 7, 1), ('prologue', 0, 1), ('strip fill', 1, 5), ('block loop', 5, 2), ('barrier wait', 2, 3), ('epilogue rows', 3, 4), ('statistics tail', 4, 6)]
        for nm, i, j in segs:
            ok = (s[:, i] != 0) & (s[:, j] != 0)
            if not ok.any():
                continue
            d = us(s[ok, j] - s[ok, i])
            print('    %-15s mean %.2f  p10 %.2f  p90 %.2f  max %.2f us' % (nm, d.mean(), np.percentile(d, 10), np.percentile(d, 90), d.max()))
