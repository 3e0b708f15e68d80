"""A/B of urn_set_option settings on the cfg3 training step IN ONE PROCESS (box-to-box differences of a few percent hide
1-2 % effects): python tools/ab_options.py "pairs_waves=4096" "pairs_waves=2560" "pairs_waves=2560,dw_blocks=640" ...
Each argument is one policy (comma-separated key=value; the first is the baseline whose keys are restored between
policies).  Rounds of 20 steps per policy, interleaved, 5 rounds; prints min / median per policy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np, torch
from uresnet_pytorch_amd import lib as L_, parallel
from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss
if os.environ.get('URN_LIB_PATH'): L_.LIB_PATH = os.environ['URN_LIB_PATH']   # an alternative build (compile-time choices)
L = L_.load(); dev = torch.device('cuda:0')
DEFAULTS = {'pairs_waves': 2560, 'pairs_wgs': 512, 'dw_blocks': 768, 'pairs_max_cin': 80, 'pairs_max_cout': 999,
            'pairs_nc': 0, 'pairs_split': 0, 'pairs_split_kc1': 0, 'pairs_split_kc2': 0, 'pairs_split_kc3': 0, 'pairs_split_kc4': 0, 'pairs_split_kc5': 0, 'pairs_cbg': 0, 'dw_split': 2, 'dw_2stage': 0, 'pairs_prec': 1, 'pairs_wgs16': 256, 'net_dbg_skip_dw': 0, 'pairs_waves_fwd': 0, 'pairs_v3': 0x17E, 'pairs_lds_cap16': 0, 'dw_pairs': 0, 'dw_rowmode': 1, 'dwp_cap': 2, 'dwp_waves': 2048, 'dwp_smax': 16}
# CFG=5: BASELINE configs[4] shape (768^3, 200k voxels, uf 32, uns 7); PREC=fp32|bf16|fp16 (flags -prec)
CFG5 = os.environ.get('CFG', '3') == '5'
SS, NV, UF, UNS = (768, 200000, 32, 7) if CFG5 else (512, 50000, 16, 5)
flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=UF, URESNET_NUM_STRIDES=UNS, SPATIAL_SIZE=SS, NUM_CLASS=5,
                        PRECISION=os.environ.get('PREC', 'fp32'))
torch.manual_seed(0)
model = SparseUResNet(flags).to(dev).train(); crit = SparseSegmentationLoss(flags)
blob = make_sparse_blob(list(range(int(os.environ.get('EVENTS', '1')))), SS, NV)   # EVENTS=2: two events per GPU
data = torch.from_numpy(blob['data']).to(dev); label = torch.from_numpy(blob['label']).to(dev)
grads = parallel.FlatGradients(model); opt = parallel.FlatAdam(grads, lr=1e-3)
def step():
    grads.zero(); out = model(data); loss, _ = crit(out, [data], [label], None); loss.backward(); grads.all_reduce(); opt.step()
def run(n=6 if CFG5 else 20):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def parse(a):
    d = {}
    for kv in a.split(','):
        if not kv: continue
        k, v = kv.split('=')
        d[k] = v if k.startswith('tiles_') else int(v)     # tiles_nbr=64/64/64/128/64: rows per tile of the pair lists, per level
    return d
policies = [parse(a) for a in sys.argv[1:]] or [{}]
res = [[] for _ in policies]
for rnd in range(5):
    for i, pol in enumerate(policies):
        for k, v in DEFAULTS.items(): L.urn_set_option(k.encode(), v)
        from uresnet_pytorch_amd import sparse_ops as so
        so.SparseGeometry.PAIRS_TILES = {'nbr': [64], 'chd': [64], 'up': [128]}
        for k, v in pol.items():
            if k.startswith('tiles_'):
                so.SparseGeometry.PAIRS_TILES[k[6:]] = [int(t) for t in v.split('/')]
                continue
            assert L.urn_set_option(k.encode(), v) == 0, k
        res[i].append(run())
for a, r in zip(sys.argv[1:], res):
    print('%-44s min %.3f  median %.3f  all %s' % (a, min(r), float(np.median(r)), ' '.join('%.3f' % v for v in r)), flush=True)
