"""CPU-side checks: the C-ABI library loads and exports every symbol include/uresnet_hip.h declares,
host logic (event sharding, flat gradients, flags, io blobs, checkpoints)."""
import ctypes
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from uresnet_pytorch_amd import lib
    L = lib.load()
    hdr = open(os.path.join(ROOT, 'include', 'uresnet_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)          # prose in comments is not a declaration
    declared = set(re.findall(r'\b(urn_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations found'
    for name in declared:
        assert hasattr(L, name), 'symbol %s declared in the header but not exported' % name
    assert declared == set(lib.SIGNATURES.keys()), declared ^ set(lib.SIGNATURES.keys())
    assert L.urn_version() >= 100
    # pure-host helpers can be called without a GPU
    assert L.urn_hash_capacity(50000) == 131072 and L.urn_hash_bytes(1024) == 16384
    assert L.urn_unique_scratch_bytes(1000) > 4000 and L.urn_bn_scratch_bytes(16) > 0


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md is the reference-side binding text: small, and it names every function the header declares."""
    path = os.path.join(ROOT, 'INTEGRATION.md')
    assert os.path.getsize(path) < 50 * 1024, 'INTEGRATION.md is %d bytes' % os.path.getsize(path)
    doc = open(path).read()
    hdr = open(os.path.join(ROOT, 'include', 'uresnet_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(urn_[a-z0-9_]+)\s*\(', hdr))
    named = set(re.findall(r'`(urn_[a-z0-9_]+)`', doc))
    assert declared <= named, sorted(declared - named)
    assert 'import uresnet_pytorch_amd.scn as scn' in doc and 'ctypes.CDLL' in doc


def test_product_path_has_no_cpu_fallback():
    from uresnet_pytorch_amd.models import SparseUResNet
    fl = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=2, SPATIAL_SIZE=16, NUM_CLASS=3)
    net = SparseUResNet(fl)
    pc = torch.tensor([[1., 2., 3., 0., 0.5], [1., 2., 4., 0., 0.7]])
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        net(pc)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'uresnet_pytorch_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


def test_shard_events_balanced_and_deterministic():
    from uresnet_pytorch_amd.parallel import shard_events
    sizes = [50000, 10000, 42000, 38000, 7000, 51000, 20000, 30000]
    out = shard_events(sizes, 4)
    assert sorted(sum(out, [])) == list(range(8))
    loads = [sum(sizes[i] for i in r) for r in out]
    assert max(loads) - min(loads) <= max(sizes) * 0.5
    assert out == shard_events(sizes, 4)
    assert shard_events([5, 4, 3], 1) == [[0, 1, 2]]
    assert shard_events([], 2) == [[], []]


def test_flat_gradients_views():
    from uresnet_pytorch_amd.parallel import FlatGradients
    m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    fg = FlatGradients(m)
    assert fg.flat.numel() == sum(p.numel() for p in m.parameters())
    m(torch.randn(5, 4)).sum().backward()
    assert fg.flat.abs().sum() > 0
    assert all(p.grad.data_ptr() >= fg.flat.data_ptr() for p in m.parameters())   # still views after backward
    g0 = fg.flat.clone()
    m(torch.randn(5, 4)).sum().backward()      # accumulates in place
    assert not torch.equal(g0, fg.flat)
    fg.zero()
    assert fg.flat.abs().sum() == 0 and all(p.grad.abs().sum() == 0 for p in m.parameters())


def test_flags_names_and_batch_rules(monkeypatch):
    from uresnet_pytorch_amd.flags import URESNET_FLAGS
    fl = URESNET_FLAGS().parse_args(['train', '-mn', 'uresnet_sparse', '-io', 'synthetic_sparse', '-ss', '512',
                                     '-uf', '16', '-uns', '5', '-nc', '5', '-bs', '16', '--gpus', '0,1',
                                     '-dkeys', 'data,label', '-sd', '7', '-lr', '0.01'], run=False)
    assert fl.MODEL_NAME == 'uresnet_sparse' and fl.SPATIAL_SIZE == 512 and fl.URESNET_FILTERS == 16
    assert fl.URESNET_NUM_STRIDES == 5 and fl.NUM_CLASS == 5 and fl.GPUS == [0, 1]
    assert fl.BATCH_SIZE == 16 and fl.MINIBATCH_SIZE == 8 and fl.SEED == 7 and fl.LEARNING_RATE == 0.01
    assert fl.DATA_KEYS == ['data', 'label'] and fl.DATA_DIM == 3
    with pytest.raises(ValueError):
        URESNET_FLAGS().parse_args(['train', '-mn', 'x'], run=False)               # both batch sizes negative
    with pytest.raises(ValueError):
        URESNET_FLAGS().parse_args(['train', '-bs', '3', '-mbs', '2'], run=False)   # not a multiple


def test_sparse_blob_layout():
    from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
    b = make_sparse_blob([3, 4], 64, 500, compute_weight=True)
    d = b['data']
    assert d.shape == (1000, 5) and d.dtype == np.float32
    assert set(np.unique(d[:, 3])) == {0.0, 1.0} and (d[:, 4] > 0).all()
    assert d[:, :3].min() >= 0 and d[:, :3].max() <= 63
    assert len(np.unique(d[:500, :3], axis=0)) == 500          # unique voxels per event
    assert b['label'].shape == (1000, 1) and b['weight'].shape == (1000, 1)
    b2 = make_sparse_blob([3, 4], 64, 500)
    assert np.array_equal(b2['data'], d)                        # deterministic


def test_trainer_checkpoint_roundtrip_dense_cpu(tmp_path):
    """reference trainval.py:32-40,171-196: {'global_step','state_dict','optimizer'}, resume at step+1."""
    from uresnet_pytorch_amd.trainval import trainval
    from uresnet_pytorch_amd.iotools.synthetic import make_dense_blob

    def flags(path=''):
        return SimpleNamespace(MODEL_NAME='uresnet_dense', DATA_DIM=2, URESNET_FILTERS=4, URESNET_NUM_STRIDES=2,
                               SPATIAL_SIZE=16, NUM_CLASS=3, BN_MOMENTUM=0.9, TRAIN=True, GPUS=[],
                               LEARNING_RATE=1e-3, MODEL_PATH=path, WEIGHT_PREFIX=str(tmp_path / 'snap'))
    b = make_dense_blob([0, 1], 16, 2, 3)
    blob = {'data': [[b['data'][0], b['data'][1]]], 'label': [[b['label'][0], b['label'][1]]]}
    torch.manual_seed(0)
    t = trainval(flags())
    assert t.initialize() == 0
    res = t.train_step(blob, epoch=0., batch_size=2)
    assert set(res.keys()) == {'segmentation', 'softmax', 'accuracy', 'loss_seg'}
    assert res['segmentation'][0].shape == (3, 16, 16) and abs(res['softmax'][0].sum(0) - 1).max() < 1e-5
    assert t.tspent['train'] > 0 and t.tspent_sum['forward'] > 0
    # inference metrics CSV written from the same result dict (main_funcs.log_metrics)
    from uresnet_pytorch_amd import main_funcs
    h = SimpleNamespace(iteration=3, metrics_logger=None)
    fl = flags(); fl.LOG_DIR = str(tmp_path)
    main_funcs.log_metrics(h, fl, blob, res)
    h.metrics_logger.close()
    rows = open(str(tmp_path / 'inference_metrics-0000003.csv')).read().strip().split('\n')
    assert rows[0].startswith('iter,id,acc,correct_softmax,nonzero_pixels,class_acc_0') and len(rows) == 3
    t.save_state(4)
    t2 = trainval(flags(str(tmp_path / 'snap-4.ckpt')))
    assert t2.initialize() == 5
    for (k1, v1), (k2, v2) in zip(t._net.state_dict().items(), t2._net.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_inference_metrics_against_brute_force():
    """utils.compute_metrics_sparse / compute_metrics_dense against per-voxel loops."""
    from uresnet_pytorch_amd import utils
    rng = np.random.default_rng(0)
    nc, S = 5, 48
    n = 400
    coords = rng.integers(0, S, size=(n, 3)).astype(np.float64)
    batch = np.repeat([0., 1.], n // 2)[:, None]
    energy = rng.uniform(0.1, 3.0, size=(n, 1))
    data = np.concatenate([coords, batch, energy], 1)
    label = rng.integers(0, nc, size=(n, 1)).astype(np.float64)
    soft = rng.dirichlet(np.ones(nc), size=n)
    res, _ = utils.compute_metrics_sparse([data], [label], [soft], None, N=S)
    assert len(res['acc']) == 2 and res['id'] == [0.0, 1.0]
    for e in range(2):
        sel = np.where(batch[:, 0] == e)[0]
        conf = np.zeros((nc, nc), np.int32); econf = np.zeros((nc, nc), np.float64); hit = 0; ll = 0.0
        for i in sel:
            p = int(np.argmax(soft[i])); t = int(label[i, 0])
            conf[t, p] += 1; econf[t, p] += energy[i, 0]; hit += p == t; ll -= np.log(soft[i, t])
        assert abs(res['acc'][e] - hit / len(sel)) < 1e-12 and abs(res['loss_seg'][e] - ll / len(sel)) < 1e-9
        assert np.array_equal(res['confusion_matrix'][e], conf) and np.allclose(res['energy_confusion_matrix'][e], econf, rtol=1e-5)
        assert res['nonzero_pixels'][e] == len(sel) and res['misclassified_pixels'][e].shape == (len(sel) - hit, 3 + 5)
        assert np.allclose(res['class_acc'][e], np.diag(conf) / conf.sum(1)) and res['distances'][e].sum() <= len(sel)
        assert np.array_equal(res['class_pixel'][e], conf.sum(1))
    # dense: 2-D image, background = last class at empty pixels
    H = 24
    img = np.zeros((1, H, H)); lab = np.full((1, H, H), nc - 1.0)
    m = rng.uniform(size=(H, H)) < 0.3
    img[0][m] = rng.uniform(0.1, 1.0, size=int(m.sum())); lab[0][m] = rng.integers(0, nc - 1, size=int(m.sum()))
    sm = rng.dirichlet(np.ones(nc), size=H * H).T.reshape(nc, H, H)
    r = utils.compute_metrics_dense([img], [lab], [sm], None)
    pred = sm.argmax(0)
    assert abs(r['acc'][0] - (pred[m] == lab[0][m]).mean()) < 1e-12 and r['nonzero_pixels'][0] == int(m.sum())
    conf = np.zeros((nc - 1, nc - 1), np.int32)
    for y, x in zip(*np.where(m)):
        if pred[y, x] < nc - 1:
            conf[int(lab[0, y, x]), pred[y, x]] += 1
    assert np.array_equal(r['confusion_matrix'][0], conf) and r['confusion_matrix'][0].shape == (nc - 1, nc - 1)


def test_deferred_float_behaves_like_a_float():
    """utils.DeferredFloat: the accuracy the GPU loss returns -- a device scalar that becomes a python float on first use."""
    from uresnet_pytorch_amd.utils import DeferredFloat
    d = DeferredFloat(torch.tensor(0.75))
    assert abs(d - 0.75) < 1e-7 and d / 3 == 0.25 and 1 + d == 1.75 and d * 2 == 1.5 and 2 - d == 1.25
    assert float(np.array([d, DeferredFloat(torch.tensor(0.25))]).sum()) == 1.0        # trainval sums per-entry accuracies
    assert '%.2f' % d == '0.75' and '{:f}'.format(d) == '0.750000' and 0.0 <= d <= 1.0 and d > 0.5 and bool(d)
    assert d._t is None                                                                 # the device tensor is released after use


def _cli_flags(argv):
    from uresnet_pytorch_amd.flags import URESNET_FLAGS
    return URESNET_FLAGS().parse_args(argv, run=False)


def test_checkpoint_module_prefix_flag_roundtrip(tmp_path):
    """-cmp / --ckpt_module_prefix (reference trainval.py:37,152-154,181): the reference saves the state of a
    DataParallel-wrapped module, every key 'module.<name>', and restores with strict=False -- a file without the prefix
    would restore NOTHING there, silently.  With the flag every key carries the prefix; the file loads into a
    'module.'-wrapped model (strict=False, no missing and no unexpected keys) and back through initialize()."""
    from uresnet_pytorch_amd.trainval import trainval
    from uresnet_pytorch_amd.iotools.synthetic import make_dense_blob
    base = ['train', '-mn', 'uresnet_dense', '-io', 'synthetic_dense', '-dd', '2', '-uf', '4', '-uns', '2', '-ss', '16', '-nc', '3',
            '-bs', '2', '-mbs', '2', '-wp', str(tmp_path / 'snap'), '-sd', '1']
    with pytest.raises(SystemExit):
        _cli_flags(base + ['-ls', 'x'])
    with pytest.raises(ValueError):
        _cli_flags(base + ['-ls', '3'])                   # not a power of two
    assert _cli_flags(base).CKPT_MODULE_PREFIX is False
    flags = _cli_flags(base + ['-cmp', '-ls', '4096'])
    assert flags.CKPT_MODULE_PREFIX is True and flags.LOSS_SCALE == 4096.0
    flags.GPUS = []                                        # CPU
    t = trainval(flags)
    t.initialize()
    b = make_dense_blob([0, 1], 16, 2, 3)
    t.train_step({'data': [[b['data'][0], b['data'][1]]], 'label': [[b['label'][0], b['label'][1]]]}, epoch=0., batch_size=2)
    t.save_state(9)
    ck = torch.load(str(tmp_path / 'snap-9.ckpt'), weights_only=True)
    assert set(ck.keys()) == {'global_step', 'state_dict', 'optimizer'} and ck['global_step'] == 9
    assert all(k.startswith('module.') for k in ck['state_dict']) and len(ck['state_dict']) == len(t._net.state_dict())

    class Wrapped(torch.nn.Module):                        # what the reference's GraphDataParallel(net) looks like to load_state_dict
        def __init__(self, m):
            super().__init__(); self.module = m
    from uresnet_pytorch_amd.models import DenseUResNet
    w = Wrapped(DenseUResNet(flags))
    missing, unexpected = w.load_state_dict(ck['state_dict'], strict=False)
    assert not missing and not unexpected
    for (k1, v1), (k2, v2) in zip(t._net.state_dict().items(), w.module.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    flags2 = _cli_flags(base + ['-mp', str(tmp_path / 'snap-9.ckpt')]); flags2.GPUS = []
    t2 = trainval(flags2)
    assert t2.initialize() == 10
    for (k1, v1), (k2, v2) in zip(t._net.state_dict().items(), t2._net.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_npz_readers_serve_the_blob_contract(tmp_path):
    """-io npz_sparse / npz_dense behind io_factory (reference iotools.py:5-10, iotools_sparse.py:120-160,
    311-318, iotools_dense.py:195-198): per-GPU (N, d+2) rows with the step-wide batch id, (N, 1) labels, class-balancing
    weights with -cw, wrap-around at the end of the file, store_segment -> output file; the trainer steps on the blob."""
    from uresnet_pytorch_amd.iotools import io_factory
    from uresnet_pytorch_amd.iotools import array_io
    from uresnet_pytorch_amd.iotools.synthetic import generate_event, make_dense_blob
    events = []
    for seed in range(5):
        c, v, l = generate_event(seed, 32, 100 + 10 * seed)
        events.append({'voxels': c, 'feature': v, 'label': l})
    path = str(tmp_path / 'ev.npz')
    array_io.write_sparse_npz(path, events)
    argv = ['train', '-mn', 'uresnet_sparse', '-io', 'npz_sparse', '-if', path, '-dkeys', 'data,label', '-cw', '-ss', '32', '-nc', '5',
            '-bs', '4', '-mbs', '2', '--gpus', '0,1', '-sh', '0', '-of', str(tmp_path / 'out.npz')]
    flags = _cli_flags(argv)
    assert flags.DATA_KEYS == ['data', 'label', '_weights_']
    io = io_factory(flags)
    io.initialize()
    assert io.num_entries() == 5 and io.num_channels() == 1 and io.batch_per_step() == 4
    idx, blob = io.next()
    assert [i.tolist() for i in idx] == [[0, 1], [2, 3]]
    for g in range(2):
        d, lab, w = blob['data'][g], blob['label'][g], blob['_weights_'][g]
        n = sum(len(events[e]['voxels']) for e in idx[g])
        assert d.shape == (n, 5) and d.dtype == np.float32 and lab.shape == (n, 1) and w.shape == (n, 1)
        assert sorted(np.unique(d[:, 3]).tolist()) == [2.0 * g, 2.0 * g + 1]      # batch id = position in the step
        e0 = events[idx[g][0]]
        k = len(e0['voxels'])
        assert np.array_equal(d[:k, :3], e0['voxels'].astype(np.float32)) and np.array_equal(d[:k, 4], e0['feature'])
        assert np.array_equal(lab[:k, 0], e0['label'].astype(np.float32))
        assert np.allclose(w[:k], array_io.class_weights(lab[:k]))
        cls, cnt = np.unique(lab[:k], return_counts=True)
        if cls.tolist() == list(range(len(cls))):                                  # labels 0..k-1 present: plain class balancing
            assert np.allclose(w[:k][lab[:k] == cls[0]], k / (len(cls) * cnt[0]))
    idx2, _ = io.next()
    assert [i.tolist() for i in idx2] == [[4, 0], [1, 2]]                          # wraps around
    soft = [np.random.default_rng(0).dirichlet(np.ones(5), size=len(blob['data'][g])).astype(np.float32) for g in range(2)]
    io.store_segment(idx, blob['data'], soft)
    io.finalize()
    out = np.load(str(tmp_path / 'out.npz'))
    assert sorted(out.files) == sorted(['prediction/%d' % i for i in range(4)] + ['softmax/%d' % i for i in range(4)])
    assert out['prediction/1'].shape == (len(events[1]['voxels']),)
    # dense reader + one trainer step on its blob (CPU)
    b = make_dense_blob([0, 1, 2], 16, 2, 3)
    dpath = str(tmp_path / 'dense.npz')
    array_io.write_dense_npz(dpath, {'data': b['data'], 'label': b['label']})
    fl = _cli_flags(['train', '-mn', 'uresnet_dense', '-io', 'npz_dense', '-if', dpath, '-dkeys', 'data,label', '-dd', '2', '-uf', '4',
                     '-uns', '2', '-ss', '16', '-nc', '3', '-bs', '2', '-mbs', '2', '-sh', '0', '-it', '2', '-rs', '1'])
    fl.GPUS = []
    from uresnet_pytorch_amd import main_funcs
    fl.TRAIN = True
    h = main_funcs.prepare(fl)
    assert h.data_io.num_entries() == 3 and h.data_io.num_channels() == 1
    main_funcs.train_loop(fl, h)
    assert h.iteration == 2
    with pytest.raises(NotImplementedError):
        io_factory(SimpleNamespace(IO_TYPE='larcv_sparse', GPUS=[], BATCH_SIZE=1, MINIBATCH_SIZE=1))
