"""Two ranks of the SPARSE data-parallel path on one GPU (gloo; the driver's N > 1 bench runs the same code over RCCL):
events sharded over ranks, the executor writing its gradients into the flat buffer, ONE all-reduce(SUM), flat Adam.
Asserts rank-identical parameters after the step and that the summed gradient equals the single-process sum of the two
events' gradients (reference uresnet/trainval.py:21-30 with GraphDataParallel, uresnet/ops.py:19-60)."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _flags():
    return SimpleNamespace(MODEL_NAME='uresnet_sparse', DATA_DIM=3, URESNET_FILTERS=16, URESNET_NUM_STRIDES=3, SPATIAL_SIZE=64,
                           NUM_CLASS=5, BN_MOMENTUM=0.9, TRAIN=True, GPUS=[0], LEARNING_RATE=1e-3, MODEL_PATH='', WEIGHT_PREFIX='')


def _blob():
    from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
    blobs = [make_sparse_blob([s], 64, 1500 + 300 * s) for s in (0, 1)]
    return {'data': [[b['data'] for b in blobs]], 'label': [[b['label'] for b in blobs]]}


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      URN_DIST_BACKEND='gloo')
    from uresnet_pytorch_amd.trainval import trainval
    torch.manual_seed(100 + rank)                 # different init per rank: initialize() must broadcast rank 0's
    t = trainval(_flags())
    t.initialize()
    p0 = torch.cat([p.detach().flatten() for p in t._net.parameters()]).cpu().clone()
    res = t.train_step(_blob(), epoch=0., batch_size=2)
    g = t._grads.flat.detach().cpu().clone()
    p1 = torch.cat([p.detach().flatten() for p in t._net.parameters()]).cpu().clone()
    q.put((rank, p0.numpy(), g.numpy(), p1.numpy(), float(res['loss_seg']), list(t.last_slots), int(t.last_collectives)))
    torch.distributed.destroy_process_group()


def test_two_rank_sparse_data_parallel_on_one_gpu():
    assert torch.cuda.is_available()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    outs = sorted([q.get(timeout=600) for _ in procs], key=lambda o: o[0])
    for p in procs: p.join(120)
    (_, p0a, ga, p1a, la, sa, ca), (_, p0b, gb, p1b, lb, sb, cb) = outs
    assert sorted(sa + sb) == [0, 1]                   # every event on exactly one rank
    # the all-reduce ran in TWO pieces: the decoder + bottom + head suffix from inside the backward pass (behind the
    # executor's side stream), the encoder prefix after it (parallel.OverlappedAllReduce)
    assert ca == 2 and cb == 2, (ca, cb)
    assert np.array_equal(p0a, p0b)                    # broadcast at initialize()
    assert np.array_equal(ga, gb)                      # identical summed gradient on both ranks
    assert np.array_equal(p1a, p1b)                    # replicas stay in sync after the step
    assert abs(la - lb) < 1e-9
    # single process, same initial weights: the two events one after the other, gradients summed
    from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss
    dev = torch.device('cuda:0')
    fl = _flags()
    net = SparseUResNet(fl).to(dev).train()
    off = 0
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.from_numpy(p0a[off:off + p.numel()]).view_as(p)); off += p.numel()
    crit = SparseSegmentationLoss(fl)
    blob = _blob()
    gsum = None
    for e in range(2):
        net.zero_grad(set_to_none=True)
        d = torch.from_numpy(blob['data'][0][e]).to(dev); lab = torch.from_numpy(blob['label'][0][e]).to(dev)
        loss, _ = crit(net(d), [d], [lab], None)
        loss.backward()
        gcur = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in net.parameters()]).cpu().numpy().copy()
        gsum = gcur if gsum is None else gsum + gcur
    err = float(np.linalg.norm(ga - gsum) / max(np.linalg.norm(gsum), 1e-30))
    assert err < 1e-5, err
