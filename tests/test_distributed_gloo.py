"""world_size-2 gloo tests (CPU) of the N>1 path: events sharded over ranks, ONE all-reduce(SUM) of the flat
gradient buffer, replicas stay in sync.  Uses the dense model (the sparse HIP path needs a GPU); the
parallel machinery is model-agnostic."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _flags():
    return SimpleNamespace(MODEL_NAME='uresnet_dense', DATA_DIM=2, URESNET_FILTERS=4, URESNET_NUM_STRIDES=2,
                           SPATIAL_SIZE=16, NUM_CLASS=3, BN_MOMENTUM=0.9, TRAIN=True, GPUS=[], LEARNING_RATE=1e-2,
                           MODEL_PATH='', WEIGHT_PREFIX='')


def _blob():
    from uresnet_pytorch_amd.iotools.synthetic import make_dense_blob
    b = make_dense_blob([0, 1], 16, 2, 3)
    return {'data': [[b['data'][0], b['data'][1]]], 'label': [[b['label'][0], b['label'][1]]]}


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from uresnet_pytorch_amd.trainval import trainval
    torch.manual_seed(100 + rank)            # different init per rank: initialize() must broadcast rank 0's
    t = trainval(_flags())
    t.initialize()
    p0 = torch.cat([p.detach().flatten() for p in t._net.parameters()]).clone()
    res = t.train_step(_blob(), epoch=0., batch_size=2)
    g = t._grads.flat.clone()
    p1 = torch.cat([p.detach().flatten() for p in t._net.parameters()]).clone()
    q.put((rank, p0.numpy(), g.numpy(), p1.numpy(), res['loss_seg'], res['accuracy']))
    torch.distributed.destroy_process_group()


def test_two_rank_data_parallel_matches_single_process():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    outs = sorted([q.get(timeout=240) for _ in procs], key=lambda o: o[0])
    for p in procs: p.join(60)
    (_, p0a, ga, p1a, la, aa), (_, p0b, gb, p1b, lb, ab) = outs
    assert np.array_equal(p0a, p0b)                   # broadcast at initialize()
    assert np.array_equal(ga, gb)                     # identical summed gradient on both ranks
    assert np.array_equal(p1a, p1b)                   # replicas stay in sync after the step
    assert abs(la - lb) < 1e-12 and abs(aa - ab) < 1e-12

    # single-process reference with the same initial weights and both events:
    # per-event batch statistics differ (BN is per replica, as in the reference's DataParallel), so run the two
    # events as two separate forward passes and SUM the gradients -- that is what the two ranks computed.
    from uresnet_pytorch_amd.models import DenseUResNet, DenseSegmentationLoss
    fl = _flags()
    net = DenseUResNet(fl).train()
    off = 0
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.from_numpy(p0a[off:off + p.numel()]).view_as(p)); off += p.numel()
    crit = DenseSegmentationLoss(fl)
    blob = _blob()
    total = 0.
    for e in range(2):
        x = torch.from_numpy(blob['data'][0][e])[None]; y = torch.from_numpy(blob['label'][0][e])
        loss, _ = crit(list(net(x)), [x[0]], [y], None)
        loss.backward(); total += loss.item()
    gref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in net.parameters()])
    assert np.allclose(ga, gref.numpy(), rtol=1e-4, atol=1e-6)
    assert abs(la - total / 2) < 1e-5                 # reported loss = sum over events / batch_size


def _uneven_blob():
    """five events of different sizes in one sub-step: LPT on the pixel counts must balance two ranks"""
    from uresnet_pytorch_amd.iotools.synthetic import make_dense_blob
    b = make_dense_blob([0, 1, 2, 3, 4], 16, 2, 3)
    return {'data': [[b['data'][i] for i in range(5)]], 'label': [[b['label'][i] for i in range(5)]]}


def _worker_lpt(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from uresnet_pytorch_amd.trainval import trainval
    t = trainval(_flags())
    t.initialize()
    res = t.train_step(_uneven_blob(), epoch=0., batch_size=5)
    p1 = torch.cat([p.detach().flatten() for p in t._net.parameters()]).clone()
    q.put((rank, list(t.last_slots), p1.numpy(), res['loss_seg']))
    torch.distributed.destroy_process_group()


def test_lpt_sharding_is_used_by_the_trainer_and_replicas_stay_in_sync():
    from uresnet_pytorch_amd import parallel
    # the assignment rule itself: largest first onto the least loaded rank, ties by index, per-rank lists sorted
    assert parallel.shard_events([50, 10, 40, 30, 20], 2) == [[0, 1, 4], [2, 3]]   # loads 80 / 70
    assert parallel.shard_events([7, 7, 7, 7], 4) == [[0], [1], [2], [3]]
    assert parallel.shard_events([5], 3) == [[0], [], []]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_lpt, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    outs = sorted([q.get(timeout=240) for _ in procs], key=lambda o: o[0])
    for p in procs: p.join(60)
    (_, s0, pa, la), (_, s1, pb, lb) = outs
    # five equal-size dense events over two ranks: LPT gives 3 + 2, every event exactly once
    assert sorted(s0 + s1) == [0, 1, 2, 3, 4] and {len(s0), len(s1)} == {2, 3}
    assert [s0, s1] == parallel.shard_events([256] * 5, 2)
    assert np.array_equal(pa, pb)                     # rank-identical parameters after the step
    assert abs(la - lb) < 1e-12


def _worker_idle(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from uresnet_pytorch_amd.trainval import trainval
    torch.manual_seed(100 + rank)            # as _worker: rank 0's initialisation is what everybody gets
    t = trainval(_flags())
    t.initialize()
    p0 = torch.cat([p.detach().flatten() for p in t._net.parameters()]).clone()
    res = t.train_step(_blob(), epoch=0., batch_size=2)     # TWO events, THREE ranks: one rank owns nothing
    g = t._grads.flat.clone()
    p1 = torch.cat([p.detach().flatten() for p in t._net.parameters()]).clone()
    q.put((rank, list(t.last_slots), p0.numpy(), g.numpy(), p1.numpy(), res['loss_seg'], res['accuracy'],
           len(res['segmentation'])))
    torch.distributed.destroy_process_group()


def test_rank_without_an_entry_contributes_zero_gradients():
    """Fewer events than ranks (shard_events([n, n], 3) -> [[0], [1], []]): the idle rank must not raise, contributes zeros
    to the gradient sum, takes part in the collective and the optimizer step, and ends with the same parameters."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_idle, args=(r, 3, port, q)) for r in range(3)]
    for p in procs: p.start()
    outs = sorted([q.get(timeout=240) for _ in procs], key=lambda o: o[0])
    for p in procs: p.join(60)
    slots = [o[1] for o in outs]
    assert slots == [[0], [1], []]
    assert outs[2][7] == 0                                                  # no segmentation entries on the idle rank
    for o in outs[1:]:
        assert np.array_equal(o[3], outs[0][3])                             # the same summed gradient everywhere
        assert np.array_equal(o[4], outs[0][4])                             # replicas in sync after the step
        assert abs(o[5] - outs[0][5]) < 1e-12 and abs(o[6] - outs[0][6]) < 1e-12
    assert not np.array_equal(outs[0][2], outs[0][4])                       # the step did move the parameters
    # equal to the two-rank run of the same blob (the idle rank adds exactly zero)
    q2 = ctx.Queue()
    port2 = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port2, q2)) for r in range(2)]
    for p in procs: p.start()
    two = sorted([q2.get(timeout=240) for _ in procs], key=lambda o: o[0])
    for p in procs: p.join(60)
    assert np.array_equal(two[0][1], outs[0][2])                            # same initial parameters (seeded)
    assert np.allclose(two[0][2], outs[0][3], rtol=1e-6, atol=1e-8)
