"""train / inference / iotest drivers with the reference's call contract into the trainer
(reference uresnet/main_funcs.py:17-259).  Host orchestration only; of the full-inference physics
metrics (reference :262-466) the per-event accuracy / per-class / confusion part is available
(utils.compute_metrics_*, written by inference_loop to a CSV), the Michel-electron analysis is out of scope."""
import datetime
import os
import sys
import time

import numpy as np
import torch

from . import utils
from .iotools import io_factory
from .trainval import trainval


class Handlers:
    data_io = None
    csv_logger = None
    train_logger = None
    trainer = None
    iteration = 0


def iotest(flags):
    io = io_factory(flags)
    io.initialize()
    num_entries = io.num_entries()
    ctr = 0
    data_key = flags.DATA_KEYS[0] or 'data'
    while ctr < num_entries:
        idx, blob = io.next()
        print('%d/%d ... %s %s' % (ctr, num_entries, str(idx), str(blob[data_key][0].shape)))
        ctr += io.batch_per_step()
    io.finalize()


def train(flags):
    flags.TRAIN = True
    handlers = prepare(flags)
    train_loop(flags, handlers)


def inference(flags):
    flags.TRAIN = False
    handlers = prepare(flags)
    inference_loop(flags, handlers)


def prepare(flags):
    handlers = Handlers()
    handlers.data_io = io_factory(flags)
    handlers.data_io.initialize()
    if 'sparse' in flags.IO_TYPE:
        handlers.data_io.start_threads()
    if 'sparse' in flags.MODEL_NAME and 'sparse' not in flags.IO_TYPE:
        sys.stderr.write('Sparse UResNet needs sparse IO.')
        sys.exit(1)
    flags.NUM_CHANNEL = handlers.data_io.num_channels()
    handlers.trainer = trainval(flags)
    loaded_iteration = handlers.trainer.initialize()
    handlers.iteration = loaded_iteration if flags.TRAIN else 0
    if flags.WEIGHT_PREFIX:
        save_dir = flags.WEIGHT_PREFIX[0:flags.WEIGHT_PREFIX.rfind('/')]
        if save_dir and not os.path.isdir(save_dir):
            os.makedirs(save_dir, exist_ok=True)
    if flags.LOG_DIR:
        os.makedirs(flags.LOG_DIR, exist_ok=True)
        logname = '%s/%s_log-%07d.csv' % (flags.LOG_DIR, 'train' if flags.TRAIN else 'inference', loaded_iteration)
        handlers.csv_logger = utils.CSVData(logname)
    return handlers


def get_keys(flags):
    keys = [k for k in flags.DATA_KEYS if k] or ['data', 'label']
    return keys[0], (keys[1] if len(keys) > 1 else None), (keys[2] if len(keys) > 2 else None)


def log(handlers, tstamp_iteration, tspent_iteration, tsum, res, flags, epoch):
    """Same CSV columns as reference main_funcs.py:128-143."""
    report_step = flags.REPORT_STEP and ((handlers.iteration + 1) % flags.REPORT_STEP == 0)
    loss_seg = float(np.mean(res['loss_seg']))
    acc_seg = float(np.mean(res['accuracy']))
    mem = utils.round_decimals(torch.cuda.max_memory_allocated() / 1.e9, 3) if torch.cuda.is_available() else 0.
    if handlers.csv_logger:
        lg = handlers.csv_logger
        lg.record(('iter', 'epoch', 'titer', 'tsumiter'), (handlers.iteration, epoch, tspent_iteration, tsum))
        lg.record(('tio', 'tsumio'), (handlers.data_io.tspent_io, handlers.data_io.tspent_sum_io))
        lg.record(('mem',), (mem,))
        tmap, tsum_map = handlers.trainer.tspent, handlers.trainer.tspent_sum
        if flags.TRAIN:
            lg.record(('ttrain', 'tsave', 'tsumtrain', 'tsumsave'),
                      (tmap['train'], tmap['save'], tsum_map['train'], tsum_map['save']))
        lg.record(('tforward', 'tsave', 'tsumforward', 'tsumsave'),
                  (tmap['forward'], tmap['save'], tsum_map['forward'], tsum_map['save']))
        lg.record(('loss_seg', 'acc_seg'), (loss_seg, acc_seg))
        lg.write()
    if report_step:
        tmap = handlers.trainer.tspent
        key = 'train' if flags.TRAIN else 'forward'
        print('Iter. %d (epoch %g) @ %s ... %s time %g%% (%g [s]) mem. %g GB' % (
            handlers.iteration, utils.round_decimals(epoch, 2), tstamp_iteration, key,
            utils.round_decimals(tmap[key] / tspent_iteration * 100., 2), utils.round_decimals(tmap[key], 3), mem))
        print('   Segmentation: loss %g accuracy %g' % (utils.round_decimals(loss_seg, 4), acc_seg))
        sys.stdout.flush()
        if handlers.csv_logger:
            handlers.csv_logger.flush()


def get_data_minibatched(handlers, flags, data_key, label_key, weight_key):
    """Reference main_funcs.py:166-181: BATCH_SIZE / (MINIBATCH_SIZE * nGPU) sub-steps."""
    data_blob = {'data': [], 'idx_v': []}
    if label_key is not None: data_blob['label'] = []
    if weight_key is not None: data_blob['weight'] = []
    for _ in range(int(flags.BATCH_SIZE / (flags.MINIBATCH_SIZE * max(1, len(flags.GPUS))))):
        idx, blob = handlers.data_io.next()
        data_blob['data'].append(blob[data_key])
        data_blob['idx_v'].append(idx)
        if label_key is not None: data_blob['label'].append(blob[label_key])
        if weight_key is not None: data_blob['weight'].append(blob[weight_key])
    return data_blob


def train_loop(flags, handlers):
    data_key, label_key, weight_key = get_keys(flags)
    tsum = 0.
    while handlers.iteration < flags.ITERATION:
        epoch = handlers.iteration * float(flags.BATCH_SIZE) / handlers.data_io.num_entries()
        tstamp_iteration = datetime.datetime.fromtimestamp(time.time()).strftime('%Y-%m-%d %H:%M:%S')
        tstart_iteration = time.time()
        checkpt_step = flags.CHECKPOINT_STEP and flags.WEIGHT_PREFIX and \
            ((handlers.iteration + 1) % flags.CHECKPOINT_STEP == 0)
        data_blob = get_data_minibatched(handlers, flags, data_key, label_key, weight_key)
        res = handlers.trainer.train_step(data_blob, epoch=float(epoch), batch_size=flags.BATCH_SIZE)
        if checkpt_step:
            handlers.trainer.save_state(handlers.iteration)
        tspent_iteration = time.time() - tstart_iteration
        tsum += tspent_iteration
        log(handlers, tstamp_iteration, tspent_iteration, tsum, res, flags, epoch)
        handlers.iteration += 1
    if handlers.csv_logger:
        handlers.csv_logger.close()
    handlers.data_io.finalize()


def log_metrics(handlers, flags, data_blob, res):
    """Per-event inference metrics (utils.compute_metrics_*; the non-Michel part of what the reference's
    full_inference_loop records, main_funcs.py:262-466) into <LOG_DIR>/inference_metrics-*.csv."""
    if not flags.LOG_DIR or 'label' not in data_blob or 'softmax' not in res:
        return
    if getattr(handlers, 'metrics_logger', None) is None:
        handlers.metrics_logger = utils.CSVData('%s/inference_metrics-%07d.csv' % (flags.LOG_DIR, handlers.iteration))
    lg = handlers.metrics_logger
    host = lambda a: a.detach().cpu().numpy() if torch.is_tensor(a) else a     # (-iod: the blob lives on the device)
    data_v = [host(d) for sub in data_blob['data'] for d in sub]
    label_v = [host(d) for sub in data_blob['label'] for d in sub]
    soft_v = list(res['softmax'])
    if 'sparse' in flags.MODEL_NAME:
        m, _ = utils.compute_metrics_sparse(data_v, label_v, soft_v, None, N=flags.SPATIAL_SIZE)
    else:
        m = utils.compute_metrics_dense(data_v, label_v, soft_v, None)   # one (C, [D,] H, W) array per event
    for e in range(len(m['acc'])):
        lg.record(('iter', 'id', 'acc', 'correct_softmax', 'nonzero_pixels'),
                  (handlers.iteration, m['id'][e], m['acc'][e], m['correct_softmax'][e], m['nonzero_pixels'][e]))
        for c, a in enumerate(m['class_acc'][e]):
            lg.record(('class_acc_%d' % c, 'class_pixel_%d' % c), (float(np.nan_to_num(a)), float(m['class_pixel'][e][c])))
        lg.write()
    lg.flush()


def inference_loop(flags, handlers):
    data_key, label_key, weight_key = get_keys(flags)
    tsum = 0.
    while handlers.iteration < flags.ITERATION:
        epoch = handlers.iteration * float(flags.BATCH_SIZE) / handlers.data_io.num_entries()
        tstamp_iteration = datetime.datetime.fromtimestamp(time.time()).strftime('%Y-%m-%d %H:%M:%S')
        tstart_iteration = time.time()
        data_blob = get_data_minibatched(handlers, flags, data_key, label_key, weight_key)
        res = handlers.trainer.forward(data_blob, epoch=float(epoch), batch_size=flags.BATCH_SIZE)
        log_metrics(handlers, flags, data_blob, res)
        # Store output if requested (reference :248-249); readers without a writer (synthetic) have no store_segment
        if flags.OUTPUT_FILE and hasattr(handlers.data_io, 'store_segment') and handlers.trainer._world == 1:
            at = 0
            for sub, idx in enumerate(data_blob['idx_v']):
                n_entries = len(data_blob['data'][sub])
                handlers.data_io.store_segment(idx, data_blob['data'][sub], res['softmax'][at:at + n_entries])
                at += n_entries
        tspent_iteration = time.time() - tstart_iteration
        tsum += tspent_iteration
        log(handlers, tstamp_iteration, tspent_iteration, tsum, res, flags, epoch)
        handlers.iteration += 1
    if handlers.csv_logger:
        handlers.csv_logger.close()
    handlers.data_io.finalize()
