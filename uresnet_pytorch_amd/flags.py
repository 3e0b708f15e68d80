"""Command-line flags with the reference's names (reference uresnet/flags.py:12-196): three
sub-commands train | inference | iotest, every option and attribute name identical (attributes
are the UPPER-CASED argument names, :155-158).  Fixes kept deliberate:
py3 dict iteration, `sys` imported, seed cast to int, --gpus mapped to HIP_VISIBLE_DEVICES as well."""
import argparse
import os
import sys
import time

import numpy as np
import torch


def _strtobool(v):
    if isinstance(v, bool):
        return v
    if str(v).lower() in ('y', 'yes', 't', 'true', 'on', '1'):
        return True
    if str(v).lower() in ('n', 'no', 'f', 'false', 'off', '0'):
        return False
    raise argparse.ArgumentTypeError('invalid truth value %r' % v)


class URESNET_FLAGS:
    # model
    NUM_CLASS = 2
    MODEL_NAME = ""
    TRAIN = True
    DEBUG = False
    FULL = False
    # Sparse UResNet model
    URESNET_NUM_STRIDES = 3
    URESNET_FILTERS = 16
    SPATIAL_SIZE = 192
    BN_MOMENTUM = 0.9
    # train/inference
    COMPUTE_WEIGHT = False
    SEED = -1
    LEARNING_RATE = 0.001
    GPUS = []
    WEIGHT_PREFIX = ''
    NUM_POINT = 2048
    PRECISION = 'fp32'
    LOSS_SCALE = 1.0
    GRAPH = False
    CKPT_MODULE_PREFIX = False
    IO_ON_DEVICE = False
    NUM_CHANNEL = -1
    ITERATION = 10000
    REPORT_STEP = 100
    CHECKPOINT_STEP = 500
    # IO
    IO_TYPE = ''
    INPUT_FILE = ''
    OUTPUT_FILE = ''
    MINIBATCH_SIZE = -1
    BATCH_SIZE = -1
    LOG_DIR = ''
    MODEL_PATH = ''
    DATA_KEYS = ''
    SHUFFLE = 1
    LIMIT_NUM_SAMPLE = -1
    NUM_THREADS = 1
    DATA_DIM = 3
    PARTICLE = False

    def __init__(self):
        self._build_parsers()

    def _attach_common_args(self, parser):
        a = parser.add_argument
        a('-db', '--debug', type=_strtobool, default=self.DEBUG, help='Extra verbose mode for debugging')
        a('-ld', '--log_dir', default=self.LOG_DIR, help='Log dir')
        a('-sh', '--shuffle', type=_strtobool, default=self.SHUFFLE, help='Shuffle the data entries')
        a('--gpus', type=str, default='', help='GPUs to utilize (comma-separated integers)')
        a('-nc', '--num_class', type=int, default=self.NUM_CLASS, help='Number of classes')
        a('-it', '--iteration', type=int, default=self.ITERATION, help='Iteration to run')
        a('-bs', '--batch_size', type=int, default=self.BATCH_SIZE, help='Batch size for updating weights')
        a('-mbs', '--minibatch_size', type=int, default=self.MINIBATCH_SIZE, help='Mini-batch size (sample/gpu)')
        a('-rs', '--report_step', type=int, default=self.REPORT_STEP, help='Period (steps) to print loss/accuracy')
        a('-mn', '--model_name', type=str, default=self.MODEL_NAME, help='model name identifier')
        a('-mp', '--model_path', type=str, default=self.MODEL_PATH, help='model checkpoint file path')
        a('-io', '--io_type', type=str, default=self.IO_TYPE, help='IO handler type')
        a('-if', '--input_file', type=str, default=self.INPUT_FILE, help='comma-separated input file list')
        a('-of', '--output_file', type=str, default=self.OUTPUT_FILE, help='output file name')
        a('-dkeys', '--data_keys', type=str, default=self.DATA_KEYS, help='keywords to fetch data from file')
        a('-lns', '--limit_num_sample', type=int, default=self.LIMIT_NUM_SAMPLE, help='Limit number of samples')
        a('-nt', '--num-threads', type=int, default=self.NUM_THREADS, help='Number of threads to read input')
        a('-dd', '--data-dim', type=int, default=self.DATA_DIM, help='Data dimension')
        a('-ss', '--spatial_size', type=int, default=self.SPATIAL_SIZE, help='Length of one side of the data')
        a('-uns', '--uresnet-num-strides', type=int, default=self.URESNET_NUM_STRIDES, help='Depth for UResNet')
        a('-uf', '--uresnet-filters', type=int, default=self.URESNET_FILTERS, help='Number of base filters')
        a('-bnm', '--bn-momentum', type=float, default=self.BN_MOMENTUM, help='BatchNorm momentum')
        a('-cw', '--compute_weight', default=self.COMPUTE_WEIGHT, action='store_true',
          help='Compute pixel loss weighting factor on the fly')
        a('-sd', '--seed', default=self.SEED, help='Seed for random number generators')
        a('-np', '--num_point', type=int, default=self.NUM_POINT, help='Active voxels per synthetic event')
        a('-prec', '--precision', type=str, default=self.PRECISION, choices=['fp32', 'bf16', 'fp16'],
          help='MFMA operand precision of the convolutions on the GPU (tensors and accumulation stay fp32)')
        a('-ls', '--loss_scale', type=float, default=self.LOSS_SCALE,
          help='loss scale of a training step (a power of two; gradients are unscaled before the optimizer): keeps the '
               'gradient operands of -prec fp16 inside its exponent range')
        a('-graph', '--graph', action='store_true', default=self.GRAPH,
          help='dense model on the GPU: replay the training step (forward + loss + backward) from a captured HIP graph '
               '(one sub-step of fixed shape per iteration; anything else runs eagerly)')
        a('-iod', '--io_on_device', action='store_true', default=self.IO_ON_DEVICE,
          help='array-backed IO (-io npz_sparse / npz_dense): keep the file arrays in GPU memory and assemble every per-GPU '
               'blob entry there (no host concatenation, no H2D copy per step)')
        a('-cmp', '--ckpt_module_prefix', action='store_true', default=self.CKPT_MODULE_PREFIX,
          help="write checkpoints with the reference's DataParallel key prefix ('module.<name>', reference "
               "trainval.py:37,181) so that they load into the reference with strict=False; checkpoints with or without "
               "the prefix are both read")
        return parser

    def _build_parsers(self):
        from .main_funcs import train, iotest, inference
        self.parser = argparse.ArgumentParser(description="U-ResNet configuration flags")
        subparsers = self.parser.add_subparsers(title="Modules", description="Valid subcommands", dest='script')
        train_parser = subparsers.add_parser("train", help="Train")
        train_parser.add_argument('-wp', '--weight_prefix', default=self.WEIGHT_PREFIX,
                                  help='Prefix (directory + file prefix) for snapshots of weights')
        train_parser.add_argument('-lr', '--learning_rate', type=float, default=self.LEARNING_RATE)
        train_parser.add_argument('-chks', '--checkpoint_step', type=int, default=self.CHECKPOINT_STEP)
        inference_parser = subparsers.add_parser("inference", help="Run inference")
        inference_parser.add_argument('-full', '--full', default=self.FULL, action='store_true')
        inference_parser.add_argument('-p', '--particle', default=self.PARTICLE, action='store_true')
        iotest_parser = subparsers.add_parser("iotest", help="Test iotools")
        self.train_parser = self._attach_common_args(train_parser)
        self.inference_parser = self._attach_common_args(inference_parser)
        self.iotest_parser = self._attach_common_args(iotest_parser)
        self.train_parser.set_defaults(func=train)
        self.inference_parser.set_defaults(func=inference)
        self.iotest_parser.set_defaults(func=iotest)

    def parse_args(self, argv=None, run=True):
        args = self.parser.parse_args(argv)
        self.update(vars(args))
        print("\n\n-- CONFIG --")
        for name in vars(self):
            attribute = getattr(self, name)
            if isinstance(attribute, argparse.ArgumentParser):
                continue
            print("%s = %r" % (name, attribute))
        np.random.seed(self.SEED % (2 ** 32))
        torch.manual_seed(self.SEED)
        if run:
            args.func(self)
        return self

    def update(self, args):
        for name, value in args.items():
            if name in ['func', 'script']:
                continue
            setattr(self, name.upper(), value)
        if int(os.environ.get('WORLD_SIZE', '1')) == 1:
            # single process: expose exactly the requested devices (reference :160)
            os.environ['CUDA_VISIBLE_DEVICES'] = self.GPUS
            os.environ['HIP_VISIBLE_DEVICES'] = self.GPUS
        self.GPUS = list(range(len(self.GPUS.split(',')))) if len(self.GPUS) > 0 else []
        self.INPUT_FILE = [str(f) for f in self.INPUT_FILE.split(',')]
        self.DATA_KEYS = self.DATA_KEYS.split(',')
        ls = float(getattr(self, 'LOSS_SCALE', 1.0) or 1.0)
        if not (ls > 0 and np.isfinite(ls) and np.frexp(ls)[0] == 0.5):
            sys.stderr.write('ERROR: -ls / --loss_scale must be a positive power of two (got %r): only then are scaling and '
                             'unscaling exact\n' % ls)
            raise ValueError
        self.SEED = int(self.SEED)
        if self.SEED < 0:
            self.SEED = int(time.time())
        if self.BATCH_SIZE < 0 and self.MINIBATCH_SIZE < 0:
            print('Cannot have both BATCH_SIZE (-bs) and MINIBATCH_SIZE (-mbs) negative values!')
            raise ValueError
        if self.BATCH_SIZE < 0:
            self.BATCH_SIZE = int(self.MINIBATCH_SIZE * max(1, len(self.GPUS)))
        if self.MINIBATCH_SIZE < 0:
            self.MINIBATCH_SIZE = int(self.BATCH_SIZE / max(1, len(self.GPUS)))
        if not (self.BATCH_SIZE % (self.MINIBATCH_SIZE * max(1, len(self.GPUS)))) == 0:
            print('BATCH_SIZE (-bs) must be multiples of MINIBATCH_SIZE (-mbs) and GPU count (--gpus)!')
            raise ValueError
        if self.COMPUTE_WEIGHT:
            if len(self.DATA_KEYS) > 2:
                sys.stderr.write('ERROR: cannot compute weight if producer is specified ("%s")\n' % self.DATA_KEYS[2])
                raise KeyError
            if '_weights_' in self.DATA_KEYS:
                sys.stderr.write('ERROR: cannot compute weight if any data has label "_weights_"\n')
                raise KeyError
            if len(self.DATA_KEYS) < 2:
                sys.stderr.write('ERROR: you must provide data and label (2 data product keys) to compute weights\n')
                raise KeyError
            self.DATA_KEYS.append('_weights_')
