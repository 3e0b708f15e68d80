"""io_factory with the reference's contract (reference uresnet/iotools/iotools.py:5-10,
io_base.py:7-67).  The LArCV/ROOT readers are out of scope (larcv and ROOT are absent);
'npz_sparse' / 'npz_dense' (array_io.py) read plain array files; 'synthetic_sparse' / 'synthetic_dense' serve the same blob
layout from the seeded generator:
    next() -> (idx_per_gpu, blob), blob[data_key][gpu] = (N, d+2) [coords.., batch_id, value]
    (reference iotools_sparse.py:141,152-160), blob[label_key][gpu] = (N, 1), optional
    '_weights_' class balancing (reference :311-318); dense (B? , C, [D,] H, W) per entry.
"""
import time

import numpy as np

from . import synthetic


class io_base(object):
    def __init__(self, flags):
        ngpu = max(1, len(flags.GPUS))
        if not flags.BATCH_SIZE % (flags.MINIBATCH_SIZE * ngpu) == 0:
            print('BATCH_SIZE (%d) must be divisible by GPU count (%d) times MINIBATCH_SIZE(%d)'
                  % (flags.BATCH_SIZE, len(flags.GPUS), flags.MINIBATCH_SIZE))
            raise ValueError
        self._minibatch_per_step = flags.MINIBATCH_SIZE * ngpu
        self._minibatch_per_gpu = flags.MINIBATCH_SIZE
        self._num_entries = -1
        self._num_channels = -1
        self._flags = flags
        self._blob = {}
        self.tspent_io = 0
        self.tspent_sum_io = 0

    def blob(self): return self._blob
    def batch_per_step(self): return self._minibatch_per_step
    def batch_per_gpu(self): return self._minibatch_per_gpu
    def num_entries(self): return self._num_entries
    def num_channels(self): return self._num_channels
    def initialize(self): raise NotImplementedError
    def start_threads(self): pass
    def stop_threads(self): pass

    def next(self, buffer_id=-1, release=True):
        tstart = time.time()
        res = self._next(buffer_id, release)
        self.tspent_io = time.time() - tstart
        self.tspent_sum_io += self.tspent_io
        return res

    def finalize(self): pass


class io_synthetic_sparse(io_base):
    def initialize(self):
        f = self._flags
        self._num_entries = f.LIMIT_NUM_SAMPLE if f.LIMIT_NUM_SAMPLE > 0 else 64
        self._num_channels = 1
        self._voxels = int(getattr(f, 'NUM_POINT', 2048))
        self._cursor = 0
        keys = [k for k in f.DATA_KEYS if k] or ['data', 'label']
        self._keys = keys

    def _next(self, buffer_id=-1, release=True):
        f = self._flags
        ngpu = max(1, len(f.GPUS))
        blob = {k: [] for k in self._keys}
        idx_v = []
        for g in range(ngpu):
            seeds = [(self._cursor + g * self._minibatch_per_gpu + b) % self._num_entries
                     for b in range(self._minibatch_per_gpu)]
            b = synthetic.make_sparse_blob(seeds, f.SPATIAL_SIZE, self._voxels, compute_weight=len(self._keys) > 2)
            blob[self._keys[0]].append(b['data'])
            if len(self._keys) > 1:
                blob[self._keys[1]].append(b['label'])
            if len(self._keys) > 2:
                blob[self._keys[2]].append(b['weight'])
            idx_v.append(np.asarray(seeds))
        self._cursor = (self._cursor + self._minibatch_per_step) % self._num_entries
        return idx_v, blob


class io_synthetic_dense(io_base):
    def initialize(self):
        f = self._flags
        self._num_entries = f.LIMIT_NUM_SAMPLE if f.LIMIT_NUM_SAMPLE > 0 else 64
        self._num_channels = 1
        self._cursor = 0
        self._keys = [k for k in f.DATA_KEYS if k] or ['data', 'label']

    def _next(self, buffer_id=-1, release=True):
        f = self._flags
        n = self._minibatch_per_step
        seeds = [(self._cursor + i) % self._num_entries for i in range(n)]
        b = synthetic.make_dense_blob(seeds, f.SPATIAL_SIZE, f.DATA_DIM, f.NUM_CLASS)
        self._cursor = (self._cursor + n) % self._num_entries
        blob = {self._keys[0]: [b['data'][i] for i in range(n)]}
        if len(self._keys) > 1:
            blob[self._keys[1]] = [b['label'][i] for i in range(n)]
        return np.asarray(seeds), blob


def io_factory(flags):
    if flags.IO_TYPE in ('synthetic_sparse',):
        return io_synthetic_sparse(flags)
    if flags.IO_TYPE in ('synthetic_dense',):
        return io_synthetic_dense(flags)
    if flags.IO_TYPE in ('npz_sparse', 'npz_dense'):
        # array-backed readers with the same blob contract (array_io.py): real data without LArCV/ROOT
        from . import array_io
        return (array_io.io_npz_sparse if flags.IO_TYPE == 'npz_sparse' else array_io.io_npz_dense)(flags)
    if flags.IO_TYPE in ('larcv_sparse', 'larcv_dense'):
        raise NotImplementedError('LArCV/ROOT readers are out of scope of this build (larcv is not installed); '
                                  'use -io npz_sparse / npz_dense (array files) or synthetic_sparse / synthetic_dense')
    raise NotImplementedError
