"""Array-backed readers behind io_factory: '-io npz_sparse' / '-io npz_dense'.

They serve the blob contract of the reference's LArCV readers (reference uresnet/iotools/iotools_sparse.py:120-160,
iotools_dense.py:195-198, io_base.py:7-67) from plain .npz files, so that `bin/uresnet.py train|inference|iotest`
run end-to-end on real arrays without LArCV/ROOT:

    next() -> (idx_per_gpu, blob)
    sparse: blob[DATA_KEYS[0]][gpu] = (N, d+2) float32 rows [coords.., batch_id, value], batch_id = position of the event
            in the step's batch (reference :141 `constant_values=data_id`); blob[key][gpu] = (N, 1) for the other keys;
            '_weights_' / -cw: N / (n_classes * count_c) per event (reference :311-318)
    dense:  blob[key] = list of (C, [D,] H, W) arrays, one per event of the step (reference iotools_dense.py:195-198)

File layout (written by write_sparse_npz / write_dense_npz; loaded with numpy.load(allow_pickle=False)):
    sparse: voxels (sum N, d) int32 | feature (sum N, 1) float32 | offsets (E + 1,) int64 | one (sum N, 1) float32 array
            per further data key (e.g. 'label')
    dense:  one (E, C, [D,] H, W) float32 array per data key

The per-GPU concatenation of a step's events can run ON THE DEVICE (flags.IO_ON_DEVICE, -iod): the file's arrays are
uploaded once and every blob entry is assembled by device copies -- no host concat, no H2D copy per step.
"""
import os

import numpy as np
import torch

from .iotools import io_base


def write_sparse_npz(path, events, extra_keys=('label',)):
    """events: list of dicts {'voxels': (n, d) ints, 'feature': (n,) or (n, 1), <extra key>: (n,) or (n, 1)}."""
    off = np.zeros(len(events) + 1, np.int64)
    for i, e in enumerate(events):
        off[i + 1] = off[i] + len(e['voxels'])
    out = {'voxels': np.concatenate([np.asarray(e['voxels'], np.int32) for e in events], 0),
           'feature': np.concatenate([np.asarray(e['feature'], np.float32).reshape(-1, 1) for e in events], 0),
           'offsets': off}
    for k in extra_keys:
        out[k] = np.concatenate([np.asarray(e[k], np.float32).reshape(-1, 1) for e in events], 0)
    np.savez_compressed(path, **out)


def write_dense_npz(path, arrays):
    """arrays: dict key -> (E, C, [D,] H, W)."""
    np.savez_compressed(path, **{k: np.asarray(v, np.float32) for k, v in arrays.items()})


def class_weights(labels):
    """Reference iotools_sparse.py:311-318, including its indexing: the c-th class found gets its weight written where
    the label EQUALS c (identical to per-class balancing whenever the labels present are 0..k-1)."""
    labels = np.asarray(labels)
    weights = np.zeros(shape=labels.shape, dtype=np.float32)
    classes, counts = np.unique(labels, return_counts=True)
    for c in range(len(classes)):
        idx = np.where(labels == float(c))[0]
        weights[idx] = float(len(labels)) / (len(classes)) / counts[c]
    return weights


class _io_array(io_base):
    def __init__(self, flags):
        super(_io_array, self).__init__(flags)
        self._start = 0
        self._order = None
        self._stored = {}

    def _keys(self):
        return [k for k in self._flags.DATA_KEYS if k] or ['data', 'label']

    def _files(self):
        files = [f for f in self._flags.INPUT_FILE if f]
        if not files:
            raise ValueError('array-backed IO needs -if <file.npz>[,<file.npz>...]')
        for f in files:
            if not os.path.isfile(f):
                raise IOError('input file not found: %s' % f)
        return files

    def _finish_init(self, n):
        lim = int(getattr(self._flags, 'LIMIT_NUM_SAMPLE', -1))
        self._num_entries = min(n, lim) if lim > 0 else n
        self._order = np.arange(self._num_entries)
        if getattr(self._flags, 'SHUFFLE', 0):
            self._order = np.random.permutation(self._num_entries)
        dev = getattr(self._flags, 'IO_ON_DEVICE', False)
        self._device = torch.device('cuda', torch.cuda.current_device()) if (dev and torch.cuda.is_available()) else None

    def set_index_start(self, idx):
        self._start = int(idx) % max(self._num_entries, 1)

    def _step_indices(self):
        n = self._minibatch_per_step
        pos = (self._start + np.arange(n)) % self._num_entries      # wraps like the reference's circular buffer (:118-122)
        self._start = int((self._start + n) % self._num_entries)
        return self._order[pos]

    def store_segment(self, idx, data, softmax):
        """Keeps the prediction of every event of one step for finalize() (the reference's store_segment writes them to its
        output file, iotools_sparse.py:369-416): argmax and per-class scores per voxel / pixel, keyed by entry index.
        sparse: idx / data / softmax are per-GPU lists ((n_events,), (N, d+2), (N, nc)); rows are split by the batch column.
        dense: idx (n,), softmax a list of (nc, [D,] H, W)."""
        host = lambda a: a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
        if 'sparse' in self._flags.IO_TYPE:
            d = self._flags.DATA_DIM
            for ids, rows, soft in zip(idx, data, softmax):
                rows, soft = host(rows), host(soft)
                bids = rows[:, d]
                for e, b in zip(np.asarray(ids).reshape(-1).tolist(), np.unique(bids).tolist()):   # both ascending in the step
                    sel = bids == b
                    self._stored[int(e)] = (np.argmax(soft[sel], axis=1).astype(np.int32), soft[sel].astype(np.float32))
            return
        for i, s in zip(np.asarray(idx).reshape(-1).tolist(), softmax):
            s = host(s)
            self._stored[int(i)] = (np.argmax(s, axis=0).astype(np.int32), s.astype(np.float32))

    def finalize(self):
        out = getattr(self._flags, 'OUTPUT_FILE', '')
        if out and self._stored:
            arrays = {}
            for i, (pred, soft) in sorted(self._stored.items()):
                arrays['prediction/%d' % i] = pred
                arrays['softmax/%d' % i] = soft
            np.savez_compressed(out, **arrays)
        self._stored = {}


class io_npz_sparse(_io_array):
    def initialize(self):
        f = self._flags
        keys = self._keys()
        compute_w = bool(getattr(f, 'COMPUTE_WEIGHT', False)) and len(keys) > 2
        read_keys = [k for k in keys[1:] if not (compute_w and k == keys[2])]
        vox, feat, offs, extra = [], [], [0], {k: [] for k in read_keys}
        for path in self._files():
            z = np.load(path, allow_pickle=False)
            for k in ['voxels', 'feature', 'offsets'] + read_keys:
                if k not in z.files:
                    raise KeyError('%s: array %r missing (has %s)' % (path, k, ', '.join(z.files)))
            o = z['offsets'].astype(np.int64)
            vox.append(z['voxels'].astype(np.int32)); feat.append(z['feature'].astype(np.float32).reshape(-1, 1))
            offs.extend((o[1:] + offs[-1]).tolist())
            for k in read_keys:
                extra[k].append(z[k].astype(np.float32).reshape(-1, 1))
        self._vox = np.concatenate(vox, 0); self._feat = np.concatenate(feat, 0)
        self._off = np.asarray(offs, np.int64)
        if self._vox.shape[1] != f.DATA_DIM:
            raise ValueError('file holds %d-d voxels, flags say -dd %d' % (self._vox.shape[1], f.DATA_DIM))
        self._extra = {k: np.concatenate(v, 0) for k, v in extra.items()}
        if compute_w:
            lab = self._extra[keys[1]]
            w = np.zeros_like(lab)
            for e in range(len(self._off) - 1):
                s, t = self._off[e], self._off[e + 1]
                w[s:t] = class_weights(lab[s:t])
            self._extra[keys[2]] = w
        self._num_channels = 1
        self._finish_init(len(self._off) - 1)
        # one (sum N, d + 2) row table [coords.., 0, value]: a blob entry is a concat of row ranges + the batch column
        self._rows = np.concatenate([self._vox.astype(np.float32), np.zeros((len(self._vox), 1), np.float32), self._feat], 1)
        if self._device is not None:
            self._rows_d = torch.from_numpy(self._rows).to(self._device)
            self._extra_d = {k: torch.from_numpy(v).to(self._device) for k, v in self._extra.items()}

    def _next(self, buffer_id=-1, release=True):
        keys = self._keys()
        d = self._flags.DATA_DIM
        ngpu = max(1, len(self._flags.GPUS))
        idx = self._step_indices()
        blob = {k: [] for k in keys}
        idx_v = []
        for g in range(ngpu):
            mine = idx[g * self._minibatch_per_gpu:(g + 1) * self._minibatch_per_gpu]
            ranges = [(int(self._off[e]), int(self._off[e + 1])) for e in mine]
            ids = np.arange(g * self._minibatch_per_gpu, g * self._minibatch_per_gpu + len(mine))   # data_id of the step
            if self._device is not None:
                rows = torch.cat([self._rows_d[s:t] for s, t in ranges], 0)
                lens = torch.tensor([t - s for s, t in ranges], device=self._device)
                rows[:, d] = torch.repeat_interleave(torch.as_tensor(ids, dtype=torch.float32, device=self._device), lens)
                blob[keys[0]].append(rows)
                for k in keys[1:]:
                    blob[k].append(torch.cat([self._extra_d[k][s:t] for s, t in ranges], 0))
            else:
                rows = np.concatenate([self._rows[s:t] for s, t in ranges], 0)
                rows[:, d] = np.repeat(ids.astype(np.float32), [t - s for s, t in ranges])
                blob[keys[0]].append(rows)
                for k in keys[1:]:
                    blob[k].append(np.concatenate([self._extra[k][s:t] for s, t in ranges], 0))
            idx_v.append(np.asarray(mine))
        return idx_v, blob


class io_npz_dense(_io_array):
    def initialize(self):
        keys = self._keys()
        arrays = {k: [] for k in keys}
        for path in self._files():
            z = np.load(path, allow_pickle=False)
            for k in keys:
                if k not in z.files:
                    raise KeyError('%s: array %r missing (has %s)' % (path, k, ', '.join(z.files)))
                arrays[k].append(z[k].astype(np.float32))
        self._arr = {k: np.concatenate(v, 0) for k, v in arrays.items()}
        first = self._arr[keys[0]]
        if first.ndim != self._flags.DATA_DIM + 2:
            raise ValueError('file holds %s arrays, flags say -dd %d' % (first.shape, self._flags.DATA_DIM))
        self._num_channels = int(first.shape[1])
        self._finish_init(len(first))
        if self._device is not None:
            self._arr_d = {k: torch.from_numpy(v).to(self._device) for k, v in self._arr.items()}

    def _next(self, buffer_id=-1, release=True):
        idx = self._step_indices()
        src = self._arr_d if self._device is not None else self._arr
        return np.asarray(idx), {k: [src[k][int(i)] for i in idx] for k in self._keys()}
