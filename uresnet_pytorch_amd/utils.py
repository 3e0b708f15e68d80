"""CSV logger with the reference's interface (reference uresnet/utils.py:13-44): header from
the first write(), every value formatted '{:f}'."""


class CSVData:
    def __init__(self, fout):
        self._fout = fout
        self._str = None
        self._dict = {}

    def record(self, keys, vals):
        for i, key in enumerate(keys):
            self._dict[key] = vals[i]

    def write(self):
        if self._str is None:
            self._fout = open(self._fout, 'w')
            self._fout.write(','.join(self._dict.keys()) + '\n')
            self._str = ','.join(['{:f}'] * len(self._dict)) + '\n'
        self._fout.write(self._str.format(*(self._dict.values())))

    def flush(self):
        if self._str is not None:
            self._fout.flush()

    def close(self):
        if self._str is not None:
            self._fout.close()


class DeferredFloat:
    """A python-float stand-in for a scalar that still lives on the device: the value is copied to the host (one
    synchronisation) the first time it is USED as a number.  SegmentationLoss returns the accuracy this way, so a
    training step that does not look at it (or looks at it after the backward pass has been enqueued) does not stall
    the host between forward and backward (lazy D2H)."""
    __slots__ = ('_t', '_v')

    def __init__(self, tensor):
        self._t, self._v = tensor, None

    def __float__(self):
        if self._v is None:
            self._v = float(self._t)
            self._t = None
        return self._v

    def __repr__(self): return repr(float(self))
    def __format__(self, spec): return format(float(self), spec)
    def __bool__(self): return bool(float(self))
    def __int__(self): return int(float(self))
    def __neg__(self): return -float(self)
    def __abs__(self): return abs(float(self))
    def __add__(self, o): return float(self) + o
    def __radd__(self, o): return o + float(self)
    def __sub__(self, o): return float(self) - o
    def __rsub__(self, o): return o - float(self)
    def __mul__(self, o): return float(self) * o
    def __rmul__(self, o): return o * float(self)
    def __truediv__(self, o): return float(self) / o
    def __rtruediv__(self, o): return o / float(self)
    def __eq__(self, o): return float(self) == o
    def __lt__(self, o): return float(self) < o
    def __le__(self, o): return float(self) <= o
    def __gt__(self, o): return float(self) > o
    def __ge__(self, o): return float(self) >= o
    __hash__ = None


def round_decimals(val, digits):
    factor = float(10 ** digits)
    return int(val * factor + 0.5) / factor


# ---------------------------------------------------------------------------------------------------------------
# Inference metrics: the per-event quantities of reference uresnet/utils.py:78-200 (sparse) and :382-486
# (dense) that do not need the Michel-electron particle records / DBSCAN (those stay out of scope).  Same result keys,
# shapes and conventions, written with bincount/fancy indexing instead of per-class Python loops.
import numpy as np


def _confusion(labels, predictions, num_classes, weights=None):
    """[true class][predicted class] counts (or weight sums) of flat integer arrays"""
    flat = labels.astype(np.int64) * num_classes + predictions.astype(np.int64)
    m = np.bincount(flat, weights=weights, minlength=num_classes * num_classes)
    return m.reshape(num_classes, num_classes)


def _border_distance_histogram(coords, size):
    """histogram (50 unit bins from 0) of the distance of every voxel to the nearest face of the volume"""
    d = np.minimum(coords, size - coords).min(axis=1)
    return np.histogram(d, bins=np.linspace(0, 50, 51))[0]


def _per_class(labels, predictions, correct_softmax, num_classes):
    counts = np.bincount(labels, minlength=num_classes)[:num_classes].astype(np.float64)
    hits = np.bincount(labels[labels == predictions], minlength=num_classes)[:num_classes].astype(np.float64)
    soft = np.bincount(labels, weights=correct_softmax, minlength=num_classes)[:num_classes]
    with np.errstate(invalid='ignore', divide='ignore'):
        return hits / counts, soft / counts, counts.astype(np.int64)    # empty class -> nan, like the reference


def compute_metrics_sparse(data_v, label_v, softmax_v, idx_v=None, N=192, particles=None):
    """data_v[i] (n, d+2) rows [coords.., batch id, energy]; label_v[i] (n, 1); softmax_v[i] (n, num_classes).
    Returns (res, []) like the reference; `particles` (Michel analysis) is not supported."""
    if particles is not None:
        raise NotImplementedError('Michel-electron analysis (particle records, DBSCAN) is out of scope')
    assert len(data_v) == len(label_v) == len(softmax_v)
    keys = ('acc', 'correct_softmax', 'id', 'nonzero_pixels', 'class_acc', 'class_pixel', 'class_mean_softmax',
            'confusion_matrix', 'energy_confusion_matrix', 'loss_seg', 'misclassified_pixels', 'distances')
    res = {k: [] for k in keys}
    for data, label, softmax in zip(data_v, label_v, softmax_v):
        data = np.asarray(data); softmax = np.asarray(softmax)
        label = np.asarray(label).reshape(-1).astype(np.int64)
        nc = softmax.shape[1]
        for batch_id in np.unique(data[:, -2]):
            sel = data[:, -2] == batch_id
            ev, sm, lab = data[sel], softmax[sel], label[sel]
            pred = sm.argmax(axis=1)
            rows = np.arange(len(lab))
            p_true, p_pred = sm[rows, lab], sm[rows, pred]
            res['acc'].append(float((pred == lab).mean()))
            norm = sm.sum(axis=1)
            res['loss_seg'].append(float(-np.log(np.clip(p_true / norm, 1e-15, 1.0)).mean()))   # multi-class log loss
            res['correct_softmax'].append(float(p_true.mean()))
            res['id'].append(batch_id)
            res['nonzero_pixels'].append(int(len(lab)))
            wrong = pred != lab
            res['misclassified_pixels'].append(np.column_stack([ev[wrong, :-2], p_true[wrong], p_pred[wrong], pred[wrong],
                                                                ev[wrong, -1], lab[wrong]]))
            res['distances'].append(_border_distance_histogram(ev[:, :-2], N))
            class_acc, class_soft, class_pix = _per_class(lab, pred, p_true, nc)
            res['class_acc'].append(list(class_acc))
            res['class_mean_softmax'].append(list(class_soft))
            res['class_pixel'].append(class_pix)
            res['confusion_matrix'].append(_confusion(lab, pred, nc).astype(np.int32))
            res['energy_confusion_matrix'].append(_confusion(lab, pred, nc, weights=ev[:, -1]).astype(np.float32))
    return res, []


def compute_metrics_dense(data_v, label_v, softmax_v, idx_v=None):
    """data_v[i] (1, [D,] H, W); label_v[i] same shape; softmax_v[i] (num_classes, [D,] H, W).  Only non-zero voxels
    (data > 1e-6) count; the last class is background and is left out of the class statistics."""
    assert len(data_v) == len(label_v) == len(softmax_v)
    keys = ('acc', 'correct_softmax', 'id', 'nonzero_pixels', 'class_acc', 'class_pixel', 'class_mean_softmax',
            'confusion_matrix', 'energy_confusion_matrix', 'misclassified_pixels', 'distances')
    res = {k: [] for k in keys}
    for i, (data, label, softmax) in enumerate(zip(data_v, label_v, softmax_v)):
        data = np.asarray(data); softmax = np.asarray(softmax); label = np.asarray(label)
        nc = softmax.shape[0]
        nz = data[0] > 0.000001
        coords = np.argwhere(nz)                                  # (n, dim)
        lab = label[0][nz].astype(np.int64)
        sm = softmax[:, nz].T                                     # (n, nc)
        energy = data[0][nz]
        pred = sm.argmax(axis=1)
        rows = np.arange(len(lab))
        p_true, p_pred = sm[rows, lab], sm[rows, pred]
        res['acc'].append(float((pred == lab).mean()))
        res['correct_softmax'].append(float(p_true.mean()))
        res['id'].append(i)
        res['nonzero_pixels'].append(int(nz.sum()))
        wrong = pred != lab
        lead = np.zeros((int(wrong.sum()), 1), dtype=coords.dtype)   # the channel axis of the (1, ...) image
        res['misclassified_pixels'].append(np.column_stack([lead, coords[wrong], p_true[wrong], p_pred[wrong], pred[wrong],
                                                            energy[wrong], lab[wrong]]))
        res['distances'].append(_border_distance_histogram(coords, data.shape[-1]))
        fg = lab < nc - 1                                          # class statistics over the foreground classes only
        class_acc, class_soft, _ = _per_class(lab[fg], pred[fg], p_true[fg], nc - 1)
        all_counts = np.bincount(label.reshape(-1).astype(np.int64), minlength=nc)
        res['class_acc'].append(list(class_acc))
        res['class_mean_softmax'].append(list(class_soft))
        res['class_pixel'].append(all_counts[:nc - 1])           # counted over the whole image, like the reference
        keep = fg & (pred < nc - 1)
        res['confusion_matrix'].append(_confusion(lab[keep], pred[keep], nc - 1).astype(np.int32))
        res['energy_confusion_matrix'].append(_confusion(lab[keep], pred[keep], nc - 1, weights=energy[keep]).astype(np.float32))
    return res
