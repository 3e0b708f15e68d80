"""Synthetic LArTPC-like event generator.

Stands in for the reference's LArCV readers (reference
uresnet/iotools/iotools_sparse.py:138-160 builds the same blob layout from ROOT
files, which cannot be read here: larcv/ROOT are absent).  The blob layout is
the reference's: one float array per GPU of shape (N, d+2) with columns
[x, y, z, batch_id, value] (iotools_sparse.py:141,152-160) and a label array of
shape (N, 1).

Pure numpy; deterministic for a given seed.
"""
import numpy as np


def _unique_first(vox):
    """Rows of `vox` de-duplicated, keeping generation order."""
    _, idx = np.unique(vox, axis=0, return_index=True)
    return vox[np.sort(idx)]


def _line(p0, direction, length, step=0.5):
    t = np.arange(0.0, length, step)
    return p0[None, :] + t[:, None] * direction[None, :]


def _isotropic(rng):
    v = rng.normal(size=3)
    return v / np.linalg.norm(v)


def generate_event(seed, spatial_size=512, target=50000):
    """One event: (coords int32 (target,3), value float32 (target,), label int32 (target,)).

    Tracks (label 1 MIP if long else 0 HIP), showers (label 2) and short stubs
    (labels 3/4), voxelised by rint, clipped to the volume, globally
    de-duplicated first-seen, truncated at exactly `target` voxels.
    """
    S = int(spatial_size)
    rng = np.random.default_rng(seed)
    seen = set()
    out_c, out_l = [], []
    total = 0
    # margins scale with the volume so small test volumes work too
    m1 = min(64, S // 8)
    m2 = min(96, S // 6)
    scale = min(1.0, S / 512.0)
    while total < target:
        u = rng.uniform()
        if u < 0.6:
            p0 = rng.uniform(m1, S - m1, size=3)
            d = _isotropic(rng)
            L = rng.uniform(40, 400) * scale
            pts = _line(p0, d, L)
            lab = 1 if L > 120 * scale else 0
        elif u < 0.9:
            p0 = rng.uniform(m2, S - m2, size=3)
            axis = _isotropic(rng)
            segs = []
            for _ in range(60):
                s0 = p0 + axis * rng.uniform(0, 150) * scale + rng.normal(0, 6 * scale, size=3)
                d = axis + 0.35 * rng.normal(size=3)
                d /= np.linalg.norm(d)
                segs.append(_line(s0, d, rng.uniform(4, 30)))
            pts = np.concatenate(segs, axis=0)
            lab = 2
        else:
            p0 = rng.uniform(m1, S - m1, size=3)
            d = _isotropic(rng)
            pts = _line(p0, d, rng.uniform(5, 25))
            lab = 3 if rng.uniform() < 0.5 else 4
        vox = np.clip(np.rint(pts), 0, S - 1).astype(np.int64)
        vox = _unique_first(vox)
        keys = (vox[:, 0] * S + vox[:, 1]) * S + vox[:, 2]
        keep = np.fromiter((k not in seen for k in keys.tolist()), dtype=bool, count=len(keys))
        vox = vox[keep]
        seen.update(keys[keep].tolist())
        if len(vox) == 0:
            continue
        out_c.append(vox)
        out_l.append(np.full(len(vox), lab, dtype=np.int32))
        total += len(vox)
    coords = np.concatenate(out_c, axis=0)[:target].astype(np.int32)
    label = np.concatenate(out_l, axis=0)[:target]
    value = np.exp(rng.normal(-1.0, 0.5, size=target)).astype(np.float32)
    return coords, value, label


def make_sparse_blob(seeds, spatial_size=512, target=50000, compute_weight=False):
    """Concatenate events into one per-GPU point cloud, batch ids 0..len(seeds)-1.

    Returns dict(data=(N,5) float32 [x,y,z,batch,value], label=(N,1) float32,
    weight=(N,1) float32 or None).  Weight rule mirrors the reference's
    class balancing (iotools_sparse.py:311-318): N / (n_classes * count_c).
    """
    datas, labels, weights = [], [], []
    for b, seed in enumerate(seeds):
        c, v, l = generate_event(seed, spatial_size, target)
        d = np.concatenate([c.astype(np.float32),
                            np.full((len(c), 1), b, np.float32),
                            v[:, None]], axis=1)
        datas.append(d)
        labels.append(l[:, None].astype(np.float32))
        if compute_weight:
            cls, cnt = np.unique(l, return_counts=True)
            w = np.zeros(len(l), np.float32)
            for ci, ni in zip(cls, cnt):
                w[l == ci] = float(len(l)) / (len(cls) * ni)
            weights.append(w[:, None])
    blob = dict(data=np.concatenate(datas, 0), label=np.concatenate(labels, 0), weight=None)
    if compute_weight:
        blob['weight'] = np.concatenate(weights, 0)
    return blob


def make_dense_blob(seeds, spatial_size=128, dim=3, num_class=5, fill=None):
    """Dense (B,1,[D,]H,W) image + label, background label = num_class-1 at empty
    voxels (mirrors EmptyVoxelValue, reference iotools_dense.py:30-31).

    3-D: rasterised from generate_event at S=spatial_size.  2-D: ~10 % pixels
    non-zero uniform(0,1), labels uniform 0..nc-1.
    """
    S = int(spatial_size)
    shape = (S,) * dim
    data = np.zeros((len(seeds), 1) + shape, np.float32)
    label = np.full((len(seeds), 1) + shape, num_class - 1, np.float32)
    for b, seed in enumerate(seeds):
        if dim == 3:
            n = fill if fill is not None else max(64, int(50000 * (S / 512.0) ** 2))
            c, v, l = generate_event(seed, S, n)
            data[b, 0, c[:, 0], c[:, 1], c[:, 2]] = v
            label[b, 0, c[:, 0], c[:, 1], c[:, 2]] = np.minimum(l, num_class - 1)
        else:
            rng = np.random.default_rng(seed)
            mask = rng.uniform(size=shape) < 0.1
            data[b, 0][mask] = rng.uniform(size=int(mask.sum())).astype(np.float32) + 1e-3
            label[b, 0][mask] = rng.integers(0, num_class, size=int(mask.sum()))
    return dict(data=data, label=label, weight=None)
