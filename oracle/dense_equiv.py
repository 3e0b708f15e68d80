"""Dense-equivalence pin for the sparse oracle (test infrastructure only).

The reference pins nothing for the sparse path (no tests, sparseconvnet absent),
so the CPU restatement in sparse_ref.c is pinned against the mathematical
definition evaluated with torch.nn.functional on small densified volumes
(SURVEY.md section 8c):
  SubM3          == F.conv3d(dense, W, padding=1)          sampled at the active set
  Conv k2 s2     == F.conv3d(dense, W, stride=2)           sampled at the coarse active set
  Deconv k2 s2   == F.conv_transpose3d(dense, W, stride=2) sampled at the fine active set
  BatchNormReLU  == relu(F.batch_norm(rows, training=True, eps=1e-4))
plus a brute-force numpy enumerator for active sets and (offset,in,out) triples,
and a float64 torch re-expression of the whole network (autograd) to check the
oracle's hand-written backward composition.
"""
import numpy as np
import torch
import torch.nn.functional as F


def densify(coords, feats, spatial, nbatch):
    C = feats.shape[1]
    d = torch.zeros(nbatch, C, spatial, spatial, spatial, dtype=torch.float64)
    c = torch.as_tensor(np.asarray(coords), dtype=torch.long)
    d[c[:, 3], :, c[:, 0], c[:, 1], c[:, 2]] = torch.as_tensor(np.asarray(feats), dtype=torch.float64)
    return d


def sample(dense, coords):
    c = torch.as_tensor(np.asarray(coords), dtype=torch.long)
    return dense[c[:, 3], :, c[:, 0], c[:, 1], c[:, 2]]


def subm_weight_dense(W):
    """(27,Cin,Cout) -> (Cout,Cin,3,3,3) with o = (kx*3+ky)*3+kz."""
    W = torch.as_tensor(np.asarray(W), dtype=torch.float64)
    return W.reshape(3, 3, 3, W.shape[1], W.shape[2]).permute(4, 3, 0, 1, 2).contiguous()


def down_weight_dense(W):
    W = torch.as_tensor(np.asarray(W), dtype=torch.float64)
    return W.reshape(2, 2, 2, W.shape[1], W.shape[2]).permute(4, 3, 0, 1, 2).contiguous()


def up_weight_dense(W):
    """(8,Cin,Cout) -> conv_transpose3d weight (Cin,Cout,2,2,2)."""
    W = torch.as_tensor(np.asarray(W), dtype=torch.float64)
    return W.reshape(2, 2, 2, W.shape[1], W.shape[2]).permute(3, 4, 0, 1, 2).contiguous()


def subm_dense(coords, feats, W, spatial, nbatch):
    return sample(F.conv3d(densify(coords, feats, spatial, nbatch), subm_weight_dense(W), padding=1), coords)


def down_dense(fine_coords, feats, W, coarse_coords, spatial, nbatch):
    return sample(F.conv3d(densify(fine_coords, feats, spatial, nbatch), down_weight_dense(W), stride=2),
                  coarse_coords)


def up_dense(coarse_coords, feats, W, fine_coords, spatial_coarse, nbatch):
    return sample(F.conv_transpose3d(densify(coarse_coords, feats, spatial_coarse, nbatch),
                                     up_weight_dense(W), stride=2), fine_coords)


# ---------------------------------------------------- brute-force integers --
def brute_sites(coords):
    """First-occurrence unique of (N,4) int rows: (row2site, site_coords)."""
    seen = {}
    row2site = np.empty(len(coords), np.int32)
    sc = []
    for i, c in enumerate(map(tuple, np.asarray(coords).tolist())):
        if c not in seen:
            seen[c] = len(sc); sc.append(c)
        row2site[i] = seen[c]
    return row2site, np.asarray(sc, np.int32).reshape(-1, 4)


def brute_subm_triples(site_coords, spatial):
    idx = {tuple(c): j for j, c in enumerate(np.asarray(site_coords).tolist())}
    t = []
    for j, (x, y, z, b) in enumerate(np.asarray(site_coords).tolist()):
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    q = (x + dx, y + dy, z + dz, b)
                    if min(q[:3]) < 0 or max(q[:3]) >= spatial:
                        continue
                    i = idx.get(q)
                    if i is not None:
                        t.append((((dx + 1) * 3 + (dy + 1)) * 3 + (dz + 1), i, j))
    t = np.asarray(sorted(t), np.int32).reshape(-1, 3)
    return t


def brute_down(fine_coords):
    fc = np.asarray(fine_coords)
    cc = np.concatenate([fc[:, :3] >> 1, fc[:, 3:4]], axis=1)
    parent, coarse = brute_sites(cc)
    off = ((fc[:, 0] & 1) * 2 + (fc[:, 1] & 1)) * 2 + (fc[:, 2] & 1)
    return coarse, parent, off.astype(np.int32)


# ------------------------------------------- float64 torch network (autograd) --
def _gconv(x, W, nbr):
    nbr = torch.as_tensor(np.asarray(nbr), dtype=torch.long)
    y = 0
    for o in range(nbr.shape[0]):
        m = nbr[o] >= 0
        if not bool(m.any()):
            continue
        g = torch.zeros(nbr.shape[1], x.shape[1], dtype=x.dtype)
        g[m] = x[nbr[o][m]]
        y = y + g @ W[o]
    return y


def _bnrelu(x, g, b, eps):
    return torch.relu(F.batch_norm(x, None, None, g, b, True, 0.0, eps))


def torch_network(params, geo, m, num_strides, reps, eps):
    """float64 re-expression of SparseUResNetOracle.forward on a prebuilt Geometry.
    Returns (logits, leaf-parameter dict) so the caller can run autograd."""
    P = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    planes = [i * m for i in range(1, num_strides + 1)]

    def block(prefix, idx, a, b, x, l):
        p = '%s.%d' % (prefix, idx)
        sc = x @ P[p + '.0.weight'] if a != b else x
        t = _bnrelu(x, P[p + '.1.0.weight'], P[p + '.1.0.bias'], eps)
        t = _gconv(t, P[p + '.1.1.weight'], geo.nbr[l])
        t = _bnrelu(t, P[p + '.1.2.weight'], P[p + '.1.2.bias'], eps)
        t = _gconv(t, P[p + '.1.3.weight'], geo.nbr[l])
        return sc + t

    def U(prefix, l, x):
        pl = planes[l:]
        idx = 0
        for _ in range(reps):
            x = block(prefix, idx, pl[0], pl[0], x, l); idx += 2
        if len(pl) > 1:
            p = '%s.%d.1' % (prefix, idx)
            t = _bnrelu(x, P[p + '.0.weight'], P[p + '.0.bias'], eps)
            t = _gconv(t, P[p + '.1.weight'], geo.chd[l])
            t = U(p + '.2', l + 1, t)
            t = _bnrelu(t, P[p + '.3.weight'], P[p + '.3.bias'], eps)
            t = _gconv(t, P[p + '.4.weight'], geo.up[l])
            x = torch.cat([x, t], dim=1)
            idx += 2
            for i in range(reps):
                x = block(prefix, idx, pl[0] * (2 if i == 0 else 1), pl[0], x, l); idx += 2
        return x

    feats = torch.tensor(geo.feats, dtype=torch.float64, requires_grad=True)
    x = _gconv(feats, P['sparseModel.1.weight'], geo.nbr[0])
    x = U('sparseModel.2', 0, x)
    x = _bnrelu(x, P['sparseModel.3.weight'], P['sparseModel.3.bias'], eps)
    rows = x[torch.as_tensor(geo.row2site, dtype=torch.long)]
    logits = rows @ P['linear.weight'].t() + P['linear.bias']
    return logits, P, feats
