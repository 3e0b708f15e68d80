"""CPU baseline for bench.py's `cpu_baseline` leg: an SCN-style fp32 restatement of one training step's forward +
backward of the sparse U-ResNet (reference uresnet/models/uresnet_sparse.py:19-25 + the published body of scn.UNet,
SURVEY.md App. A) on the host cores.

TEST / MEASUREMENT INFRASTRUCTURE ONLY: nothing under uresnet_pytorch_amd/ imports this module.

Two forms of the convolutions (`kernel=`): 'omp' (default when liboracle_cpu.so is built) -- oracle/cpu_fast.c, the gather
convolution as ONE OpenMP loop over tiles of output rows, offset by offset inside a tile, fp32, all host cores, private
weight-gradient copies reduced at the end; 'torch' -- the per-offset index_select -> mm -> index_add_ form below.

How SparseConvNet computes on a CPU (the reference's own CPU path, which cannot run here: the library is absent): the
rulebook is built on the host, and every convolution is, per filter offset, a gather of the input rows of that offset's
rules, one dense sgemm with W[offset], and a scatter-add into the output rows; BatchNorm is a pass over the (N, C) row
matrix.  This module does exactly that with torch CPU ops (index_select -> mm -> index_add_, F.batch_norm), fp32,
all host threads (torch.get_num_threads()), autograd for the backward pass -- where the numpy/C oracle (sparse_ref.c)
accumulates in fp64 and is written for checking, not for speed.  The rulebook comes from the oracle's Geometry (the
integer phase is not what this baseline times: only forward + backward, like the metric).

Its results are checked against the oracle in tests/test_oracle_sparse.py (logits and gradients, 1e-4)."""
import ctypes
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import sparse_oracle as orc

BN_EPS = 1e-4
_FAST = None


def fast_lib():
    """oracle/liboracle_cpu.so (cpu_fast.c), or None when it has not been built"""
    global _FAST
    if _FAST is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'liboracle_cpu.so')
        if not os.path.exists(path):
            _FAST = False
        else:
            L = ctypes.CDLL(path)
            vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
            L.cpuf_gconv.argtypes = [vp, vp, vp, i64, ci, ci, i64, ci, ci, ci, vp]
            L.cpuf_gconv_dw.argtypes = [vp, vp, vp, i64, ci, i64, ci, ci, vp]
            L.cpuf_set_threads.argtypes = [ci]
            L.cpuf_max_threads.restype = ci
            _FAST = L
    return _FAST or None


class _FastConv(torch.autograd.Function):
    """y = gather-conv(x, W) on cpu_fast.c; tbl (K, n_out) int32 forward table, inv (K, n_in) its inverse, flip_b as in
    uresnet_pytorch_amd.sparse_ops.GConvFunction (a submanifold table is its own inverse with the offsets mirrored)"""

    @staticmethod
    def forward(ctx, x, W, tbl, inv, flip_b, n_out):
        L = fast_lib()
        x = x.contiguous(); W = W.contiguous()
        K, cin, cout = W.shape
        y = torch.empty((n_out, cout), dtype=torch.float32)
        L.cpuf_gconv(x.data_ptr(), W.data_ptr(), tbl.data_ptr(), tbl.shape[1], K, 0, n_out, cin, cout, 0, y.data_ptr())
        ctx.save_for_backward(x, W)
        ctx.t = (tbl, inv, flip_b, n_out)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        tbl, inv, flip_b, n_out = ctx.t
        L = fast_lib()
        K, cin, cout = W.shape
        dy = dy.contiguous()
        n_in = x.shape[0]
        dx = torch.empty((n_in, cin), dtype=torch.float32)
        # dx[i] = sum_o dy[inv[o][i]] @ W[o]^T: the same loop, gathering cout channels, W[o] read as (cout, cin)^T
        L.cpuf_gconv(dy.data_ptr(), W.data_ptr(), inv.data_ptr(), inv.shape[1], K, int(flip_b), n_in, cout, cin, 1, dx.data_ptr())
        dw = torch.zeros_like(W)
        L.cpuf_gconv_dw(x.data_ptr(), dy.data_ptr(), tbl.data_ptr(), tbl.shape[1], K, n_out, cin, cout, dw.data_ptr())
        return dx, dw, None, None, None, None


class CpuPort:
    def __init__(self, params, m, num_strides, num_class, spatial, reps=2, kernel=None):
        self.kernel = kernel or ('omp' if fast_lib() is not None else 'torch')
        self.P = {k: torch.from_numpy(np.ascontiguousarray(v)).clone().requires_grad_(True) for k, v in params.items()}
        self.m, self.L, self.nc, self.spatial, self.reps = m, num_strides, num_class, spatial, reps
        self.planes = [i * m for i in range(1, num_strides + 1)]

    # -- geometry: per table, the rules of every offset as (input rows, output rows) index tensors --------------------
    @staticmethod
    def _rules(tbl):
        out = []
        for o in range(tbl.shape[0]):
            j = np.nonzero(tbl[o] >= 0)[0]
            out.append((torch.from_numpy(tbl[o, j].astype(np.int64)), torch.from_numpy(j.astype(np.int64))))
        return out

    def set_geometry(self, point_cloud):
        pc = np.asarray(point_cloud)
        coords = pc[:, :4].astype(np.float32).astype(np.int64).astype(np.int32)
        feats = pc[:, 4:5].astype(np.float32)
        g = orc.Geometry(coords, feats, self.spatial, self.L, mode=3)
        self.n = g.n
        if self.kernel == 'omp':
            # dense tables + their inverses (submanifold: itself mirrored; strided: chd and up are each other's inverse)
            T = lambda t: torch.from_numpy(np.ascontiguousarray(t.astype(np.int32)))
            self.nbr = [(T(t), T(t), 1) for t in g.nbr]
            self.chd = [(T(c), T(u), 0) for c, u in zip(g.chd, g.up)]
            self.up = [(T(u), T(c), 0) for c, u in zip(g.chd, g.up)]
        else:
            self.nbr = [self._rules(t) for t in g.nbr]
            self.chd = [self._rules(t) for t in g.chd]
            self.up = [self._rules(t) for t in g.up]
        self.feats = torch.from_numpy(g.feats)
        self.row2site = torch.from_numpy(g.row2site.astype(np.int64))

    # -- operators -----------------------------------------------------------------------------------------------
    def _conv(self, x, W, rules, n_out):
        if self.kernel == 'omp':
            tbl, inv, flip_b = rules
            return _FastConv.apply(x, W, tbl, inv, flip_b, n_out)
        y = torch.zeros((n_out, W.shape[2]), dtype=torch.float32)
        for o, (i_in, i_out) in enumerate(rules):
            if i_in.numel():
                y = y.index_add(0, i_out, x.index_select(0, i_in) @ W[o])      # gather -> sgemm -> scatter-add
        return y

    def _bn(self, prefix, x):
        return F.relu(F.batch_norm(x, None, None, self.P[prefix + '.weight'], self.P[prefix + '.bias'], True, 0.0, BN_EPS))

    def _block(self, prefix, idx, a, b, x, l):
        p = '%s.%d' % (prefix, idx)
        sc = x @ self.P[p + '.0.weight'] if a != b else x
        t = self._conv(self._bn(p + '.1.0', x), self.P[p + '.1.1.weight'], self.nbr[l], self.n[l])
        t = self._conv(self._bn(p + '.1.2', t), self.P[p + '.1.3.weight'], self.nbr[l], self.n[l])
        return sc + t

    def _U(self, prefix, l, x):
        pl = self.planes[l:]
        idx = 0
        for _ in range(self.reps):
            x = self._block(prefix, idx, pl[0], pl[0], x, l); idx += 2
        if len(pl) > 1:
            p = '%s.%d.1' % (prefix, idx)
            t = self._conv(self._bn(p + '.0', x), self.P[p + '.1.weight'], self.chd[l], self.n[l + 1])
            t = self._U(p + '.2', l + 1, t)
            t = self._conv(self._bn(p + '.3', t), self.P[p + '.4.weight'], self.up[l], self.n[l])
            x = torch.cat([x, t], dim=1)
            idx += 2
            for i in range(self.reps):
                x = self._block(prefix, idx, pl[0] * (2 if i == 0 else 1), pl[0], x, l); idx += 2
        return x

    def forward(self):
        x = self._conv(self.feats, self.P['sparseModel.1.weight'], self.nbr[0], self.n[0])
        x = self._U('sparseModel.2', 0, x)
        x = self._bn('sparseModel.3', x)
        rows = x.index_select(0, self.row2site)
        return rows @ self.P['linear.weight'].t() + self.P['linear.bias']

    def step(self, data, label):
        """forward + loss + backward (gradients land in self.P[*].grad); returns (logits, loss)"""
        for p in self.P.values():
            p.grad = None
        logits = self.forward()
        bid = torch.from_numpy(np.asarray(data)[:, -2].astype(np.int64))
        lab = torch.from_numpy(np.asarray(label).reshape(-1).astype(np.int64))
        ce = F.cross_entropy(logits, lab, reduction='none')
        loss = sum(ce[bid == b].mean() for b in torch.unique(bid))      # sum over events of the per-event mean
        loss.backward()
        return logits.detach(), float(loss.detach())


def time_step(params, m, num_strides, num_class, spatial, data, label, warmup=2, repeats=5, threads=None):
    """median seconds of `repeats` forward+backward passes after `warmup` untimed ones; (median, all times, threads).
    threads=None keeps torch's thread count; 'auto' first times one pass at 8 / 16 / 32 / 64 threads (as far as the host
    has them) and measures at the fastest: the many small gathers and sgemms of a 50k-voxel event do not scale to 128
    threads (measured on the GPU box: 19.9 s per step with all 128 threads against well under a second with 8)."""
    port = CpuPort(params, m, num_strides, num_class, spatial)
    port.set_geometry(data)
    before = torch.get_num_threads()
    sweep = {}
    L = fast_lib() if port.kernel == 'omp' else None
    omp_before = L.cpuf_max_threads() if L is not None else 0

    def set_threads(c):
        torch.set_num_threads(min(c, 64))       # (BatchNorm / head / loss passes: torch ops on small matrices)
        if L is not None:
            L.cpuf_set_threads(c)
    if threads == 'auto':
        ncpu = os.cpu_count() or before
        # (the GPU boxes hand a job a SHARE of the host -- 16 of 256 CPUs for one GPU -- and more threads than that only
        #  spin against each other: 128 threads took 3.7 s per step, 256 threads 107 s, against 0.14 s at 16.  The sweep
        #  therefore stops as soon as a count is 1.5x slower than the best so far.)
        cands = (8, 16, 32, 64, 128, 256) if L is not None else (8, 16, 32, 64)
        for c in [c for c in cands if c <= ncpu] or [before]:
            if sweep and min(sweep.values()) * 1.5 < sweep[max(sweep)]:
                break
            set_threads(c)
            port.step(data, label)
            t0 = time.perf_counter()
            port.step(data, label)
            sweep[c] = time.perf_counter() - t0
        threads = min(sweep, key=sweep.get)
    if threads:
        set_threads(int(threads))
    try:
        for _ in range(warmup):
            port.step(data, label)
        ts = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            port.step(data, label)
            ts.append(time.perf_counter() - t0)
        used = int(threads) if (threads and L is not None) else int(torch.get_num_threads())
    finally:
        torch.set_num_threads(before)
        if L is not None:
            L.cpuf_set_threads(omp_before)
    time_step.last_sweep = sweep
    time_step.last_kernel = port.kernel
    return float(np.median(ts)), ts, used
