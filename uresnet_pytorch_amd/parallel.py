"""Data parallelism, one process per GPU (replaces the reference's single-process
GraphDataParallel, reference uresnet/ops.py:8-60 and uresnet/trainval.py:152-154).

Events of a batch are independent through the whole network, so they are sharded
over ranks (whole events, balanced by active-voxel count); BatchNorm statistics stay
per rank like the reference's per-replica statistics.  The only data-path collective
is ONE all-reduce(SUM) of a flat fp32 gradient buffer per optimizer step (RCCL over
xGMI with backend "nccl"; "gloo" in CPU tests).  SUM, not MEAN: the reference loss is
the sum over all events on all GPUs (reference uresnet_sparse.py:72-74, trainval.py:25).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = os.environ.get('URN_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            local_rank = local_rank % max(1, torch.cuda.device_count())   # rehearsal: several ranks on one GPU (gloo only)
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_events(sizes, world):
    """Greedy LPT assignment of events (by active-voxel count) to ranks.
    Returns a list (per rank) of event indices; deterministic."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += int(sizes[i])
    for r in range(world):
        out[r].sort()
    return out


class FlatGradients:
    """Keeps every parameter's .grad as a view into one flat fp32 buffer so that a step
    needs one memset and one all-reduce."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device('cpu')
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, async_op=False):
        """SUM over ranks (no-op when not distributed)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        return None


class OverlappedAllReduce:
    """The gradient all-reduce of a data-parallel step in TWO pieces, the first one overlapped with the backward pass
    (replaces the reference's DataParallel gradient reduce, uresnet/trainval.py:21-30,152-154).

    The sparse executor finishes the flat gradient buffer from the back: [decoder + bottom level + head] is complete when the
    bottom level's backward has run (urn_net_suffix_offset), the encoder prefix only at the end.  arm() hooks the executor:
    at that point the suffix is all-reduced (SUM) asynchronously BEHIND THE EXECUTOR'S SIDE STREAM (which carries the weight
    gradients and has been ordered behind the caller's stream), while the encoder half of the backward pass -- about half of
    its time -- still runs; finish() reduces the prefix and joins both before the optimizer reads the gradients.
    Only for ONE executor backward per step (gradient accumulation over several forwards completes the suffix in the last
    backward only: arm(expected=k) fires at the k-th).  Without an executor, or when the hook never fires, finish() is the
    plain single all-reduce."""

    def __init__(self, flat_grads, force=False):
        self._fg = flat_grads
        self._force = force           # tools / tests: run the split path at world size 1 as well
        self._works, self._offset, self._left, self._ex = [], None, 0, None

    def active(self):
        return self._force or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)

    def _reduce(self, t):
        if dist.is_available() and dist.is_initialized():
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
        return None

    def arm(self, model, expected=1):
        ex = getattr(model, '_executor', None)
        self._works, self._offset, self._left, self._ex = [], None, expected, ex
        if ex is None or not self.active():
            return False
        ex.suffix_hook = self._on_suffix
        return True

    def _on_suffix(self, offset, side_stream_ptr):
        self._left -= 1
        if self._left > 0:
            return
        flat = self._fg.flat
        if side_stream_ptr:
            # behind the side stream's tail: the weight gradients of the suffix are queued there
            with torch.cuda.stream(torch.cuda.ExternalStream(side_stream_ptr, device=flat.device)):
                w = self._reduce(flat[offset:])
        else:
            w = self._reduce(flat[offset:])
        self._offset = offset
        if w is not None:
            self._works.append(w)

    def finish(self):
        """after backward(): reduce what is left and wait; returns how many collectives ran"""
        if self._ex is not None:
            self._ex.suffix_hook = None
        if not self.active():
            return 0
        flat = self._fg.flat
        w = self._reduce(flat if self._offset is None else flat[:self._offset])
        if w is not None:
            self._works.append(w)
        n = len(self._works)
        for w in self._works:
            w.wait()
        self._works = []
        return n


def broadcast_parameters(module, src=0):
    """One broadcast at initialize()/checkpoint load; replicas then stay in sync because every
    rank applies the identical summed gradient."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)


def all_reduce_scalars(values, device):
    """Sum a few python floats over ranks (loss / accuracy reporting)."""
    t = torch.tensor(values, dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam (reference uresnet/trainval.py:37) for GPU parameters whose gradients live in a
    FlatGradients buffer: both moments are one flat buffer each and a step is one urn_adam_flat launch per
    contiguous parameter segment (the executor's flat trunk parameters are one segment) instead of a
    multi-tensor launch over every small tensor.  state_dict()/load_state_dict() keep torch.optim.Adam's layout
    ({'step', 'exp_avg', 'exp_avg_sq'} per parameter), so checkpoints interchange with the reference's optimizer."""

    def __init__(self, flat_grads, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        from . import lib
        self._lib = lib
        self._fg = flat_grads
        params = flat_grads.params
        if not params or not params[0].is_cuda:
            raise RuntimeError('FlatAdam needs GPU parameters (use torch.optim.Adam on the CPU)')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        total = flat_grads.flat.numel()
        self._m = torch.zeros(total, dtype=torch.float32, device=params[0].device)
        self._v = torch.zeros_like(self._m)
        self._step = 0
        self._bind_state()
        self._segments, self._seg_key = [], None

    def _segment(self):
        """maximal runs of parameters that are adjacent in memory (their gradients are adjacent by construction);
        recomputed when the storage moved (the executor re-homes the trunk parameters at the first forward)"""
        params = self._fg.params
        key = (params[0].data_ptr(), params[len(params) // 2].data_ptr(), params[-1].data_ptr())
        if key == self._seg_key:
            return self._segments
        segs = []
        off = 0
        for p in params:
            n = p.numel()
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError('FlatAdam: fp32 contiguous parameters only')
            last = segs[-1] if segs else None
            if last is not None and last[0] + 4 * last[2] == p.data_ptr() and last[1] + last[2] == off:
                last[2] += n
            else:
                segs.append([p.data_ptr(), off, n])
            off += n
        self._segments, self._seg_key = segs, key
        return segs

    def _bind_state(self):
        off = 0
        for p in self._fg.params:
            n = p.numel()
            self.state[p] = {'step': torch.tensor(float(self._step)),
                             'exp_avg': self._m[off:off + n].view_as(p), 'exp_avg_sq': self._v[off:off + n].view_as(p)}
            off += n

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        grp = self.param_groups[0]
        self._step += 1
        L = self._lib.load()
        st = self._lib.stream()
        g0, m0, v0 = self._fg.flat.data_ptr(), self._m.data_ptr(), self._v.data_ptr()
        for ptr, off, n in self._segment():
            self._lib.check(L.urn_adam_flat(ptr, g0 + 4 * off, m0 + 4 * off, v0 + 4 * off, n, float(grp['lr']),
                                            float(grp['betas'][0]), float(grp['betas'][1]), float(grp['eps']),
                                            float(grp['weight_decay']), self._step, st))
        return loss

    def state_dict(self):
        for p in self._fg.params:
            self.state[p]['step'] = torch.tensor(float(self._step))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)      # replaces the state tensors: copy them back into the flat buffers
        off = 0
        steps = []
        for p in self._fg.params:
            n = p.numel()
            s = self.state.get(p, {})
            if 'exp_avg' in s:
                self._m[off:off + n].copy_(s['exp_avg'].reshape(-1))
                self._v[off:off + n].copy_(s['exp_avg_sq'].reshape(-1))
                steps.append(int(float(s['step'])))
            off += n
        self._step = max(steps) if steps else 0
        self._bind_state()
