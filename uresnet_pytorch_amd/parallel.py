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
