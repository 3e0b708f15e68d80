"""Sparse U-ResNet behind the reference's nn.Module surface.

Mirrors reference uresnet/models/uresnet_sparse.py:7-82 (class names, constructor
signature, forward contract, loss semantics); the `sparseconvnet` calls are served
by uresnet_pytorch_amd.scn, i.e. hand-written gfx950 kernels behind the C ABI.
"""
import torch

from ..utils import DeferredFloat

from .. import lib as _lib
from .. import scn
from .. import sparse_ops as so
from ..trunk import TrunkExecutor


class UResNet(torch.nn.Module):
    def __init__(self, flags):
        super(UResNet, self).__init__()
        self._flags = flags
        dimension = flags.DATA_DIM
        reps = 2          # conv block repetition factor       (reference :13)
        kernel_size = 2   # strided conv filter size           (reference :14)
        m = flags.URESNET_FILTERS
        nPlanes = [i * m for i in range(1, flags.URESNET_NUM_STRIDES + 1)]   # linear widths (reference :16)
        nInputFeatures = 1
        self.sparseModel = scn.Sequential().add(
            scn.InputLayer(dimension, flags.SPATIAL_SIZE, mode=3)).add(
            scn.SubmanifoldConvolution(dimension, nInputFeatures, m, 3, False)).add(
            scn.UNet(dimension, reps, nPlanes, residual_blocks=True, downsample=[kernel_size, 2])).add(
            scn.BatchNormReLU(m)).add(
            scn.OutputLayer(dimension))
        self.sparseModel[0].num_levels = len(nPlanes)   # build every strided level in one integer phase
        self.linear = torch.nn.Linear(m, flags.NUM_CLASS)
        # fast path: the whole sparseModel inside the C++ executor (same kernels, one autograd node);
        # m must be a multiple of 16 (MFMA tiles).  use_executor=False keeps the per-layer path.
        self.use_executor = (dimension == 3 and m % 16 == 0)
        self.executor_flags = 0     # lib URN_NET_UNFUSED (1) / URN_NET_SINGLE_STREAM (2) / URN_NET_SLAB_STATS (4): debug and A/B switches
        self._executor = None
        self.fuse_head = True       # A/B switch: False keeps OutputLayer rows + HeadFunction as separate launches

    def _trunk(self, coords, features):
        if self._executor is None:
            f = self._flags
            self._executor = TrunkExecutor(self.sparseModel, f.URESNET_FILTERS, f.URESNET_NUM_STRIDES, 2, f.NUM_CLASS)
            self._executor.flags = self.executor_flags
        ex = self._executor
        inp = self.sparseModel[0]
        c = coords.to(torch.int32) if coords.dtype != torch.int32 else coords
        # everything that does not need the level counts goes BEFORE their read-back (geo.sync() inside forward):
        # host time after that synchronisation is exposed in the step, host time before it is not
        ex.flatten(c.device, tail=(self.linear.weight, self.linear.bias))
        ex.prepare_weights()      # transposed / fragment-ordered weight copies on the side stream, beside the integer phase
        geo = so.SparseGeometry(c, inp.spatial_size, inp.num_levels, defer_sync=True, counts_hint=getattr(self, '_counts_hint', None))
        self._last_geo = geo
        feats = so.input_features(geo, features)
        # the Linear head inside the executor (last BatchNormReLU + OutputLayer + Linear = one kernel) where it can be
        head = (self.linear.weight, self.linear.bias) if (self.fuse_head and ex.head_ok(self.linear.weight, self.linear.bias)) else None
        ex.prepare(geo, self.training, head)
        return ex.forward(geo, feats, self.training, head), head is not None

    def forward(self, point_cloud):
        """point_cloud: (N, d+2) rows [x, y, z, batch_id, value]; returns [ (N, NUM_CLASS) ]."""
        coords = point_cloud[:, 0:-1].float()
        features = point_cloud[:, -1][:, None].float()
        if coords.is_cuda:
            _lib.set_precision(getattr(self._flags, 'PRECISION', 'fp32'))   # flags -prec: fp32 (default) | bf16 | fp16
        # the executor serves training steps and (with the running BatchNorm statistics) inference without gradients;
        # eval mode WITH autograd, hooks and CPU tensors take the per-layer path
        done = False
        if self.use_executor and coords.is_cuda and (self.training or not torch.is_grad_enabled()):
            x, done = self._trunk(coords, features)
        else:
            x = self.sparseModel((coords, features))
        if done:        # the executor ran the head as well: x are the logits
            pass
        elif x.is_cuda:   # Linear on the HIP head kernel (same parameters: self.linear.weight / .bias)
            x = so.HeadFunction.apply(x, self.linear.weight, self.linear.bias)
        else:
            x = self.linear(x)
        return [x]


class SegmentationLoss(torch.nn.modules.loss._Loss):
    """Per-GPU-entry, per-event mean voxel cross-entropy, SUMMED over events
    (reference uresnet_sparse.py:46-82).  Returns (loss tensor, accuracy sum)."""

    def __init__(self, flags, reduction='sum'):
        super(SegmentationLoss, self).__init__(reduction=reduction)
        self._flags = flags
        self.cross_entropy = torch.nn.CrossEntropyLoss(reduction='none')

    def forward(self, segmentation, data, label, weight):
        assert len(segmentation) == len(data)
        assert len(data) == len(label)
        if weight is not None:
            assert len(data) == len(weight)
        total_loss = 0
        total_acc = 0
        for i in range(len(segmentation)):
            if segmentation[i].is_cuda:
                # one fused pass on the device: per-event sums, no host sync per event
                d = data[i] if (data[i].dtype == torch.float32 and data[i].is_contiguous()) else data[i].float().contiguous()
                lab = label[i] if label[i].dtype == torch.float32 else label[i].float()
                w = None if weight is None else weight[i]
                loss_i, out = so.SegmentationCEFunction.apply(segmentation[i], d, lab, w)
                # (the first entry is taken as it is: "0 + tensor" would be one more launch on the device)
                total_loss = loss_i if i == 0 else total_loss + loss_i
                total_acc = out[1] if i == 0 else total_acc + out[1]
                continue
            batch_ids = data[i][:, -2]
            ids, inv = torch.unique(batch_ids, return_inverse=True)
            nev = ids.numel()
            event_label = torch.squeeze(label[i], dim=-1).long()
            loss_seg = self.cross_entropy(segmentation[i], event_label)
            if weight is not None:
                loss_seg = loss_seg * torch.squeeze(weight[i], dim=-1).float()
            cnt = torch.zeros(nev, device=loss_seg.device, dtype=loss_seg.dtype).index_add_(
                0, inv, torch.ones_like(loss_seg))
            per_event = torch.zeros(nev, device=loss_seg.device, dtype=loss_seg.dtype).index_add_(0, inv, loss_seg)
            total_loss = total_loss + (per_event / cnt).sum()
            correct = (torch.argmax(segmentation[i], dim=-1) == event_label).to(loss_seg.dtype)
            acc = torch.zeros(nev, device=loss_seg.device, dtype=loss_seg.dtype).index_add_(0, inv, correct)
            total_acc = total_acc + (acc / cnt).sum()
        if torch.is_tensor(total_acc) and total_acc.is_cuda:
            return total_loss, DeferredFloat(total_acc)   # a float when used; no host sync between forward and backward
        return total_loss, float(total_acc)
