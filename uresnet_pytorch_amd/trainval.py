"""Trainer with the reference's API (reference uresnet/trainval.py:15-198):
trainval(flags).initialize() / train_step(data_blob, epoch, batch_size) / forward(...) /
save_state(iteration), timing dicts tspent / tspent_sum, result keys
segmentation / softmax / accuracy / loss_seg.

Differences, all on purpose:
  * one process per GPU instead of single-process DataParallel: each rank runs the events
    of data_blob[...][substep][local gpu slot]; gradients are summed over ranks with ONE
    RCCL all-reduce of a flat buffer (uresnet_pytorch_amd.parallel);
  * tensors stay batched without a GPU (the reference's no-CUDA branch drops the batch
    dimension, trainval.py:95-102) -- only the dense model can run on CPU, the sparse HIP
    path raises without a GPU;
  * logits/softmax are moved to the host once per forward, as the reference does (:127-128).
"""
import os
import sys
import time

import numpy as np
import torch

from . import models
from . import parallel


class trainval(object):
    def __init__(self, flags):
        self._flags = flags
        self.tspent = {}
        self.tspent_sum = {}

    # -- reference trainval.py:21-30
    def backward(self):
        total_loss = 0.0
        for loss in self._loss:
            total_loss = total_loss + loss
        total_loss = total_loss / max(len(self._loss), 1)
        self._loss = []
        self._grads.zero()
        # a rank that owned no entry in any sub-step (fewer events than ranks: shard_events([5], 3) -> [[0], [], []]) has no
        # graph to differentiate: it contributes ZERO gradients to the sum and still takes part in the collective and the
        # optimizer step, so that the replicas stay identical
        has_graph = torch.is_tensor(total_loss) and total_loss.requires_grad
        # loss scaling (flags -ls, a power of two): the backward pass is linear in the incoming gradient, so scaling the
        # loss scales every gradient operand -- with -prec fp16 the unscaled ones (1e-6 .. 1e-8 at cfg5 size) fall below
        # fp16's range when they are rounded for the matrix cores; the flat gradient buffer is unscaled in one pass
        scale = float(getattr(self._flags, 'LOSS_SCALE', 1.0) or 1.0)
        # SUM over ranks (the reference loss is a sum over all events).  Sparse model with ONE executor backward in this step:
        # the decoder + bottom + head suffix of the flat buffer is all-reduced from inside the backward pass, overlapped with
        # its encoder half, the prefix behind it (parallel.OverlappedAllReduce); otherwise ONE collective over the whole
        # buffer behind the backward kernels.  (N > 1 over RCCL is unmeasured on hardware so far -- DESIGN section 5.)
        overlap = getattr(self, '_overlap', None)
        armed = has_graph and overlap is not None and getattr(self, '_n_trunk_fwd', 0) >= 1 and \
            overlap.arm(self._net, expected=self._n_trunk_fwd)
        self._n_trunk_fwd = 0
        if has_graph:
            # the incoming gradient of the loss IS the scale: one cached scalar instead of a mul and a ones-fill launch per step
            seeds = self.__dict__.setdefault('_loss_seeds', {})
            key = (total_loss.device, total_loss.dtype, tuple(total_loss.shape), scale)
            if key not in seeds:
                seeds[key] = torch.full(tuple(total_loss.shape), scale, dtype=total_loss.dtype, device=total_loss.device)
            total_loss.backward(seeds[key])
        if armed:
            self.last_collectives = overlap.finish()
        else:
            self._grads.all_reduce()
            self.last_collectives = 1
        if has_graph and scale != 1.0:
            self._grads.flat.mul_(1.0 / scale)        # (after the sum: unscaling commutes with it)
        self.skipped_step = False
        if scale != 1.0:
            # loss scaling: an operand that overflowed at this scale leaves inf / NaN in the gradients, and Adam's moments
            # would keep it for good.  Checked AFTER the all-reduce (a non-finite value survives the sum, so every rank takes
            # the same decision); one host read-back, on the scaled (-prec fp16) path only
            if not bool(torch.isfinite(self._grads.flat).all()):
                self.skipped_step = True
                self.skipped_steps = getattr(self, 'skipped_steps', 0) + 1
                sys.stderr.write('WARNING: non-finite gradients at loss scale %g: optimizer step skipped (%d so far); '
                                 'lower -ls\n' % (scale, self.skipped_steps))
                return
        self._optimizer.step()

    # -- reference trainval.py:32-40
    def save_state(self, iteration):
        tstart = time.time()
        filename = '%s-%d.ckpt' % (self._flags.WEIGHT_PREFIX, iteration)
        if self._rank == 0:
            state = self._net.state_dict()
            if getattr(self._flags, 'CKPT_MODULE_PREFIX', False):
                # the reference saves a DataParallel-wrapped module: every key carries 'module.' (reference :37, :181);
                # with this flag a checkpoint written here loads into the reference with strict=False as well
                state = {'module.' + k: v for k, v in state.items()}
            torch.save({
                'global_step': iteration,
                'state_dict': state,
                'optimizer': self._optimizer.state_dict()
            }, filename)
        self.tspent['save'] = time.time() - tstart

    # -- reference trainval.py:42-51
    def _graph_step(self, data_blob, batch_size):
        """flags -graph: the dense training step (forward + loss + backward) replayed from a captured HIP graph
        (graphed.GraphedDenseStep: ~1,100 launches per step otherwise issued from Python).  Only for ONE sub-step of fixed
        shape on one rank; returns None when the step does not qualify (the eager path then runs)."""
        if not (getattr(self._flags, 'GRAPH', False) and self._flags.TRAIN and 'dense' in self._flags.MODEL_NAME
                and self._device.type == 'cuda' and self._world == 1 and len(data_blob['data']) == 1
                and data_blob.get('label') is not None):
            return None
        from .graphed import GraphedDenseStep
        data = torch.stack([torch.as_tensor(d) for d in data_blob['data'][0]]).to(self._device)
        label = torch.stack([torch.as_tensor(l) for l in data_blob['label'][0]]).to(self._device)
        weight = data_blob.get('weight')
        weight = None if weight is None else torch.stack([torch.as_tensor(w) for w in weight[0]]).to(self._device)
        key = (tuple(data.shape), tuple(label.shape), None if weight is None else tuple(weight.shape))
        scale = float(getattr(self._flags, 'LOSS_SCALE', 1.0) or 1.0)
        if getattr(self, '_gstep_key', None) != key:
            self._gstep = GraphedDenseStep(self._net, self._criterion, data, label, weight, zero_grad=self._grads.zero,
                                           loss_scale=scale)
            self._gstep_key = key
        loss, acc = self._gstep(data, label, weight)
        if scale != 1.0:
            self._grads.flat.mul_(1.0 / scale)
        self.skipped_step = False
        if scale != 1.0 and not bool(torch.isfinite(self._grads.flat).all()):   # as in backward(): no step on overflow
            self.skipped_step = True
            self.skipped_steps = getattr(self, 'skipped_steps', 0) + 1
            sys.stderr.write('WARNING: non-finite gradients at loss scale %g: optimizer step skipped\n' % scale)
        else:
            self._optimizer.step()
        self.last_slots = list(range(data.shape[0]))
        out = self._gstep.out
        return {'segmentation': [out[i].detach().clone() for i in range(out.shape[0])], 'accuracy': [acc.clone()],
                'loss_seg': [loss.clone()]}

    def train_step(self, data_blob, epoch=None, batch_size=1):
        tstart = time.time()
        res_combined = self._graph_step(data_blob, batch_size)
        if res_combined is not None:
            res_combined = self._finish(res_combined, batch_size)
            self.tspent['train'] = time.time() - tstart
            self.tspent_sum['train'] += self.tspent['train']
            return res_combined
        self._loss = []
        # the host copies of logits/softmax/loss/accuracy (reference :126-131) are made AFTER the backward pass and
        # the optimizer step have been enqueued: same dict, no host stall between forward and backward
        res_combined = self.forward(data_blob, epoch=epoch, batch_size=batch_size, _defer=True)
        self.backward()
        res_combined = self._finish(res_combined, batch_size)
        self.tspent['train'] = time.time() - tstart
        self.tspent_sum['train'] += self.tspent['train']
        return res_combined

    # -- reference trainval.py:53-73
    def forward(self, data_blob, epoch=None, batch_size=1, _defer=False):
        res_combined = {}
        for idx in range(len(data_blob['data'])):
            blob = {}
            for key in data_blob.keys():
                blob[key] = data_blob[key][idx]
            res = self._forward(blob, epoch=epoch)
            for key in res.keys():
                if key not in res_combined:
                    res_combined[key] = res[key]
                else:
                    res_combined[key].extend(res[key])
        return res_combined if _defer else self._finish(res_combined, batch_size)

    def _finish(self, res_combined, batch_size):
        """device results -> the reference's host dict: numpy logits/softmax per entry, loss and accuracy summed
        over entries (and ranks) and divided by batch_size (reference :71-72, :126-131)"""
        seg = res_combined['segmentation']
        res_combined['softmax'] = [self._softmax(s).detach().cpu().numpy() for s in seg]
        res_combined['segmentation'] = [s.detach().cpu().numpy() for s in seg]
        acc = float(sum(float(a) for a in res_combined['accuracy']))
        loss = float(sum(l if isinstance(l, float) else l.item() for l in res_combined['loss_seg']))
        acc, loss = parallel.all_reduce_scalars([acc, loss], self._device)
        res_combined['accuracy'] = acc / batch_size
        res_combined['loss_seg'] = loss / batch_size
        return res_combined

    def _local_slots(self, sizes):
        """Entries of a per-GPU list that this rank owns.  The reference scatters contiguous chunks (uresnet/ops.py:
        28-36); here whole entries are assigned by greedy LPT on their sizes (active voxels / pixels), so that ranks
        finish together when events differ in size (parallel.shard_events; deterministic, the same on every rank)."""
        if self._world <= 1:
            return list(range(len(sizes)))
        return parallel.shard_events(sizes, self._world)[self._rank]

    # -- reference trainval.py:75-134
    def _forward(self, data_blob, epoch=None):
        data = data_blob['data']
        label = data_blob.get('label', None)
        weight = data_blob.get('weight', None)
        slots = self._local_slots([int(np.asarray(d.shape[0] if hasattr(d, 'shape') and len(d.shape) == 2 else np.prod(np.shape(d))))
                                   for d in data])
        self.last_slots = slots
        sparse = 'sparse' in self._flags.MODEL_NAME
        with torch.set_grad_enabled(self._flags.TRAIN):
            data = [torch.as_tensor(data[i]).to(self._device) for i in slots]
            tstart = time.time()
            if not slots:
                # no entry of this sub-step is ours: nothing to run; the loss term is a plain 0 (see backward())
                if label is not None and self._flags.TRAIN:
                    self._loss.append(0.)
                self.tspent['forward'] = time.time() - tstart
                self.tspent_sum['forward'] += self.tspent['forward']
                return {'segmentation': [], 'accuracy': [0.], 'loss_seg': [0.]}
            if sparse:
                segmentation = []
                for d in data:
                    segmentation.extend(self._net(d))
                if self._flags.TRAIN:
                    self._n_trunk_fwd = getattr(self, '_n_trunk_fwd', 0) + len(data)
            else:
                segmentation = list(self._net(torch.stack(data)))
            loss_seg, acc = 0., 0.
            if label is not None:
                label = [torch.as_tensor(label[i]).to(self._device) for i in slots]
                if weight is not None:
                    weight = [torch.as_tensor(weight[i]).to(self._device) for i in slots]
                loss_seg, acc = self._criterion(segmentation, data, label, weight)
                if self._flags.TRAIN:
                    self._loss.append(loss_seg)
            res = {   # still on the device: _finish() makes the host copies
                'segmentation': [s.detach() for s in segmentation],
                'accuracy': [acc],
                'loss_seg': [loss_seg if isinstance(loss_seg, float) else loss_seg.detach()]
            }
            self.tspent['forward'] = time.time() - tstart
            self.tspent_sum['forward'] += self.tspent['forward']
            return res

    # -- reference trainval.py:136-198
    def initialize(self):
        model = None
        if self._flags.MODEL_NAME == 'uresnet_sparse':
            model = models.SparseUResNet
            self._criterion = models.SparseSegmentationLoss(self._flags)
        elif self._flags.MODEL_NAME == 'uresnet_dense':
            model = models.DenseUResNet
            self._criterion = models.DenseSegmentationLoss(self._flags)
        else:
            raise Exception("Unknown model name provided")

        self.tspent_sum['forward'] = self.tspent_sum['train'] = self.tspent_sum['save'] = 0.
        self.tspent['forward'] = self.tspent['train'] = self.tspent['save'] = 0.

        self._rank, self._world, local_rank = parallel.init_distributed()
        use_gpu = torch.cuda.is_available() and len(getattr(self._flags, 'GPUS', [0])) > 0
        self._device = torch.device('cuda', local_rank) if use_gpu else torch.device('cpu')
        if use_gpu:
            torch.cuda.set_device(self._device)
        self._net = model(self._flags).to(self._device)
        if self._flags.TRAIN:
            self._net.train()
        else:
            self._net.eval()
        self._criterion.to(self._device)

        # gradients live in one flat buffer (one memset, one all-reduce per step); on the GPU the Adam step of the
        # reference (trainval.py:37) is one streaming pass over the flat buffers
        self._grads = parallel.FlatGradients(self._net)
        self._overlap = parallel.OverlappedAllReduce(self._grads, force=bool(os.environ.get('URN_SPLIT_ALLREDUCE'))) \
            if (use_gpu and 'sparse' in self._flags.MODEL_NAME) else None
        if use_gpu:
            self._optimizer = parallel.FlatAdam(self._grads, lr=self._flags.LEARNING_RATE)
        else:
            self._optimizer = torch.optim.Adam(self._net.parameters(), lr=self._flags.LEARNING_RATE)
        self._softmax = torch.nn.Softmax(dim=1 if 'sparse' in self._flags.MODEL_NAME else 0)

        iteration = 0
        if self._flags.MODEL_PATH:
            if not os.path.isfile(self._flags.MODEL_PATH):
                sys.stderr.write('File not found: %s\n' % self._flags.MODEL_PATH)
                raise ValueError
            print('Restoring weights from %s...' % self._flags.MODEL_PATH)
            with open(self._flags.MODEL_PATH, 'rb') as f:
                checkpoint = torch.load(f, map_location=self._device, weights_only=True)
            state = {(k[len('module.'):] if k.startswith('module.') else k): v
                     for k, v in checkpoint['state_dict'].items()}     # DataParallel prefix (reference :181)
            self._net.load_state_dict(state, strict=False)
            if self._flags.TRAIN:
                self._optimizer.load_state_dict(checkpoint['optimizer'])
                for g in self._optimizer.param_groups:
                    g['lr'] = self._flags.LEARNING_RATE
            iteration = checkpoint['global_step'] + 1
            print('Done.')
        parallel.broadcast_parameters(self._net)
        # what is alive now (modules, the imported libraries) stays alive: out of the cyclic collector's way, so that its full
        # passes (~70 ms each on the ~10^6 objects torch brings along, once every ~250 steps; tools/hiccup.py) stay short
        import gc
        gc.collect()
        gc.freeze()
        return iteration
