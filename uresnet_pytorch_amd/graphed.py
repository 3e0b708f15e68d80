"""Training step of the DENSE model replayed from a captured HIP graph (torch.cuda.CUDAGraph).

The dense route is ~1,100 kernel launches per step issued from Python (one autograd node per convolution / BatchNorm pass):
at BASELINE configs[1] the kernels sum to 20.9 ms, the eager step takes 23.5 ms -- the difference is launch gaps and host
time.  The step has static shapes and no host synchronisation (statistics, loss and accuracy stay on the device, the
library launches on torch's current stream), so forward + loss + backward can be captured once and replayed:
20.3 ms per step measured (`URN_GRAPH=1 python tools/run_dense_cfg2.py 128 5 1`).

    step = GraphedDenseStep(net, criterion, data, label)      # warm-up + capture; data (B, 1, *S), label like the trainer's
    loss, acc = step(data, label)                             # copies the inputs into the static buffers, replays
    optimizer.step()                                          # gradients are in p.grad (static tensors of the graph)

The optimizer stays outside the graph (flat Adam passes the step count as a kernel argument).  Sparse steps are not
captured: their launch shapes depend on the event (and the executor is already one C++ call per pass).
"""
import torch


class GraphedDenseStep(object):
    def __init__(self, net, criterion, data, label, weight=None, warmup=3, zero_grad=None, loss_scale=1.0):
        """zero_grad: callable that clears the gradients IN PLACE (parallel.FlatGradients.zero: the parameters' .grad stay views
        of the flat buffer, the backward pass accumulates into them); default: net.zero_grad(set_to_none=True).
        loss_scale: the loss is multiplied by it before backward (flags -ls); the caller unscales the gradients."""
        assert data.is_cuda, 'graph capture needs GPU tensors'
        self.net, self.crit = net, criterion
        self.zero_grad, self.loss_scale = zero_grad, float(loss_scale)
        self.data, self.label = data.clone(), label.clone()
        self.weight = None if weight is None else weight.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                   # one-time work (function attributes, pools, lazy buffers) before the capture
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.acc, self.out = self._step()
        torch.cuda.synchronize()
        # the gradient tensors the graph writes (its private pool): re-attached after every replay, whatever happened to
        # p.grad in between (an eager step, zero_grad(set_to_none=True))
        self.grads = [(p, p.grad) for p in self.net.parameters() if p.grad is not None]
        # buffers of the library's module-level caches that the captured launches address: kept alive with the graph
        from . import dense_conv as _dc
        self._keep = (_dc._POOL.buf, dict(_dc._SCRATCH))

    def _step(self):
        if self.zero_grad is not None:
            self.zero_grad()
        else:
            self.net.zero_grad(set_to_none=True)
        out = self.net(self.data)
        w = None if self.weight is None else [self.weight[i] for i in range(self.weight.shape[0])]
        loss, acc = self.crit(out, self.data, self.label, w)
        (loss * self.loss_scale if self.loss_scale != 1.0 else loss).backward()
        acc_t = getattr(acc, '_t', None)              # utils.DeferredFloat: the accuracy tensor, still on the device
        return loss.detach(), acc_t, out.detach()

    def __call__(self, data, label, weight=None):
        self.data.copy_(data, non_blocking=True)
        self.label.copy_(label, non_blocking=True)
        if self.weight is not None and weight is not None:
            self.weight.copy_(weight, non_blocking=True)
        self.graph.replay()
        for p, g in self.grads:
            p.grad = g
        return self.loss, self.acc
