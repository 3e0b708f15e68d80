"""Fast path for the sparse trunk: the whole `sparseModel` (stem conv + UNet + final BNReLU +
OutputLayer) runs inside liburesnet_hip.so's executor (csrc/urn_net.hip) instead of one
autograd node per layer.  Same kernels, same results; the per-layer module tree stays the
flexible path (eval mode, hooks, tests).

Parameters of the module tree are re-pointed to views of ONE flat fp32 tensor (registration
order), BatchNorm running statistics likewise, and gradients are written straight into the flat
gradient buffer that the parameters' .grad tensors alias -- one memset and one all-reduce per step.
"""
import ctypes
import weakref

import torch

from . import lib as _l
from . import scn
from . import sparse_ops as so


MAX_SIDE_STREAMS = 8
_SIDE_HANDLES = 0      # executor handles of this process that own a side stream


class _Token:
    """Lives as long as the autograd node of one forward; a slot is busy while its token is alive."""
    pass


class _Slot:
    """One executor handle (it holds the tape of its last forward) plus its workspace."""

    def __init__(self, handle):
        self.handle = handle
        self.token = None
        self.ws = None
        self.ws_bytes = 0
        self.wbuf = None          # weight copies written ahead of the forward (prepare_weights)

    def busy(self):
        return self.token is not None and self.token() is not None


class TrunkExecutor:
    def __init__(self, sparse_model, m, num_levels, reps, num_class):
        self.model = sparse_model
        self.cfg = (m, num_levels, reps, num_class)
        self.flags = 0            # URN_NET_UNFUSED = 1, URN_NET_SINGLE_STREAM = 2, URN_NET_SLAB_STATS = 4 (debug / A-B switches)
        self.handle = None
        self.slots = []
        self.flat = None
        self.running = None
        self.flat_grad = None
        self._ws_coef = {}
        self._prep = None
        self._early_slot = None
        self._side_handles = 0
        # data-parallel overlap (parallel.OverlappedAllReduce): called from inside the backward pass with the offset of the
        # flat buffers' decoder + bottom suffix and the executor's side stream, once the kernels that write it are enqueued
        self.suffix_hook = None

    # -- handle ------------------------------------------------------------------------
    def _new_handle(self):
        """A new executor handle.  Every handle with a side stream adds a HIP stream to the process, and HIP multiplexes
        streams onto a few hardware queues: some (caller stream, side stream) pairs run a step 2.7x slower (measured:
        the 4th and 5th handle of a process).  The executor's first backward therefore probes a few candidate side
        streams and keeps the fastest (urn_net.hip pick_side); MAX_SIDE_STREAMS only bounds how many streams a process
        with many handles (many models, gradient accumulation over many forwards) creates -- further handles run
        single-stream."""
        global _SIDE_HANDLES
        L = _l.load()
        h = ctypes.c_void_p()
        flags = int(self.flags)
        with_side = not (flags & 2) and _SIDE_HANDLES < MAX_SIDE_STREAMS
        if not with_side:
            flags |= 2                                  # URN_NET_SINGLE_STREAM
        _l.check(L.urn_net_create(*self.cfg, float(self.eps), float(self.momentum), flags, ctypes.byref(h)), 'net_create')
        if with_side:
            _SIDE_HANDLES += 1
            self._side_handles += 1
            if torch.cuda.is_available():
                # the one synchronising call of the executor, here at creation (never inside a step): pick a side stream
                # that does not share a hardware queue with the caller's stream
                _l.check(L.urn_net_probe(h, _l.stream()), 'net_probe')
        return h

    def acquire(self):
        """A slot whose last forward has been consumed by backward (or dropped)."""
        for s in self.slots:
            if not s.busy():
                return s
        s = _Slot(self.handle if not self.slots else self._new_handle())
        self.slots.append(s)
        return s

    def _create(self):
        L = _l.load()
        bn = [mod for mod in self.model.modules() if isinstance(mod, scn.BatchNormLeakyReLU)]
        self.eps = bn[0].eps if bn else 1e-4
        self.momentum = bn[0].momentum if bn else 0.9
        h = self._new_handle()
        self.handle = h
        self.params = [p for p in self.model.parameters()]
        self.bns = bn
        nt = L.urn_net_num_tensors(h)
        assert nt == len(self.params), 'executor and module tree disagree on the number of tensors (%d vs %d)' % (
            nt, len(self.params))
        self.offsets = []
        off, num = ctypes.c_int64(), ctypes.c_int64()
        for i, p in enumerate(self.params):
            _l.check(L.urn_net_tensor(h, i, ctypes.byref(off), ctypes.byref(num)))
            assert num.value == p.numel(), 'tensor %d: %d vs %d elements' % (i, num.value, p.numel())
            self.offsets.append(off.value)
        self.n_params = L.urn_net_param_count(h)
        self.n_running = L.urn_net_running_count(h)
        assert self.n_running == sum(2 * b.nPlanes for b in bn)

    def __del__(self):
        global _SIDE_HANDLES
        try:
            _SIDE_HANDLES -= getattr(self, '_side_handles', 0)
        except TypeError:        # interpreter shutdown: module globals are already gone
            pass
        try:
            handles = [s.handle for s in self.slots] or ([self.handle] if self.handle is not None else [])
            for h in handles:
                _l.load().urn_net_destroy(h)
        except Exception:
            pass

    # -- flat storage ------------------------------------------------------------------
    def _aliased(self, dev, tail=()):
        if self.flat is None or self.flat.device != dev:
            return False
        base = self.flat.data_ptr()
        o = self.n_params
        for p in tail:
            if p.data_ptr() != base + 4 * o:
                return False
            o += p.numel()
        # re-homing moves every parameter at once: three probes are enough (190 data_ptr() calls per step otherwise)
        probe = (0, len(self.params) // 2, len(self.params) - 1)
        return all(self.params[i].data_ptr() == base + 4 * self.offsets[i] for i in probe)

    def flatten(self, dev, tail=()):
        """(Re)point parameters and BN buffers at the flat tensors; values are preserved.  `tail`: further
        parameters of the model (the Linear head) homed right behind the trunk's, so that the whole model is ONE
        contiguous segment for the optimizer."""
        if self.handle is None:
            self._create()
        if self._aliased(dev, tail):
            return
        flat = torch.empty(self.n_params + sum(p.numel() for p in tail), dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
            p.data = flat[o:o + p.numel()].view(p.shape)
        o = self.n_params
        for p in tail:
            flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
            p.data = flat[o:o + p.numel()].view(p.shape)
            o += p.numel()
        running = torch.empty(max(self.n_running, 1), dtype=torch.float32, device=dev)
        o = 0
        for b in self.bns:
            c = b.nPlanes
            running[o:o + c].copy_(b.running_mean); running[o + c:o + 2 * c].copy_(b.running_var)
            b.running_mean = running[o:o + c]; b.running_var = running[o + c:o + 2 * c]
            o += 2 * c
        self.flat, self.running = flat, running
        self.flat_grad = None

    def grad_buffer(self, tail=()):
        """Flat gradient buffer that every trunk parameter's .grad aliases.  Adopts an existing flat
        buffer (parallel.FlatGradients) when the .grad tensors already sit at the right offsets.  `tail`: the head's
        parameters (homed behind the trunk's by flatten), whose gradients the executor writes right behind the trunk's."""
        params = list(self.params) + list(tail)
        offsets = list(self.offsets)
        o = self.n_params
        for p in tail:
            offsets.append(o); o += p.numel()
        total = o
        g0 = params[0].grad
        if g0 is not None:
            base = g0.data_ptr() - 4 * offsets[0]
            if all(p.grad is not None and p.grad.data_ptr() == base + 4 * o for p, o in zip(params, offsets)):
                return base, None
        if self.flat_grad is None or self.flat_grad.device != self.flat.device or self.flat_grad.numel() < total:
            self.flat_grad = torch.zeros(total, dtype=torch.float32, device=self.flat.device)
        fresh = all(p.grad is None for p in params)
        if not fresh:
            # gradients exist elsewhere (accumulation by the user): carry them over once
            for p, o in zip(params, offsets):
                if p.grad is not None and p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                    self.flat_grad[o:o + p.numel()].copy_(p.grad.reshape(-1))
                elif p.grad is None:
                    self.flat_grad[o:o + p.numel()].zero_()
        else:
            self.flat_grad.zero_()      # zero_grad(set_to_none=True) semantics
        for p, o in zip(params, offsets):
            p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)
        return self.flat_grad.data_ptr(), self.flat_grad

    def _ws_model(self, slot, num_levels, with_backward):
        """The workspace need is linear in the level sizes and the row count (every allocation is rows x channels,
        rounded up to 256 B): one dry run per unit vector, cached, replaces a dry run of the whole graph per step --
        that call sat between the level-count synchronisation and the first kernel of the forward pass."""
        key = (num_levels, int(with_backward), so.DETERMINISTIC_DW)   # (the two-stage weight gradient needs its partial slabs)
        if key not in self._ws_coef:
            L = _l.load()
            U = 1 << 16

            def need(n, rows):
                v = L.urn_net_workspace_bytes(slot.handle, num_levels, (ctypes.c_int64 * num_levels)(*n), rows, int(with_backward))
                if v < 0:
                    raise RuntimeError('urn_net_workspace_bytes failed')
                return v
            zero = [0] * num_levels
            f0 = need(zero, 0)
            coef = [(need([U if k == l else 0 for k in range(num_levels)], 0) - f0) / U for l in range(num_levels)]
            crow = (need(zero, U) - f0) / U
            self._ws_coef[key] = (f0, coef, crow)
        return self._ws_coef[key]

    def workspace(self, slot, geo, with_backward):
        f0, coef, crow = self._ws_model(slot, geo.num_levels, with_backward)
        need = int((f0 + sum(c * n for c, n in zip(coef, geo.n)) + crow * geo.n_rows) * 1.01) + (1 << 20)
        if slot.ws is None or slot.ws_bytes < need or slot.ws.device != geo.device:
            slot.ws_bytes = int(need * 1.25)
            slot.ws = torch.empty(slot.ws_bytes, dtype=torch.uint8, device=geo.device)
        return slot.ws, slot.ws_bytes

    def prepare_weights(self):
        """The weight copies of the coming forward (transposed + fragment orders), enqueued NOW on the executor's side
        stream: call it before the integer phase is enqueued, they then run beside it (urn_net_prepare_weights)."""
        slot = self.acquire()
        self._early_slot = slot
        n3 = 3 * self.n_params
        if slot.wbuf is None or slot.wbuf.numel() < n3 or slot.wbuf.device != self.flat.device:
            slot.wbuf = torch.empty(n3, dtype=torch.float32, device=self.flat.device)
        _l.check(_l.load().urn_net_prepare_weights(slot.handle, self.flat.data_ptr(), slot.wbuf.data_ptr(), n3, _l.stream()),
                 'net_prepare_weights')

    def prepare(self, geo, training, head=None):
        """Host-side preparation of forward() that does not need the level counts; call it BEFORE geo.sync().
        head = (weight, bias) of the Linear: run inside the executor (urn_net_set_head), the output is then the logits."""
        Lv = geo.num_levels
        slot, self._early_slot = (self._early_slot or self.acquire()), None
        need_bwd = bool(training and torch.is_grad_enabled())
        self._ws_model(slot, Lv, need_bwd)
        self._prep = dict(
            geo=geo, slot=slot, need_bwd=need_bwd,
            nbr=(ctypes.c_void_p * Lv)(*[t.data_ptr() for t in geo.nbr]),
            chd=(ctypes.c_void_p * Lv)(*([t.data_ptr() for t in geo.chd] + [None])),
            up=(ctypes.c_void_p * Lv)(*([t.data_ptr() for t in geo.up] + [None])),
            p_nbr=(ctypes.c_void_p * Lv)(*[t[0].data_ptr() for t in geo.pairs['nbr']]),
            p_chd=(ctypes.c_void_p * Lv)(*([t[0].data_ptr() for t in geo.pairs['chd']] + [None])),
            p_up=(ctypes.c_void_p * Lv)(*([t[0].data_ptr() for t in geo.pairs['up']] + [None])),
            t_nbr=(ctypes.c_int * Lv)(*[t[1] for t in geo.pairs['nbr']]),
            t_chd=(ctypes.c_int * Lv)(*([t[1] for t in geo.pairs['chd']] + [0])),
            t_up=(ctypes.c_int * Lv)(*([t[1] for t in geo.pairs['up']] + [0])),
            head=head,
            out=torch.empty((geo.n_rows, self.cfg[0] if head is None else head[0].shape[0]), dtype=torch.float32, device=geo.device))

    def head_ok(self, weight, bias):
        """Can the Linear head run inside the executor?  Fused path with accumulated statistics, m <= 32, nc <= 8, and the
        head's parameters homed right behind the trunk's (flatten(tail=...)), so that their gradients are one buffer."""
        if self.flags & 5 or self.cfg[0] > 32 or weight.shape[0] > 8 or bias is None or self.flat is None:
            return False
        base = self.flat.data_ptr() + 4 * self.n_params
        return weight.data_ptr() == base and bias.data_ptr() == base + 4 * weight.numel()

    def forward(self, geo, feats, training, head=None):
        """site features -> (n_rows, m) rows in input order, or the (n_rows, num_class) logits with head = (weight, bias).
        Recorded for backward when training."""
        need_bwd = bool(training and torch.is_grad_enabled())   # grad mode is off inside Function.forward
        return _TrunkFunction.apply(feats, self.params[0], self, geo, need_bwd, bool(training), head)


_BOTTOM_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p)


class _TrunkFunction(torch.autograd.Function):
    """The trunk as one autograd node.  The stem weight rides along as an input only so that
    autograd schedules backward(); parameter gradients are written into the flat gradient
    buffer directly (the parameters' .grad tensors alias it)."""

    @staticmethod
    def forward(ctx, feats, anchor, ex, geo, need_bwd, training=True, head=None):
        L = _l.load()
        Lv = geo.num_levels
        prep = ex._prep if (ex._prep is not None and ex._prep['geo'] is geo and ex._prep['need_bwd'] == need_bwd
                            and (ex._prep['head'] is None) == (head is None)) else None
        ex._prep = None
        if prep is None:
            ex.prepare(geo, need_bwd, head)
            prep, ex._prep = ex._prep, None
        slot, nbr, chd, up, out = prep['slot'], prep['nbr'], prep['chd'], prep['up'], prep['out']
        geo.sync()
        ws, ws_bytes = ex.workspace(slot, geo, need_bwd)
        n = (ctypes.c_int64 * Lv)(*geo.n)
        feats = feats.contiguous()
        _l.check(L.urn_net_set_pairs(slot.handle, Lv, prep['p_nbr'], prep['p_chd'], prep['p_up'], prep['t_nbr'], prep['t_chd'],
                                     prep['t_up']), 'net_set_pairs')
        if head is not None:
            _l.check(L.urn_net_set_head(slot.handle, head[0].data_ptr(), head[1].data_ptr()), 'net_set_head')
        _l.check(L.urn_net_forward(slot.handle, Lv, geo.ld, n, nbr, chd, up, geo.row2site.data_ptr(), geo.n_rows,
                                   ex.flat.data_ptr(), ex.running.data_ptr(), feats.data_ptr(), ws.data_ptr(),
                                   ws_bytes, out.data_ptr(), int(training), _l.stream()), 'net_forward')
        ctx.ex, ctx.geo, ctx.ws, ctx.feats, ctx.slot, ctx.head = ex, geo, ws, feats, slot, head   # keep workspace/inputs alive
        if need_bwd:
            ctx.token = _Token()
            slot.token = weakref.ref(ctx.token)    # busy until backward ran or the graph is dropped
        return out

    @staticmethod
    def backward(ctx, d_rows):
        L = _l.load()
        ex = ctx.ex
        gptr, keep = ex.grad_buffer(() if ctx.head is None else ctx.head)
        d_rows = d_rows.contiguous()
        hook = ex.suffix_hook
        if hook is None:
            _l.check(L.urn_net_backward(ctx.slot.handle, d_rows.data_ptr(), gptr, _l.stream()), 'net_backward')
        else:
            err = []
            h = ctx.slot.handle

            def bottom_done(_user):
                try:
                    hook(int(L.urn_net_suffix_offset(h)), L.urn_net_side_stream(h))
                except BaseException as e:      # (an exception cannot cross the C frames: kept and raised below)
                    err.append(e)
            cb = _BOTTOM_CB(bottom_done)
            _l.check(L.urn_net_backward_cb(h, d_rows.data_ptr(), gptr, _l.stream(), ctypes.cast(cb, ctypes.c_void_p), None), 'net_backward')
            if err:
                raise err[0]
        ctx.slot.token = None
        return None, None, None, None, None, None, None
