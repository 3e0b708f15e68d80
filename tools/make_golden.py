"""Generates tests/golden/dense_*.npz by importing the REFERENCE dense model in this container
(/root/reference/uresnet/models/uresnet_dense.py, loaded by file path with the py2 `xrange` shim;
SURVEY.md 8c).  Runs only here; the .npz files (inputs + expected outputs + weights) are the
committed fixtures.  No reference source is copied."""
import builtins
import importlib.util
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference/uresnet/models/uresnet_dense.py'


def load_reference():
    builtins.xrange = range
    spec = importlib.util.spec_from_file_location('ref_uresnet_dense', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make(name, dim, ss, uf, uns, nc, B, seed):
    from uresnet_pytorch_amd.iotools.synthetic import make_dense_blob
    ref = load_reference()
    flags = SimpleNamespace(DATA_DIM=dim, URESNET_FILTERS=uf, URESNET_NUM_STRIDES=uns, SPATIAL_SIZE=ss,
                            NUM_CLASS=nc, BN_MOMENTUM=0.9)
    torch.manual_seed(seed)
    net = ref.UResNet(flags).train()
    crit = ref.SegmentationLoss(flags)
    blob = make_dense_blob(list(range(seed, seed + B)), ss, dim, nc, fill=300 if dim == 3 else None)
    x = torch.from_numpy(blob['data']); lab = torch.from_numpy(blob['label'])
    rng = np.random.default_rng(seed)
    w = torch.from_numpy(rng.uniform(0.5, 2.0, size=blob['label'].shape).astype(np.float32))
    out = {}
    # every ReLU output of the reference's forward, in call order (nn.ReLU and F.relu both end in torch.nn.functional.relu):
    # the masks let the parity tests pin ReLU branches to the REFERENCE's own, so that gradients are compared as arithmetic
    # (a pre-activation within fp32 rounding of zero takes either branch in any two evaluation orders)
    import torch.nn.functional as F_
    relu_masks = []
    orig_relu = F_.relu

    def recording_relu(inp, inplace=False):
        y = orig_relu(inp, inplace=inplace)
        relu_masks.append((y > 0).numpy().copy())
        return y
    F_.relu = recording_relu
    try:
        logits = net(x)
    finally:
        F_.relu = orig_relu
    loss, acc = crit(list(logits), list(x), list(lab), None)
    net.zero_grad(); loss.backward()
    grads = {k: p.grad.detach().numpy().copy() for k, p in net.named_parameters() if p.grad is not None}
    logits_w = net(x)
    loss_w, acc_w = crit(list(logits_w), list(x), list(lab), list(w))
    for k, v in net.state_dict().items():
        out['sd/' + k] = v.detach().numpy()
    keep = ('conv1.0.weight', 'conv1.1.weight', 'double_resnet.1.resnet1.residual1.0.weight',
            'double_resnet.0.resnet1.shortcut.0.weight', 'double_resnet.1.resnet2.residual2.1.bias',
            'decode_conv.0.0.weight', 'decode_double_resnet.1.resnet1.shortcut.0.weight', 'conv2.0.weight',
            'conv3.0.weight', 'conv3.1.bias')
    for k, v in grads.items():
        if k in keep:
            out['grad/' + k] = v
    out['grad_keys_with_grad'] = np.array(sorted(grads.keys()))
    for i, mk in enumerate(relu_masks):
        out['relu_mask/%03d' % i] = np.packbits(mk.reshape(-1))
        out['relu_shape/%03d' % i] = np.array(mk.shape)
    out.update(dict(input=blob['data'], label=blob['label'], weight=w.numpy(), logits=logits.detach().numpy(),
                    loss=np.float64(loss.item()), acc=np.float64(acc), loss_w=np.float64(loss_w.item()),
                    acc_w=np.float64(acc_w), flags=np.array([dim, ss, uf, uns, nc, B])))
    # padding() table
    pads = []
    for k, s, n in [(3, 1, 64), (3, 2, 64), (1, 2, 64), (3, 2, 65), (1, 1, 7), (3, 1, 5)]:
        pads.append([k, s, n] + list(ref.padding(k, s, (1, 1, n, n))))
    out['padding_table'] = np.array(pads)
    path = os.path.join(ROOT, 'tests', 'golden', name + '.npz')
    np.savez_compressed(path, **out)
    print(name, 'params', sum(p.numel() for p in net.parameters()), 'keys', len(net.state_dict()),
          'loss', loss.item(), 'acc', acc, 'size %.1f KB' % (os.path.getsize(path) / 1024))


if __name__ == '__main__':
    make('dense_cfg1_2d', 2, 64, 8, 3, 5, 2, 0)       # BASELINE configs[0]
    make('dense_mini_3d', 3, 16, 4, 2, 5, 2, 1)
