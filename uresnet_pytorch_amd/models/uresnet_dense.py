"""Dense U-ResNet behind the reference's nn.Module surface (2-D or 3-D).

Same class names, constructor, forward contract, parameter names and loss semantics as
reference uresnet/models/uresnet_dense.py:12-260 (168 state_dict keys at -dd 2 -uf 8 -uns 3,
including the registered-but-unused shortcut convs of identity ResNetModules, :36-46 vs :72-73).
Arithmetic goes through uresnet_pytorch_amd.dense_ops: torch/ATen on CPU tensors (BASELINE
configs[0] is the reference's CPU plumbing case), the dense HIP kernels on GPU tensors.
"""
import torch
import torch.nn as nn

from .. import dense_ops as D
from ..utils import DeferredFloat


def get_conv(is_3d):
    if is_3d:
        return nn.Conv3d, nn.ConvTranspose3d, nn.BatchNorm3d
    return nn.Conv2d, nn.ConvTranspose2d, nn.BatchNorm2d


def padding(kernel, stride, input_size):
    """Asymmetric 'same' padding per spatial dim (reference :19-26):
    k3 s1 -> (1,1); k3 s2 on even sizes -> (0,1); k1 s2 -> (0,0)."""
    if input_size[-1] % stride == 0:
        p = max(kernel - stride, 0)
    else:
        p = max(kernel - (input_size[-1] % stride), 0)
    p1 = int(p // 2)
    p2 = p - p1
    return (p1, p2,) * (len(input_size) - 2)


def _conv_bn(seq, x, relu=False, residual=None, defer=False):
    """replicate-pad -> conv -> batch-stat BN (-> +residual) (-> ReLU) of a Sequential(conv, bn[, ReLU])."""
    conv, bn = seq[0], seq[1]
    pad = padding(conv.kernel_size[0], conv.stride[0], x.size())
    return D.conv_bn_act(x, conv.weight, conv.bias, conv.stride[0], pad, bn.weight, bn.bias, bn.eps, relu, residual, defer)


class ResNetModule(nn.Module):
    def __init__(self, is_3d, num_inputs, num_outputs, kernel=3, stride=1, bn_momentum=0.9):
        super(ResNetModule, self).__init__()
        fn_conv, fn_conv_transpose, batch_norm = get_conv(is_3d)
        self.kernel, self.stride = kernel, stride
        self.use_shortcut = (num_outputs != num_inputs or stride != 1)
        self.shortcut = torch.nn.Sequential(
            fn_conv(in_channels=num_inputs, out_channels=num_outputs, kernel_size=1, stride=stride, padding=0),
            batch_norm(num_features=num_outputs, momentum=bn_momentum, track_running_stats=False))
        self.residual1 = torch.nn.Sequential(
            fn_conv(in_channels=num_inputs, out_channels=num_outputs, kernel_size=kernel, stride=stride, padding=0),
            batch_norm(num_features=num_outputs, momentum=bn_momentum, track_running_stats=False))
        self.residual2 = torch.nn.Sequential(
            fn_conv(in_channels=num_outputs, out_channels=num_outputs, kernel_size=kernel, stride=1, padding=0),
            batch_norm(num_features=num_outputs, momentum=bn_momentum, track_running_stats=False))

    def forward(self, input_tensor):
        shortcut = _conv_bn(self.shortcut, input_tensor, defer=True) if self.use_shortcut else input_tensor
        # no ReLU between the two (reference :78-81): on the GPU route residual1's BatchNorm stays pending and is folded into
        # residual2's load (defer)
        residual = _conv_bn(self.residual1, input_tensor, defer=True)
        return _conv_bn(self.residual2, residual, relu=True, residual=shortcut)   # relu(shortcut + residual)


class DoubleResnet(nn.Module):
    def __init__(self, is_3d, num_inputs, num_outputs, kernel=3, stride=1, bn_momentum=0.9):
        super(DoubleResnet, self).__init__()
        self.resnet1 = ResNetModule(is_3d=is_3d, num_inputs=num_inputs, num_outputs=num_outputs, kernel=kernel,
                                    stride=stride, bn_momentum=bn_momentum)
        self.resnet2 = ResNetModule(is_3d=is_3d, num_inputs=num_outputs, num_outputs=num_outputs, kernel=kernel,
                                    stride=1, bn_momentum=bn_momentum)

    def forward(self, input_tensor):
        return self.resnet2(self.resnet1(input_tensor))


class UResNet(nn.Module):
    def __init__(self, flags):
        super(UResNet, self).__init__()
        self._flags = flags
        self.is_3d = flags.DATA_DIM == 3
        fn_conv, fn_conv_transpose, batch_norm = get_conv(self.is_3d)
        self.base_num_outputs = flags.URESNET_FILTERS
        self.num_strides = flags.URESNET_NUM_STRIDES
        self.num_inputs = 1
        self.image_size = flags.SPATIAL_SIZE
        self.num_classes = flags.NUM_CLASS
        mom = flags.BN_MOMENTUM   # inert: track_running_stats=False (reference :45,57,68,135)

        self.conv1 = torch.nn.Sequential(
            fn_conv(in_channels=self.num_inputs, out_channels=self.base_num_outputs, kernel_size=3, stride=1,
                    padding=0),
            batch_norm(num_features=self.base_num_outputs, momentum=mom, track_running_stats=False),
            torch.nn.ReLU())
        self.double_resnet = nn.ModuleList()
        current_num_outputs = self.base_num_outputs
        for step in range(self.num_strides):
            self.double_resnet.append(DoubleResnet(is_3d=self.is_3d, num_inputs=current_num_outputs,
                                                   num_outputs=current_num_outputs * 2, kernel=3, stride=2,
                                                   bn_momentum=mom))
            current_num_outputs *= 2
        self.decode_conv = nn.ModuleList()
        self.decode_double_resnet = nn.ModuleList()
        for step in range(self.num_strides):
            self.decode_double_resnet.append(DoubleResnet(is_3d=self.is_3d, num_inputs=current_num_outputs,
                                                          num_outputs=int(current_num_outputs / 2), kernel=3,
                                                          stride=1, bn_momentum=mom))
            self.decode_conv.append(torch.nn.Sequential(
                fn_conv_transpose(in_channels=current_num_outputs, out_channels=int(current_num_outputs / 2),
                                  kernel_size=3, stride=2, padding=1, output_padding=1),
                batch_norm(num_features=int(current_num_outputs / 2), momentum=mom, track_running_stats=False),
                torch.nn.ReLU()))
            current_num_outputs = int(current_num_outputs / 2)
        self.conv2 = torch.nn.Sequential(
            fn_conv(in_channels=current_num_outputs, out_channels=self.base_num_outputs, padding=0, kernel_size=3,
                    stride=1),
            batch_norm(num_features=current_num_outputs, momentum=mom, track_running_stats=False),
            torch.nn.ReLU())
        self.conv3 = torch.nn.Sequential(
            fn_conv(in_channels=self.base_num_outputs, out_channels=self.num_classes, padding=0, kernel_size=3,
                    stride=1),
            batch_norm(num_features=self.num_classes, momentum=mom, track_running_stats=False))

    def forward(self, input):
        """input (B, C, (N,)*dim) -> logits (B, num_classes, (N,)*dim), no softmax."""
        if input.is_cuda:
            from .. import dense_conv as _dc
            prec = getattr(self._flags, 'PRECISION', 'fp32')                # flags -prec: fp32 (default) | bf16
            if prec not in ('fp32', 'bf16'):
                raise RuntimeError('dense kernels: MFMA operand precision fp32 or bf16 (got %s)' % prec)
            _dc.set_precision(prec)
            _dc.pool_begin(input.device)        # one memset for the step's fp64 accumulation slabs
            # both kernel layouts of every convolution's weight in one launch (the parameters as they are NOW: the backward
            # pass of this forward reads the input-gradient layouts written here)
            if getattr(self, '_conv_list', None) is None:
                self._conv_list = [(mod.weight, isinstance(mod, (nn.ConvTranspose2d, nn.ConvTranspose3d)))
                                   for mod in self.modules() if isinstance(mod, (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d))]
            _dc.prepare_weights(self._conv_list)
        conv_feature_map = {}
        net = _conv_bn(self.conv1, input, relu=True)
        conv_feature_map[net.size()[1]] = net            # skip links keyed by channel count (reference :210,214)
        for step in range(self.num_strides):
            net = self.double_resnet[step](net)
            conv_feature_map[net.size()[1]] = net
        for step in range(self.num_strides):
            dc = self.decode_conv[step]
            net = D.convT_bn_act(net, dc[0].weight, dc[0].bias, dc[1].weight, dc[1].bias, dc[1].eps, True)
            net = torch.cat((net, conv_feature_map[net.size()[1]]), dim=1)
            net = self.decode_double_resnet[step](net)
        net = _conv_bn(self.conv2, net, relu=True)
        net = _conv_bn(self.conv3, net)
        return net


class SegmentationLoss(torch.nn.modules.loss._Loss):
    """Per-event CE over all pixels, masked to data > 1e-6, normalised by the non-zero count,
    optional per-pixel weight; SUM over events (reference uresnet_dense.py:235-260)."""

    def __init__(self, flags, reduction='sum'):
        super(SegmentationLoss, self).__init__(reduction=reduction)
        self._flags = flags
        self.cross_entropy = torch.nn.CrossEntropyLoss(reduction='none')

    def forward(self, segmentation, data, label, weight):
        total_loss = 0.
        total_acc = 0.
        assert len(segmentation) == len(data)
        assert len(data) == len(label)
        if weight is not None:
            assert len(weight) == len(label)
        for i in range(len(data)):
            if segmentation[i].is_cuda:
                # one fused pass per event on the device; the accuracy stays there until it is looked at
                from .. import dense_hip
                loss_i, out = dense_hip.segmentation_loss_event(segmentation[i], data[i], label[i],
                                                                None if weight is None else weight[i])
                total_loss = loss_i if i == 0 else total_loss + loss_i
                total_acc = out[1] if i == 0 else total_acc + out[1]
                continue
            nonzero_idx = data[i] > 0.000001
            event_segmentation = segmentation[i].unsqueeze(0)
            event_label = label[i].squeeze(0).unsqueeze(0).long()
            loss = self.cross_entropy(event_segmentation, event_label)
            if weight is not None:
                loss = loss * weight[i]
            prediction = torch.argmax(event_segmentation, dim=1).squeeze(1)
            nnz = nonzero_idx.long().sum()
            acc = ((prediction == event_label)[nonzero_idx].sum().float() / nnz.float())
            loss = (loss * nonzero_idx.float()).sum() / nnz.float()
            total_loss = total_loss + loss
            total_acc = total_acc + acc
        if torch.is_tensor(total_acc) and total_acc.is_cuda:
            return total_loss, DeferredFloat(total_acc)   # a float when used; no host sync between forward and backward
        return total_loss, float(total_acc)
