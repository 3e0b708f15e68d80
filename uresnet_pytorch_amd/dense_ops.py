"""Dense building blocks used by models/uresnet_dense.py.

conv_bn_act: replicate-pad (asymmetric) -> conv k{1,3} s{1,2} -> BatchNorm with batch statistics
(track_running_stats=False: batch stats in eval too, reference uresnet_dense.py:45) -> optional
residual add -> optional ReLU.  convT_bn_act: ConvTranspose k3 s2 p1 op1 -> BN -> ReLU.

CPU tensors run on torch/ATen (BASELINE configs[0], the reference's own CPU-runnable case).
GPU tensors run on the dense HIP kernels (dense_hip.py -> dense_conv.py -> csrc/urn_dense.hip: implicit-GEMM
convolutions with the replicate clamp in the addressing, fp32 or bf16 MFMA operands) and the BatchNorm row
kernels; a missing library raises, there is no silent fallback.
"""
import torch
import torch.nn.functional as F

_GPU_IMPL = None

def _relu(y):
    """the CPU route's ReLU (one seam: the parity tests pin ReLU branches by patching this function, see tests/relu_hooks.py)"""
    return F.relu(y)


def _gpu():
    global _GPU_IMPL
    if _GPU_IMPL is None:
        from . import dense_hip
        _GPU_IMPL = dense_hip
    return _GPU_IMPL


def conv_bn_act(x, w, b, stride, pad, gamma, beta, eps, relu, residual=None, defer=False):
    """defer=True (shortcut branch of a ResNetModule): the GPU route may hand back the raw convolution output with its
    BatchNorm pending (dense_hip.DeferredBN), to be passed as `residual` of the module's last conv_bn_act -- the shortcut's
    BatchNorm, the add and the ReLU are then one pass.  The CPU route ignores it."""
    if x.is_cuda:
        return _gpu().conv_bn_act(x, w, b, stride, pad, gamma, beta, eps, relu, residual, defer)
    if any(pad):
        x = F.pad(x, pad, mode='replicate')
    conv = F.conv3d if x.dim() == 5 else F.conv2d
    y = conv(x, w, b, stride=stride)
    y = F.batch_norm(y, None, None, gamma, beta, True, 0.0, eps)
    if residual is not None:
        y = y + residual
    return _relu(y) if relu else y


def convT_bn_act(x, w, b, gamma, beta, eps, relu):
    if x.is_cuda:
        return _gpu().convT_bn_act(x, w, b, gamma, beta, eps, relu)
    convT = F.conv_transpose3d if x.dim() == 5 else F.conv_transpose2d
    y = convT(x, w, b, stride=2, padding=1, output_padding=1)
    y = F.batch_norm(y, None, None, gamma, beta, True, 0.0, eps)
    return _relu(y) if relu else y
