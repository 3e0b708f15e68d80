"""Same export names as reference uresnet/models/__init__.py:1-4."""
from .uresnet_dense import UResNet as DenseUResNet
from .uresnet_dense import SegmentationLoss as DenseSegmentationLoss
from .uresnet_sparse import UResNet as SparseUResNet
from .uresnet_sparse import SegmentationLoss as SparseSegmentationLoss

__all__ = ['DenseUResNet', 'DenseSegmentationLoss', 'SparseUResNet', 'SparseSegmentationLoss']
