"""Same export names as reference uresnet/models/__init__.py:1-4."""
from .uresnet_sparse import UResNet as SparseUResNet
from .uresnet_sparse import SegmentationLoss as SparseSegmentationLoss

__all__ = ['SparseUResNet', 'SparseSegmentationLoss']
