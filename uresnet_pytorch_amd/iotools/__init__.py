from .iotools import io_factory  # noqa: F401  (same export as reference uresnet/iotools/__init__.py:1)
