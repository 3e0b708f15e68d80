#!/usr/bin/env python
"""Same entry point as reference bin/uresnet.py:9-11:
    python bin/uresnet.py train -mn uresnet_sparse -io synthetic_sparse -dd 3 -ss 512 -uf 16 -uns 5 -nc 5 \
        -bs 1 -it 10 --gpus 0 -dkeys data,label
Multi-GPU: torchrun --nproc-per-node N --master-addr 127.0.0.1 bin/uresnet.py train ... --gpus 0,..,N-1"""
import os
import sys

URESNET_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, URESNET_DIR)
from uresnet_pytorch_amd.flags import URESNET_FLAGS  # noqa: E402


def main():
    flags = URESNET_FLAGS()
    flags.parse_args()


if __name__ == '__main__':
    main()
