#!/bin/bash
# Diagnostic build of the library (in-kernel stamps + timing-only ablation switches): uresnet_pytorch_amd/liburesnet_hip_diag.so
# (tools only: URN_LIB_PATH=uresnet_pytorch_amd/liburesnet_hip_diag.so python tools/stamp_pairs.py ...; the product library carries neither)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/urn_diag_build
rm -rf $B && mkdir -p $B/uresnet_pytorch_amd && cp -r $ROOT/uresnet_pytorch_amd/csrc $B/uresnet_pytorch_amd/csrc && cp -r $ROOT/include $B/include
rm -f $B/uresnet_pytorch_amd/csrc/*.o
make -s -C $B/uresnet_pytorch_amd/csrc -j8 EXTRA="-DURN_PAIRS_STAMP -DURN_DIAG" TARGET=$ROOT/uresnet_pytorch_amd/liburesnet_hip_diag.so
ls -la $ROOT/uresnet_pytorch_amd/liburesnet_hip_diag.so
