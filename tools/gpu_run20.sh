cd $GRAFT_REPO_ROOT
export URN_LIB_PATH=$GRAFT_REPO_ROOT/uresnet_pytorch_amd/liburesnet_hip_diag.so
for shape in "0 16 16" "1 32 32"; do
  for mode in plain fwd; do
    python tools/stamp_pairs.py $shape $mode 2>&1 | grep -v "amdgpu.ids"
  done
done
