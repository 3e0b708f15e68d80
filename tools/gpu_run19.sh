cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_pairs.py -x -q 2>&1 | tail -2
python tools/run_cfg5.py 512 50000 16 5 0 2>&1 | grep "ms/step"
python tools/run_cfg5.py 512 50000 16 5 0 2>&1 | grep "ms/step"
export URN_LIB_PATH=$GRAFT_REPO_ROOT/uresnet_pytorch_amd/liburesnet_hip_diag.so
for shape in "0 16 16" "1 32 32"; do
  for mode in plain fwd; do
    python tools/stamp_pairs.py $shape $mode 2>&1 | grep -v "amdgpu.ids\|waves;"
  done
done
